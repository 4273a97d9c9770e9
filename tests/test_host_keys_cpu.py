"""The long-serial settlement of a group round (csrc/host_keys.h, used by engine/group.inc round_finish) on the CPU: wire
form round trip, malformed lists, and the verdict against a plain-Python model of "one Redis set answers for every shard"
(storage/rediscache.go:57-65): a member some rank held before the round is known to everybody; otherwise the lowest log
index keeps WasUnknown."""
import ctypes as C
import random

from tests import harness


def wire(order, exp_hour, canon, member):
    L = harness.host_keys_lib()
    buf = (C.c_uint64 * (4 + len(member) // 8 + 1))()
    n = L.harness_host_keys_append(order, exp_hour, canon, member, len(member), buf, len(buf))
    assert n == 3 + (len(member) + 7) // 8
    return list(buf[:n])


def verdict(lists, held):
    """lists[r] = [(order, exp_hour, canon, member), ...] of rank r; held[c] per candidate, rank by rank in list order"""
    L = harness.host_keys_lib()
    words, n_words = [], []
    for lst in lists:
        w = [x for item in lst for x in wire(*item)]
        words += w
        n_words.append(len(w))
    nc = sum(len(lst) for lst in lists)
    wa = (C.c_uint64 * max(len(words), 1))(*words)
    na = (C.c_uint64 * len(lists))(*n_words)
    ha = (C.c_uint64 * max(nc, 1))(*held)
    lost = (C.c_uint8 * max(nc, 1))()
    got = L.harness_host_keys_verdict(wa, na, len(lists), ha, lost, nc)
    assert got == nc, got
    return [int(lost[c]) for c in range(nc)]


def model(lists, held):
    cands = [(item, r) for r, lst in enumerate(lists) for item in lst]
    lost = []
    for c, ((order, eh, canon, member), r) in enumerate(cands):
        same = [(o2, r2, c2) for c2, ((o2, e2, k2, m2), r2) in enumerate(cands) if (e2, k2, m2) == (eh, canon, member)]
        before = any(held[c2] for _, _, c2 in same)
        first = min(same)[2]
        lost.append(int(before or c != first))
    return lost


def test_wire_form_round_trip_and_order_of_fields():
    assert wire(7, -3, 5, b"") == [7, (5 << 32) | 0xfffffffd, 0]
    w = wire(1 << 40, 123456, 0xabcdef, bytes(range(1, 42)))
    assert w[:3] == [1 << 40, (0xabcdef << 32) | 123456, 41] and len(w) == 3 + 6
    assert b"".join(x.to_bytes(8, "little") for x in w[3:])[:41] == bytes(range(1, 42))


def test_verdict_hand_cases():
    a, b = (10, 500, 1, b"A" * 41), (20, 500, 1, b"B" * 44)
    assert verdict([[a], [b]], [0, 0]) == [0, 0]                                   # different members: both keep
    assert verdict([[(30,) + a[1:]], [(12,) + a[1:]], [(25,) + a[1:]]], [0, 0, 0]) == [1, 0, 1]     # lowest log index keeps
    assert verdict([[(30,) + a[1:]], [(12,) + a[1:]], [(25,) + a[1:]]], [0, 0, 1]) == [1, 1, 1]     # somebody held it before
    assert verdict([[(5,) + a[1:]], [(5,) + a[1:]]], [0, 0]) == [0, 1]                             # equal order: lower rank
    assert verdict([[a], [(11, 501, 1, a[3])], [(12, 500, 2, a[3])]], [0, 0, 0]) == [0, 0, 0]       # another set: another member
    assert verdict([[], [a], []], [0]) == [0]
    assert verdict([[], []], []) == []


def test_verdict_against_the_model_on_random_rounds():
    rng = random.Random(12)
    members = [bytes(rng.randrange(256) for _ in range(rng.choice((41, 41, 44, 48, 64, 200)))) for _ in range(12)]
    for _ in range(3000):
        world = rng.randrange(1, 6)
        lists, order = [], 0
        for r in range(world):
            lst, seen = [], set()
            for _ in range(rng.randrange(0, 6)):
                order += rng.randrange(0, 3)                                       # ties across ranks happen
                key = (rng.choice((100, 101)), rng.choice((0, 1)), rng.choice(members))
                if key in seen:
                    continue                                                        # a rank lists a member once per round
                seen.add(key)
                lst.append((order,) + key)
            lists.append(lst)
        nc = sum(len(x) for x in lists)
        held = [int(rng.random() < 0.15) * rng.randrange(1, 4) for _ in range(nc)]
        assert verdict(lists, held) == model(lists, held)


def test_malformed_lists_are_refused():
    L = harness.host_keys_lib()
    good = wire(1, 2, 3, b"x" * 41)
    for bad in (good[:2], good[:-1], good[:2] + [1 << 30] + good[3:], good + [9]):
        wa = (C.c_uint64 * len(bad))(*bad)
        na = (C.c_uint64 * 1)(len(bad))
        ha = (C.c_uint64 * 4)()
        lost = (C.c_uint8 * 4)()
        assert L.harness_host_keys_verdict(wa, na, 1, ha, lost, 4) == -1
