"""Shared helpers for the -m gpu parity tests: run a host batch through the C ABI and through
the oracle, and compare every observable bit for bit."""
import numpy as np

from oracle import oracle as orc


def run_oracle(batch, issuers, filt=b"", log_expired=False, now=0, engine=None):
    o = engine or orc.Engine(filt, log_expired, now)
    io = np.zeros(len(issuers) + 1, np.uint64)
    if issuers:
        io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8) if issuers else np.zeros(1, np.uint8)
    st, unk, eh = o.batch(batch.payload if len(batch.payload) else np.zeros(1, np.uint8), batch.offsets,
                          batch.issuer_idx, blob, io, entry_type=batch.entry_type)
    return o, st, unk, eh


def expected_records(batch, st, unk, eh, strict_strings=True, strict_spki=True, strict_ext=True):
    """What the 32-byte records must contain, derived from the oracle.  The switches default to the engine's own defaults:
    CTMR_PROFILE_REFERENCE, everything on (round 6)."""
    n = batch.n
    serial_len = np.zeros(n, np.uint16)
    serial = np.zeros((n, 20), np.uint8)
    flags = np.zeros(n, np.uint8)
    exp_hour = np.zeros(n, np.int32)
    for i in range(n):
        der = batch.cert(i)
        c = orc.parse_cert(der, strict_spki)
        if batch.entry_type[i] == 1:
            flags[i] |= 1
        nonfatal = c.nonfatal or (strict_strings and c.string_findings) or \
            (strict_ext and (c.ext_findings or (strict_strings and c.ext_string_findings)))
        if not c.ok or (strict_ext and c.ext_fatal) or (batch.entry_type[i] == 1 and nonfatal):     # a dropped certificate reports no fields
            continue
        exp_hour[i] = eh[i]
        serial_len[i] = min(c.serial_len, 0xffff)
        s = der[c.serial_off:c.serial_off + min(c.serial_len, 20)]
        serial[i, :len(s)] = np.frombuffer(s, np.uint8)
        if c.serial_len > 20:
            flags[i] |= 4
        if unk[i]:
            flags[i] |= 2
    return flags, serial_len, exp_hour, serial


def assert_records_equal(res, batch, st, unk, eh, strict_strings=True, strict_spki=True, strict_ext=True):
    flags, serial_len, exp_hour, serial = expected_records(batch, st, unk, eh, strict_strings, strict_spki, strict_ext)
    r = res.records
    assert (r["status"] == st).all(), np.nonzero(r["status"] != st)[0][:10]
    assert (r["flags"] == flags).all(), np.nonzero(r["flags"] != flags)[0][:10]
    assert (r["serial_len"] == serial_len).all()
    assert (r["exp_hour"] == exp_hour).all()
    assert (r["issuer_idx"] == batch.issuer_idx).all()
    assert (r["serial"] == serial).all()
    assert (res.new_idx == np.nonzero(unk)[0]).all()
    assert res.stats.n_new == int(unk.sum())
    assert res.stats.n == batch.n
    for k in range(8):
        assert res.stats.by_status[k] == int((st == k).sum()), k


def assert_state_equal(eng, o, n_issuers, sample_keys=50):
    okeys = [k for k in o.keys() if k.startswith(b"serials::")]
    gkeys = sorted(eng.keys(b"serials::*"))
    assert gkeys == okeys
    assert eng.total_count() == o.total_count()
    counts = eng.issuer_counts()
    for k in range(n_issuers):
        info = eng.issuer_info(k)
        if info.valid:
            assert int(counts[k]) == o.issuer_count(info.issuer_id.decode()), k
    step = max(1, len(okeys) // sample_keys)
    for key in okeys[::step]:
        assert eng.set_cardinality(key) == o.set_cardinality(key)
        assert eng.set_list(key) == o.members(key)
