"""One-off CPU fuzz campaign: the product's RFC 6962 entry decoder (csrc/entry_decode.h, host build through
tests/harness) against the oracle's LogEntryFromLeaf restatement on entries with damaged framing.  No GPU.
    python scripts/fuzz_cpu_entries.py <iterations> <seed>
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_mapreduce_amd import synth  # noqa: E402
from tests.test_entry_decode_cpu import mutate_entry, both  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    pairs = []
    for s_, prof in ((31, 0), (32, 1)):
        raw = synth.host_entries(synth.config(seed=s_, n_issuers=6, profile=prof), 0, 300)
        pairs += [(raw.leaf_input(i), raw.extra_data(i)) for i in range(raw.n)]
    n_ok = 0
    t0 = time.time()
    done = 0
    while done < total:
        leaf, extra = pairs[rng.randrange(len(pairs))]
        for _ in range(rng.randrange(1, 4)):
            leaf, extra = mutate_entry(rng, leaf, extra)
            if not leaf or not extra:
                break
        if not leaf or not extra:
            continue
        n_ok += bool(both(leaf, extra).ok)   # asserts on any difference, with and without a misaligning prefix
        done += 1
    print(f"FUZZ CPU ENTRIES OK {total} entries, seed {seed}, {n_ok} decoded, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
