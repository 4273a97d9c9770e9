"""One-off CPU fuzz campaign: the product's DER walk (csrc/der_walk.h, host build through tests/harness) against the
oracle on mutated certificates — golden, synthetic (both profiles) and the Go-rule edge seeds.  No GPU.
    python scripts/fuzz_cpu.py <iterations> <seed>
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_mapreduce_amd import synth  # noqa: E402
from tests.test_walk_cpu import mutate, same, edge_seeds  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    seeds = edge_seeds() * 6
    for s_, prof in ((101, 0), (102, 1)):
        cfg = synth.config(seed=s_, n_issuers=8, ca_permille=150, expired_permille=50, profile=prof)
        seeds += [synth.leaf(cfg, i)[0] for i in range(60)]
    gd = os.path.join(ROOT, "tests", "golden")
    for f in sorted(os.listdir(gd)):
        if f.endswith(".der"):
            seeds.append(open(os.path.join(gd, f), "rb").read())
    accepted = 0
    t0 = time.time()
    for r in range(total):
        der = mutate(rng, seeds[rng.randrange(len(seeds))])
        if rng.randrange(3) == 0 and len(der) > 1:
            der = mutate(rng, der)
        accepted += same(der)          # asserts on any difference (status, fields, nonfatal findings, meta positions)
    print(f"FUZZ CPU OK {total} certificates, seed {seed}, {accepted} accepted, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
