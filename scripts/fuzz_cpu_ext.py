"""CPU campaign for strict_extensions: the product walk (host build) with the switch on and off against the oracle's
ok / ext_fatal on mutated certificates — synthetic (both profiles), golden, the system CA bundle, hand-built seeds with
every modelled extension.  No GPU.
    python scripts/fuzz_cpu_ext.py <iterations> <seed>
"""
import glob
import os
import random
import ssl
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_mapreduce_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import der as D  # noqa: E402
from tests.test_walk_cpu import mutate  # noqa: E402
from tests.test_ext_cpu import verdicts, rich_seeds, mutate_exts  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    seeds = rich_seeds() * 15          # every extension body the switch looks into (round 5: SAN, CRL DPs, name constraints, SCTs)
    for s_, prof in ((101, 0), (102, 1)):
        cfg = synth.config(seed=s_, n_issuers=8, ca_permille=150, expired_permille=50, profile=prof)
        seeds += [synth.leaf(cfg, i)[0] for i in range(40)]
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.der"))):
        seeds.append(open(f, "rb").read())
    try:
        txt = open("/etc/ssl/certs/ca-certificates.crt").read()
        for blk in txt.split("-----BEGIN CERTIFICATE-----")[1:81]:
            seeds.append(ssl.PEM_cert_to_DER_cert("-----BEGIN CERTIFICATE-----" + blk.split("-----END CERTIFICATE-----")[0]
                                                  + "-----END CERTIFICATE-----\n"))
    except OSError:
        pass
    ext_ranges = []
    for s in seeds:
        o = orc.parse_cert(s)
        ext_ranges.append((o.exts_off, o.exts_end) if o.ok and o.exts_end > o.exts_off else None)
    parsed = strict_rejected = nonfatal = 0
    t0 = time.time()
    for r in range(total):
        k = rng.randrange(len(seeds))
        der = seeds[k]
        if ext_ranges[k] and rng.randrange(3):               # two in three: damage inside the extensions block
            der = mutate_exts(rng, der, *ext_ranges[k])
        else:
            der = mutate(rng, der)
        f = []
        a, b = verdicts(der, f)                               # asserts oracle == product: switch off, on, on with strict_strings
        parsed += a
        strict_rejected += a and not b
        nonfatal += bool(f and (f[0][0] or f[0][1]))
    print(f"FUZZ CPU EXT OK {total} certificates, seed {seed}, {parsed} parse, {strict_rejected} of them rejected by "
          f"strict_extensions only, {nonfatal} accepted with a non-fatal finding inside an extension, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
