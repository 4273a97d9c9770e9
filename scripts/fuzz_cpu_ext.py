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
from tests.test_ext_cpu import verdicts, x, aia, EKU_SRV, EKU_CLI, POL, OCSP  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    full = D.cert(exts=[x(15, D.tlv(0x03, b"\x05\xa0")), x(14, D.tlv(0x04, b"\x11" * 20)), x(37, D.seq(EKU_SRV, EKU_CLI)),
                        x(35, D.seq(D.tlv(0x80, b"\x22" * 20))), x(32, D.seq(D.seq(POL))),
                        aia(D.seq(D.seq(OCSP, D.tlv(0x86, b"http://o.example")))), D.BC_NOT_CA])
    seeds = [full] * 40
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
    parsed = strict_rejected = 0
    t0 = time.time()
    for r in range(total):
        k = rng.randrange(len(seeds))
        der = seeds[k]
        if ext_ranges[k] and rng.randrange(3):               # two in three: damage inside the extensions block
            c = bytearray(der)
            lo, hi = ext_ranges[k]
            for _ in range(rng.choice((1, 1, 2, 3))):
                p = rng.randrange(lo, hi)
                c[p] = rng.choice((c[p] ^ (1 << rng.randrange(8)), rng.randrange(256), 0x00, 0x80, 0x30, 0x06, 0x04, 0x03))
            der = bytes(c)
        else:
            der = mutate(rng, der)
        a, b = verdicts(der)                                  # asserts oracle == product, switch off and on
        parsed += a
        strict_rejected += a and not b
    print(f"FUZZ CPU EXT OK {total} certificates, seed {seed}, {parsed} parse, {strict_rejected} of them rejected by "
          f"strict_extensions only, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
