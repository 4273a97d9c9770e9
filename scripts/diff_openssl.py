"""Differential ACCEPT / REJECT campaign against OpenSSL (round 4; needs no Go and no GPU).

The oracle's — and, bit for bit, the product's — accept/reject rules restate certificate-transparency-go's
x509.ParseCertificate from memory (parity unpinned).  OpenSSL 3 is a second, unrelated X.509 parser on
this machine; it is NOT the reference either, but every certificate the two disagree on is a place where one of them is
wrong about DER or where Go and OpenSSL genuinely differ — and each such place must be known.  This script mutates
certificates (the reference's goldens, both synthetic corpora, the Go-rule edge seeds, the key seeds, the system's CA
bundle), asks both parsers, buckets every disagreement by the rule that decided it, and FAILS when a bucket has no entry
in EXPLAINED below.

    python scripts/diff_openssl.py [mutants=400000] [seed=1] > profiles/r04/diff_openssl_buckets.txt

Direction key:  A = the oracle accepts what OpenSSL rejects;  B = the oracle rejects what OpenSSL accepts.
"""
import collections
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ct_mapreduce_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import harness  # noqa: E402
from tests.test_walk_cpu import mutate, edge_seeds  # noqa: E402
from tests.test_spki_cpu import key_seeds, spki_mutate  # noqa: E402

# what each error site of oracle/ctmr_oracle.c's parse stands for (err_site is diagnostic output, not parity)
SITES = {
    1: "length", 2: "outer SEQUENCE", 3: "trailing data", 4: "tbsCertificate", 5: "[0] wrapper", 6: "version tag",
    8: "version INTEGER", 9: "serial tag", 10: "serial empty", 11: "tbs sigalg", 12: "tbs sigalg OID", 13: "tbs sigalg params",
    17: "validity", 18: "notBefore hdr", 19: "notBefore value", 20: "notAfter hdr", 21: "notAfter value", 23: "SPKI SEQUENCE",
    24: "[1] uniqueID", 25: "[2] uniqueID", 26: "[3] header", 27: "extensions SEQUENCE", 28: "Extension", 29: "extnID",
    30: "critical/extnValue hdr", 31: "critical length", 32: "critical value", 33: "extnValue hdr", 34: "extnValue tag",
    35: "basicConstraints SEQUENCE", 36: "basicConstraints trailing", 37: "bc field hdr", 38: "bc cA length", 39: "bc cA value",
    40: "bc hdr 2", 41: "pathLen", 42: "outer sigalg", 43: "outer sigalg OID", 44: "outer sigalg params", 45: "signature tag",
    46: "signature BIT STRING", 50: "issuer Name", 51: "issuer RDN", 52: "issuer ATV", 53: "issuer attr OID", 54: "issuer attr value",
    60: "subject Name", 61: "subject RDN", 62: "subject ATV", 63: "subject attr OID", 64: "subject attr value",
    70: "SPKI alg", 71: "SPKI alg OID", 72: "SPKI alg params", 73: "SPKI BIT STRING",
    80: "RSA key SEQUENCE", 81: "RSA trailing", 82: "RSA modulus", 83: "RSA exponent", 84: "RSA exponent > 8 octets",
    85: "RSA exponent <= 0", 86: "DSA y", 87: "DSA trailing", 88: "DSA y <= 0", 89: "DSA params", 90: "DSA param INTEGER",
    91: "DSA param <= 0", 92: "EC params not an OID", 93: "EC unknown curve", 94: "EC point",
}

EXPLAINED = {}   # filled in below, bucket key → (who is closer to Go, and why)


def X(key, text):
    EXPLAINED[key] = text


def load_explanations():
    # imported late so that the table can sit at the end of the file, behind the code that uses it
    from scripts.diff_openssl_explained import fill
    fill(X)


def seeds():
    out = [(s, "edge") for s in edge_seeds()] * 4 + [(s, "key") for s in key_seeds()] * 3
    for s_, prof in ((201, 0), (202, 1)):
        cfg = synth.config(seed=s_, n_issuers=8, ca_permille=150, expired_permille=50, profile=prof)
        out += [(synth.leaf(cfg, i)[0], "synth%d" % prof) for i in range(40)]
        out += [(synth.issuer(cfg, k), "synth-issuer") for k in range(4)]
    import base64
    gd = os.path.join(ROOT, "tests", "golden")
    for f in sorted(os.listdir(gd)):
        if f.endswith(".pem"):
            pem = open(os.path.join(gd, f)).read()
            out.append((base64.b64decode("".join(l for l in pem.splitlines() if not l.startswith("-----"))), "golden"))
    try:
        from tests.test_real_certs_cpu import bundle_ders
        out += [(d, "root") for d in bundle_ders()[:120]]
    except Exception:
        pass
    return out


def bucket_of(o, v):
    """None when the two agree; else the bucket key."""
    o_ok = bool(o.ok)
    s_ok = v.stage == 0
    if o_ok == s_ok:
        return None
    if o_ok:
        r = v.reason.decode()
        return ("A", {1: "d2i", 2: "trailing", 3: "pubkey", 4: "time", 5: "ext"}[v.stage] + (":nid%d" % v.ext_nid if v.stage == 5 else ""), r)
    return ("B", SITES.get(o.err_site, "site %d" % o.err_site), "")


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    sd = seeds()
    spans = []
    for s, _ in sd:
        c = orc.parse_cert(s)
        spans.append((c.spki_off, c.spki_off + c.spki_len) if c.ok else None)
    buckets = collections.Counter()
    example = {}
    agree_acc = agree_rej = 0
    t0 = time.time()
    for r in range(total):
        i = rng.randrange(len(sd))
        der = sd[i][0]
        if spans[i] and rng.randrange(4) == 0:
            der = spki_mutate(rng, der, *spans[i])
        else:
            der = mutate(rng, der)
        if rng.randrange(3) == 0 and len(der) > 1:
            der = mutate(rng, der)
        o = orc.parse_cert(der)
        v = harness.ossl_verdict(der)
        b = bucket_of(o, v)
        if b is None:
            if o.ok:
                agree_acc += 1
            else:
                agree_rej += 1
            continue
        buckets[b] += 1
        if b not in example or len(der) < len(example[b]):
            example[b] = der
    try:
        load_explanations()
    except ImportError:
        pass
    print("# differential accept/reject campaign: oracle (= product, bit for bit) vs OpenSSL %s" % "3")
    print("# %d mutants (seed %d) of %d seed certificates, %.0f s; both accept %d, both reject %d, disagree %d in %d buckets"
          % (total, seed, len(sd), time.time() - t0, agree_acc, agree_rej, sum(buckets.values()), len(buckets)))
    print("# A = the oracle accepts what OpenSSL rejects; B = the oracle rejects what OpenSSL accepts")
    unexplained = 0
    for b, n in sorted(buckets.items(), key=lambda kv: (kv[0][0], -kv[1])):
        why = EXPLAINED.get(b) or EXPLAINED.get(b[:2])
        if not why:
            unexplained += 1
        print("%s | %-28s | %-40s | %7d | %s" % (b[0], b[1], b[2], n, why or "UNEXPLAINED  example: " + example[b].hex()[:600]))
    print("# unexplained buckets: %d" % unexplained)
    sys.exit(1 if unexplained else 0)


if __name__ == "__main__":
    main()
