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

from scripts.diff_openssl_explained import a_rule, B_RULES  # noqa: E402

PK_OIDS = {bytes.fromhex("2a864886f70d010101"): "rsa", bytes.fromhex("2a864886f70d010107"): "rsa",
           bytes.fromhex("2a8648ce380401"): "dsa", bytes.fromhex("2a8648ce3d0201"): "ec"}


def key_class(der, o):
    """How the oracle's parsePublicKey saw the key of a certificate it accepted: unknown-alg / finding / plain."""
    sp = der[o.spki_off:o.spki_off + o.spki_len]
    i = sp.find(b"\x06")
    alg = PK_OIDS.get(sp[i + 2:i + 2 + sp[i + 1]]) if 0 <= i < 12 else None
    if alg is None:
        return "unknown-alg"
    if o.nonfatal & orc.NF_SPKI:
        return "finding"
    return alg


MODELLED_EXTENSIONS = ("keyUsage", "subjectKeyIdentifier", "extKeyUsage", "authorityKeyIdentifier", "certificatePolicies",
                       "authorityInfoAccess",
                       "subjectAltName", "crlDistributionPoints", "nameConstraints", "ctPrecertSCTs",   # round 5
                       "subjectInfoAccess", "sbgp-ipAddrBlock", "sbgp-autonomousSysNum")                 # round 6: CT-go's own


def seeds():
    out = [(s, "edge") for s in edge_seeds()] * 4 + [(s, "key") for s in key_seeds()] * 3
    from tests.test_ext_cpu import rich_seeds       # round 5: every extension body strict_extensions looks into
    out += [(s, "ext") for s in rich_seeds()] * 12
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


def bucket_of(der, o, v):
    """None when the two agree; else (direction, rule, status, text-or-None, raw) — raw = what to print when no rule matches."""
    o_ok = bool(o.ok)
    s_ok = v.stage == 0
    if o_ok == s_ok:
        return None
    if o_ok:
        stage = {1: "d2i", 2: "trailing", 3: "pubkey", 4: "time", 5: "ext"}[v.stage]
        r = v.reason.decode()
        sub = ""
        if stage == "pubkey":
            sub = key_class(der, o)
            if sub in ("rsa", "dsa", "ec"):
                sub = "trailing-in-struct" if pubkey_has_trailing(der, o) else "other"
        if stage == "ext":
            stage = "ext:%s" % ext_name(v.ext_nid)
        hit = a_rule(stage, r, sub)
        if hit and stage.startswith("ext:") and stage.split(":")[1] in MODELLED_EXTENSIONS:
            # strict_extensions (opt-in) restates Go's rule for this extension: say which side of it the mutant lies on
            if o.ext_fatal:
                hit = (hit[0] + " / fails Go's rules", "modelled",
                       "the value breaks the rule Go's parseCertificate applies to this extension (one BIT STRING / OCTET STRING, "
                       "SEQUENCE OF OID, SEQUENCE OF SEQUENCE { OID, … }, the distributionPoint struct, forEachSAN + url.Parse, the "
                       "cryptobyte reading of nameConstraints; no trailing data): ctmr_set_strict_extensions / CTMR_PROFILE_REFERENCE "
                       "rejects it too, as a fatal parse error (what CT-go's fork makes of it is not verifiable here: hence a profile)")
            elif o.ext_findings:
                hit = (hit[0] + " / a non-fatal finding", "modelled",
                       "what CT-go files as a NON-fatal finding inside the value (an iPAddress of another length than 4 or 16, an "
                       "SCT list that does not decode, an INTEGER only the lax parser takes, an RFC 3779 address block or AS "
                       "identifier list its rpki.go does not decode): under strict_extensions an X509 entry "
                       "keeps its certificate, a precertificate and a Chain[0] issuer are dropped")
            else:
                hit = (hit[0] + " / passes Go's rules", "openssl",
                       "the value satisfies the struct rules Go applies (what follows a policy's OID, the contents of a GeneralName or "
                       "of a key identifier are not looked at by encoding/asn1): OpenSSL's extension decoders go deeper")
        raw = "%s | %s %s" % (stage, r, sub)
        return ("A",) + (hit if hit else (raw, None, None))
    site = SITES.get(o.err_site, "site %d" % o.err_site)
    hit = B_RULES.get(site)
    return ("B", site) + (hit if hit else (None, None))


def pubkey_has_trailing(der, o):
    """An RSA key with octets behind the exponent inside RSAPublicKey / a DSA parameter set with a fourth element."""
    try:
        sp = der[o.spki_off:o.spki_off + o.spki_len]
        c = orc.parse_cert(der, strict_spki=False)
        # walk RSAPublicKey by hand: BIT STRING content behind the AlgorithmIdentifier
        def hdr(b, p):
            ln = b[p + 1]
            if ln < 0x80:
                return p + 2, p + 2 + ln
            k = ln & 0x7f
            return p + 2 + k, p + 2 + k + int.from_bytes(b[p + 2:p + 2 + k], "big")
        s0, _ = hdr(sp, 0)
        a0, a1 = hdr(sp, s0)
        b0, b1 = hdr(sp, a1)
        key = sp[b0 + 1:b1]
        if sp.find(bytes.fromhex("2a864886f70d0101")) >= 0:
            q0, q1 = hdr(key, 0)
            n0, n1 = hdr(key, q0)
            e0, e1 = hdr(key, n1)
            return e1 < q1
        if sp.find(bytes.fromhex("2a8648ce380401")) >= 0:
            p0, p1 = hdr(sp, a0 + 2 + sp[a0 + 1])
            x = p0
            for _ in range(3):
                _, x = hdr(sp, x)
            return x < p1
    except Exception:
        pass
    return False


_EXT = {}


def ext_name(nid):
    names = {82: "subjectKeyIdentifier", 83: "keyUsage", 85: "subjectAltName", 86: "issuerAltName", 87: "basicConstraints",
             88: "crlNumber", 89: "certificatePolicies", 90: "authorityKeyIdentifier", 103: "crlDistributionPoints",
             126: "extKeyUsage", 177: "authorityInfoAccess", 666: "nameConstraints", 747: "policyMappings",
             401: "policyConstraints", 430: "holdInstructionCode", 140: "deltaCRL", 857: "freshestCRL", 748: "inhibitAnyPolicy",
             71: "netscapeCertType", 72: "nsBaseUrl", 78: "nsComment", 951: "ctPrecertSCTs", 952: "ctPrecertPoison",
             398: "subjectInfoAccess", 290: "sbgp-ipAddrBlock", 291: "sbgp-autonomousSysNum"}
    return names.get(nid, "nid%d" % nid)


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
        b = bucket_of(der, o, v)
        if b is None:
            if o.ok:
                agree_acc += 1
            else:
                agree_rej += 1
            continue
        buckets[b] += 1
        if b not in example or len(der) < len(example[b]):
            example[b] = der
    print("# differential accept/reject campaign: oracle (= product, bit for bit) vs OpenSSL %s" % "3")
    print("# %d mutants (seed %d) of %d seed certificates, %.0f s; both accept %d, both reject %d, disagree %d in %d buckets"
          % (total, seed, len(sd), time.time() - t0, agree_acc, agree_rej, sum(buckets.values()), len(buckets)))
    print("# A = the oracle accepts what OpenSSL rejects; B = the oracle rejects what OpenSSL accepts")
    unexplained = 0
    by_status = collections.Counter()
    for b, n in sorted(buckets.items(), key=lambda kv: (kv[0][0], -kv[1])):
        direction, rule, status, text = b[0], b[1], b[2], b[3]
        if status is None:
            unexplained += 1
            print("%s | %-52s | %7d | UNEXPLAINED  example: %s" % (direction, rule, n, example[b].hex()[:1200]))
        else:
            by_status[direction, status] += n
            print("%s | %-52s | %7d | %-8s | %s" % (direction, rule, n, status, text))
    print("# by status: " + ", ".join("%s/%s %d" % (k[0], k[1], v) for k, v in sorted(by_status.items())))
    print("# unexplained buckets: %d" % unexplained)
    sys.exit(1 if unexplained else 0)


if __name__ == "__main__":
    main()
