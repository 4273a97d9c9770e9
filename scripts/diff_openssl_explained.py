"""The rules behind scripts/diff_openssl.py's buckets: what decides each disagreement between the oracle (= the product)
and OpenSSL 3, who is closer to Go's crypto/x509 + encoding/asn1 as certificate-transparency-go v1.1.0 forked them, and
what this repository does about it.  `status` is one of
    go-rule      the oracle follows a Go rule (recalled) that OpenSSL does not have — nothing to do
    modelled     … and the rule sits behind a switch (named)
    looser       the oracle accepts what Go would also reject: a gap (round 4: the bodies of subjectAltName, cRLDistributionPoints,
                 nameConstraints; round 5: none left)
    openssl      an OpenSSL-only rule (or an OpenSSL leniency) with no counterpart in Go
"""
import re

# ---- direction A: the oracle ACCEPTS, OpenSSL rejects.  (stage, regex over "reason @ where") → (rule, status, text)
A_RULES = [
    ("pubkey-unknown-alg", None, "go-rule",
     "parsePublicKey returns (nil, nil) for an algorithm it does not know (the key is not looked at); OpenSSL cannot build an EVP_PKEY"),
    ("pubkey-finding", None, "go-rule",
     "a key CT-go files a NON-FATAL finding for (INTEGER not minimally encoded, RSA parameters not NULL, modulus <= 0, secp192r1): the certificate is handed out (kept as an X509 entry, dropped as precertificate / Chain[0]); OpenSSL refuses the key"),
    ("pubkey-trailing-in-struct", None, "go-rule",
     "octets behind the publicExponent INSIDE RSAPublicKey, or behind g inside Dss-Parms: encoding/asn1 ignores what follows a struct's last field; OpenSSL's templates do not"),
    ("pubkey-other", None, "openssl",
     "a key OpenSSL refuses for a reason Go has no rule for (examples in the campaign output: a DSA parameter set OpenSSL range-checks, an RSA modulus it finds too small/even) — Go's parsePublicKey only looks at structure and sign"),
    ("d2i", r"explicit length mismatch", "go-rule",
     "EXPLICIT wrapper ([0] version, [3] extensions) whose own length disagrees with its inner element: Go parses the inner element against the enclosing SEQUENCE and resumes behind IT (DESIGN §3.1); OpenSSL checks the wrapper"),
    ("d2i", r"(invalid utf8string|illegal characters|universalstring is wrong length|bmpstring is wrong length|invalid (universal|bmp)string length)", "modelled",
     "character set / length of a Name value: ctmr_set_strict_strings models the stdlib's rules for UTF8String, PrintableString, IA5String and NumericString as a non-fatal finding (opt-in: CT-go's fork is more lenient, unverifiable here); UniversalString and BMPString are not decoded by the Go 1.13-era package at all (value left nil)"),
    ("d2i", r"sequence length mismatch", "go-rule",
     "octets behind the last field of a SEQUENCE (Certificate excepted: 'trailing data'): ignored by encoding/asn1's struct unmarshalling, rejected by OpenSSL's templates"),
    ("d2i", r"illegal padding @ Field=(serialNumber|version)", "go-rule",
     "INTEGER not minimally encoded: only CT-go's lax re-parse accepts it — non-fatal finding WALK_NF_LAX_INTEGER"),
    ("d2i", r"(wrong tag|header too long|too long|explicit tag not constructed|unexpected eoc|type not constructed) @ Field=extensions", "go-rule",
     "[3] whose inner element is not a SEQUENCE (the optional field stays unset: no extensions) or whose length overruns: Go only requires the inner header to parse and the inner element to fit the TBSCertificate"),
    ("d2i", r"(wrong tag|too long|header too long|type not constructed|unexpected eoc|sequence not constructed|illegal padding|boolean is wrong length|invalid object encoding) @ (Field=\w+, )?Type=X509_EXTENSION", "go-rule",
     "[3] whose inner element carries tag 0x10 (SEQUENCE without the constructed bit): not a SEQUENCE to Go — the optional field stays unset and the octets are never looked at; OpenSSL matches the tag number alone and parses them as Extensions"),
    ("d2i", r"mstring (not universal|wrong tag)", "go-rule",
     "AttributeTypeAndValue.Value is `interface{}`: any well-formed TLV that fits is accepted (a non-universal or constructed value is left nil); OpenSSL wants a DirectoryString-like type"),
    ("d2i", r"@ Field=parameter, Type=X509_ALGOR", "go-rule",
     "AlgorithmIdentifier.Parameters is asn1.RawValue: any well-formed TLV; OpenSSL decodes it by its universal type (BOOLEAN length, NULL with contents, primitive/constructed bit, OID arcs)"),
    ("d2i", r"(too long|invalid bit string bits left|header too long|wrong tag|type not primitive) @ Field=(issuerUID|subjectUID)", "go-rule",
     "issuerUniqueID / subjectUniqueID are OPTIONAL implicit BIT STRINGs: an element with the right number but the constructed bit set does not match, is skipped, and — nothing behind it being mandatory — whatever follows is ignored; OpenSSL matches on the number alone"),
    ("d2i", r"(too long|header too long|wrong tag) @ Field=version", "go-rule",
     "[0] wrapper that claims more than its INTEGER / than the TBSCertificate holds: Go never checks the wrapper's length (as for 'explicit length mismatch')"),
    ("d2i", r"(too long|header too long|wrong tag|type not constructed|unexpected eoc|type not primitive|invalid bit string bits left|illegal zero content|string too short|boolean is wrong length|invalid object encoding|illegal padding|bad object header|illegal (null|boolean) value) @ (Field=value, )?Type=X509_NAME_ENTRY", "go-rule",
     "an AttributeTypeAndValue value of a universal type Go does not decode (BOOLEAN, NULL, SEQUENCE, …) or octets behind the value inside the AttributeTypeAndValue: accepted as any TLV that fits / ignored; OpenSSL decodes the value by its type"),
    ("d2i", r"unexpected eoc", "go-rule",
     "a 00 00 element (tag 0, length 0) where Go reads an ANY / ignores trailing octets: a well-formed TLV to encoding/asn1, an end-of-contents marker out of place to OpenSSL"),
    ("time", r"", "go-rule",
     "validity time with a numeric zone whose hours are above 23 (…+8100): time.Parse of the Go 1.13 toolchain does not range-check the zone's hours and Format prints them back, so the serialise-back test passes (DESIGN §3.1); OpenSSL's ASN1_TIME_check refuses it"),
    ("extbc", r"", "go-rule",
     "basicConstraints IS modelled, by Go's struct rules: `struct { IsCA bool optional; MaxPathLen int optional }` — an element of another type leaves the optional field at its default and whatever follows inside the SEQUENCE is ignored (DESIGN §3.1); OpenSSL's BASIC_CONSTRAINTS template wants BOOLEAN then INTEGER and nothing else"),
    ("ext", r"", "go-rule",
     "the BODY of an extension Go 1.13's parseCertificate has no case for (issuerAltName, policyMappings, policyConstraints, "
     "inhibitAnyPolicy, the Netscape extensions, …) is malformed: Go does not look into it (the extension is 'unhandled', which only "
     "matters to Verify when it is critical) — OpenSSL has a decoder for it.  The extensions Go DOES parse — keyUsage, subjectAltName, "
     "nameConstraints, cRLDistributionPoints, authorityKeyIdentifier, extKeyUsage, subjectKeyIdentifier, certificatePolicies, "
     "authorityInfoAccess, CT-go's SCT list — are restated behind strict_extensions / CTMR_PROFILE_REFERENCE and land in the "
     "'modelled' buckets (round 5: no 'looser' bucket is left)"),
]

# ---- direction B: the oracle REJECTS, OpenSSL accepts.  oracle error site name → (status, text)
B_RULES = {
    "RSA exponent <= 0": ("go-rule", "parsePublicKey: 'x509: RSA public exponent is not a positive number' is fatal; OpenSSL takes any INTEGER"),
    "RSA exponent > 8 octets": ("go-rule", "publicExponent is a Go `int`: parseInt64 refuses more than 8 octets; OpenSSL reads a BIGNUM"),
    "RSA exponent": ("go-rule", "publicExponent INTEGER missing / empty / wrong tag inside RSAPublicKey"),
    "RSA modulus": ("go-rule", "modulus INTEGER empty or missing: checkInteger"),
    "RSA key SEQUENCE": ("go-rule", "asn1Data is not a SEQUENCE that fits (after RightAlign of a BIT STRING with pad bits: OpenSSL ignores the pad count for keys)"),
    "RSA trailing": ("go-rule", "'x509: trailing data after RSA public key'"),
    "SPKI BIT STRING": ("go-rule", "parseBitString: pad count above 7, pad bits not zero, or pad bits in an empty string; OpenSSL masks the unused bits"),
    "signature BIT STRING": ("go-rule", "parseBitString on signatureValue, as above"),
    "[1] uniqueID": ("go-rule", "parseBitString on issuerUniqueID"),
    "[2] uniqueID": ("go-rule", "parseBitString on subjectUniqueID"),
    "critical value": ("go-rule", "BOOLEAN must be 0x00 or 0xff ('invalid boolean'); OpenSSL takes any non-zero octet for TRUE"),
    "critical length": ("go-rule", "BOOLEAN of another length than 1"),
    "bc cA value": ("go-rule", "basicConstraints.cA BOOLEAN must be 0x00 or 0xff"),
    "bc cA length": ("go-rule", "basicConstraints.cA BOOLEAN length"),
    "pathLen": ("go-rule", "pathLenConstraint is a Go `int`: non-empty, at most 8 octets, fits int32"),
    "basicConstraints trailing": ("go-rule", "'x509: trailing data after X.509 BasicConstraints'"),
    "basicConstraints SEQUENCE": ("go-rule", "basicConstraints value is not one SEQUENCE"),
    "bc field hdr": ("go-rule", "an optional field's position must hold a well-formed header even when the field is skipped"),
    "bc hdr 2": ("go-rule", "as above, behind cA"),
    "issuer RDN": ("go-rule", "RelativeDistinguishedName must carry tag 0x31 (SET, constructed); OpenSSL's template matching ignores the constructed bit here"),
    "subject RDN": ("go-rule", "as for the issuer"),
    "issuer Name": ("go-rule", "Name must carry tag 0x30 (SEQUENCE, constructed)"),
    "subject Name": ("go-rule", "as for the issuer"),
    "issuer ATV": ("go-rule", "AttributeTypeAndValue must be a constructed SEQUENCE that fits its SET"),
    "subject ATV": ("go-rule", "as for the issuer"),
    "issuer attr OID": ("go-rule", "parseObjectIdentifier: an arc above 2^31 - 1 or longer than 5 octets ('base 128 integer too large'); OpenSSL has no such bound"),
    "subject attr OID": ("go-rule", "as for the issuer"),
    "issuer attr value": ("go-rule", "a Name value of type INTEGER (more than 8 octets), UTCTime / GeneralizedTime (Go's time rules: ±0000, minutes 60..99, Feb 30 …) that Go decodes and OpenSSL keeps as opaque octets"),
    "subject attr value": ("go-rule", "as for the issuer"),
    "EC point": ("go-rule", "elliptic.Unmarshal (Go 1.13) knows the uncompressed form 04 only; OpenSSL also takes the compressed (02/03) and hybrid (06/07) forms"),
    "EC params not an OID": ("go-rule", "explicit / implicit-CA EC parameters: Go wants a named curve"),
    "EC unknown curve": ("go-rule", "a named curve outside P-224/256/384/521 + secp192r1 (secp256k1, brainpool, …): 'unsupported elliptic curve'"),
    "DSA y <= 0": ("go-rule", "'x509: zero or negative DSA parameter'"),
    "DSA param <= 0": ("go-rule", "as above"),
    "DSA param INTEGER": ("go-rule", "Dss-Parms goes through the strict parser only: a not minimally encoded p, q or g is fatal"),
    "DSA params": ("go-rule", "Dss-Parms absent or not a SEQUENCE: asn1.Unmarshal fails; OpenSSL accepts DSA keys with inherited parameters"),
    "DSA trailing": ("go-rule", "'x509: trailing data after DSA public key'"),
    "DSA y": ("go-rule", "the key is not one INTEGER"),
    "notBefore value": ("go-rule", "time.Parse + serialise-back: ±0000, offset minutes 60..99, fractions, a day the month does not have; OpenSSL's d2i does not look inside and ASN1_TIME_check is looser on some of these"),
    "notAfter value": ("go-rule", "as for notBefore"),
    "notBefore hdr": ("go-rule", "validity's first element does not fit"),
    "notAfter hdr": ("go-rule", "validity's second element does not fit"),
    "version INTEGER": ("go-rule", "version is a Go `int`: empty, or does not fit int32"),
    "serial empty": ("go-rule", "empty INTEGER"),
    "critical/extnValue hdr": ("go-rule", "an optional field's position (critical) must hold a well-formed header"),
    "extnValue tag": ("go-rule", "extnValue must be an OCTET STRING (primitive)"),
    "extnValue hdr": ("go-rule", "header behind critical does not parse"),
    "extnID": ("go-rule", "parseObjectIdentifier on extnID: arc above 2^31 - 1 / longer than 5 octets"),
    "Extension": ("go-rule", "Extension must be a constructed SEQUENCE"),
    "extensions SEQUENCE": ("go-rule", "the [3] wrapper's inner header does not parse / overruns the TBSCertificate"),
    "[3] header": ("go-rule", "empty [3] ('zero length explicit tag was not an asn1.Flag') or a malformed header where an optional field is tried"),
    "[0] wrapper": ("go-rule", "empty [0]"),
    "version tag": ("go-rule", "[0] does not hold an INTEGER"),
    "tbs sigalg OID": ("go-rule", "parseObjectIdentifier: arc above 2^31 - 1 / longer than 5 octets"),
    "outer sigalg OID": ("go-rule", "as above"),
    "SPKI alg OID": ("go-rule", "as above"),
    "tbs sigalg": ("go-rule", "AlgorithmIdentifier must be a constructed SEQUENCE"),
    "outer sigalg": ("go-rule", "as above"),
    "SPKI alg": ("go-rule", "as above"),
    "SPKI SEQUENCE": ("go-rule", "subjectPublicKeyInfo must be a constructed SEQUENCE"),
    "validity": ("go-rule", "validity must be a constructed SEQUENCE"),
    "tbsCertificate": ("go-rule", "tbsCertificate must be a constructed SEQUENCE"),
    "outer SEQUENCE": ("go-rule", "Certificate must be a constructed SEQUENCE"),
    "signature tag": ("go-rule", "signatureValue must be a primitive BIT STRING"),
    "serial tag": ("go-rule", "serialNumber must be a primitive INTEGER"),
}


def a_rule(stage, reason_where, sub):
    """(rule name, status, text) or None."""
    if stage == "pubkey":
        for name, _, status, text in A_RULES:
            if name == "pubkey-" + sub:
                return name, status, text
        return None
    if stage == "ext:basicConstraints":
        stage = "extbc:basicConstraints"
    for st, rx, status, text in A_RULES:
        if st == stage.split(":")[0] and rx is not None and re.search(rx, reason_where):
            name = st + ": " + (rx[:48] if rx else "(any)")
            if st in ("ext", "extbc"):
                name = "ext:" + stage.split(":")[1]
            return name, status, text
    return None
