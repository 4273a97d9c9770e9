/*
 * ctmr_oracle.h — CPU ORACLE for the ct-mapreduce map/reduce hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker.  The product (ct_mapreduce_amd/) never links or calls it.
 *
 * It is a plain-C restatement of the reference's algorithm (jcjones/ct-mapreduce),
 * each function citing the reference file:line it follows.
 *
 * PARITY PIN STATUS
 *   pinned by the reference's own golden vectors (tests/test_oracle_golden.py):
 *     G1 Issuer.ID known answer          storage/types_test.go:41-57
 *     G2 leading-zero serial 00aa / AKo= storage/types_test.go:21-39,81-101
 *     G3/G4 kRealSPKI / kEmptySPKI parse storage/filesystemdatabase_test.go:16-65,80-111
 *     G5 key format + expiry hour        storage/knowncertificates_test.go:85-110
 *     G6 set semantics                   storage/knowncertificates_test.go:11-83
 *     G9 ExpDate / UniqueCertIdentifier  storage/types_test.go:203-269
 *   PARITY UNPINNED at the third-party parser boundary: x509.ParseCertificate is
 *   github.com/google/certificate-transparency-go v1.1.0 (go.mod:10), absent from
 *   /root/reference and from this machine; no Go toolchain exists here.  The fields
 *   Issuer.CommonName, NotAfter, IsCA/BasicConstraintsValid and every accept/reject
 *   decision follow Go's encoding/asn1 struct-unmarshalling rules applied to the x509 struct definitions (the
 *   "DER walk profile" written down in DESIGN.md §3), cross-checked against OpenSSL 3 on well-formed
 *   certificates (tests/test_walk_cpu.py) but NOT against Go.
 */
#ifndef CTMR_ORACLE_H
#define CTMR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-entry status: order == the order the reference tests things in
 *      insertCTWorker (cmd/ct-fetch/ct-fetch.go:191-235) ---- */
enum {
  ORC_ST_PASS = 0,               /* reached database.Store                         :229 */
  ORC_ST_PARSE_ERROR = 1,        /* x509.ParseCertificate(leaf) failed             :202-209 */
  ORC_ST_FILTERED_CA = 2,        /* certIsFilteredOut: BasicConstraintsValid&&IsCA :47-50 */
  ORC_ST_FILTERED_EXPIRED = 3,   /* NotAfter.Before(now) && !LogExpiredEntries     :52-55 */
  ORC_ST_FILTERED_CN = 4,        /* issuerCNFilter prefix miss                     :57-69 */
  ORC_ST_NO_ISSUER = 5,          /* len(Chain) < 1                                 :215-219 */
  ORC_ST_ISSUER_PARSE_ERROR = 6, /* x509.ParseCertificate(Chain[0]) failed         :221-225 */
  ORC_ST_ENTRY_DECODE_ERROR = 7  /* ct.LogEntryFromLeaf failed: dropped by the downloader :452-459 */
};

/* Result of the TBSCertificate field walk (the subset of *x509.Certificate the
 * reference's path consumes — SURVEY.md §8(a) a2). Offsets are into the DER buffer. */
typedef struct {
  int32_t ok;            /* 1 = accepted by the walk profile, 0 = parse error */
  int32_t err_site;      /* diagnostic: which check failed (not part of parity) */
  uint32_t serial_off, serial_len;   /* raw INTEGER content octets (types.go:165-178) */
  int64_t not_before;    /* unix seconds */
  int64_t not_after;     /* unix seconds */
  uint32_t cn_off, cn_len;           /* issuer CommonName (last CN wins), len 0 if none */
  int32_t bc_valid;      /* basicConstraints extension present and parsed */
  int32_t is_ca;
  uint32_t spki_off, spki_len;       /* RawSubjectPublicKeyInfo: full TLV */
  uint32_t tbs_off, tbs_len;         /* RawTBSCertificate: full TLV */
  uint32_t issuer_off, issuer_len;   /* the issuer Name TLV (RawIssuer) */
  uint32_t exts_off, exts_end;       /* contents of the SEQUENCE OF Extension, 0/0 when the certificate has none */
  int32_t nonfatal;      /* ORC_NF_*: findings CT-go reports as x509.NonFatalErrors — the certificate is handed out
                            all the same; kept for X509 entries, dropped for precertificates and Chain[0] issuers */
  int32_t string_findings; /* ORC_SF_*: Name values that break their string type's character set (Go stdlib rules; kept apart
                              from `nonfatal`: only an engine with strict_strings set acts on them) */
  int32_t spki_fatal;    /* parsePublicKey (CT-go, recalled): 0 = the key parses, else the check that failed — a FATAL error
                            of x509.ParseCertificate; kept apart from `ok`: an engine acts on it unless strict_spki is off */
  int32_t spki_findings; /* ORC_PK_*: what parsePublicKey files as non-fatal (same switch) */
  uint32_t ext_fatal;    /* the body of an extension Go unmarshals does not parse (ext_body_site; Go stdlib rules): 0 = none,
                            else the check that failed.  Only an engine with strict_extensions set acts on it (fatal) */
  int32_t ext_findings;  /* ORC_XF_*: what CT-go files as NON-fatal inside an extension body (same switch; dropped for a
                            precertificate or a Chain[0] issuer, kept for an X509 entry) */
  int32_t ext_string_findings; /* ORC_SF_* inside a distribution point's nameRelativeToCRLIssuer: strict_extensions AND
                                  strict_strings */
} orc_cert;

#define ORC_XF_SAN_IP 1   /* subjectAltName iPAddress of a length other than 4 or 16 */
#define ORC_XF_SCT 2      /* the embedded SCT list (1.3.6.1.4.1.11129.2.4.2) does not decode */
#define ORC_XF_LAX 4      /* an INTEGER inside nameRelativeToCRLIssuer that only the lax re-parse accepts */
#define ORC_XF_RPKI 8     /* RFC 3779 sbgp-ipAddrBlock / sbgp-autonomousSysNum that CT-go's rpki.go does not decode (round 6) */

#define ORC_SF_PRINTABLE 1
#define ORC_SF_NUMERIC 2
#define ORC_SF_IA5 4
#define ORC_SF_UTF8 8
#define ORC_PK_RSA_PARAMS 1     /* "x509: RSA key missing NULL parameters" */
#define ORC_PK_LAX_INTEGER 2    /* an INTEGER of the key only the lax re-parse accepts */
#define ORC_PK_RSA_MODULUS 4    /* "x509: RSA modulus is not a positive number" */
#define ORC_PK_INSECURE_CURVE 8 /* secp192r1 */
#define ORC_NF_NEGATIVE_SERIAL 1 /* "x509: negative serial number" */
#define ORC_NF_LAX_INTEGER 2     /* an INTEGER only CT-go's lax asn1 re-parse accepts: not minimally encoded */

void orc_parse_cert(const uint8_t* der, size_t len, orc_cert* out);
/* a bare TBSCertificate (CT-go x509.ParseTBSCertificate: what LogEntryFromLeaf applies to a precertificate entry's leaf) */
void orc_parse_tbs(const uint8_t* tbs, size_t len, orc_cert* out);

/* FIPS 180-4 SHA-256 (Go stdlib crypto/sha256 in types.go:155-159). */
void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]);
/* base64.URLEncoding (padded) — types.go:146-159, 210-212. out must hold 4*ceil(n/3)+1. */
size_t orc_b64url(const uint8_t* in, size_t n, char* out);
/* storage/filesystemdatabase.go:167-175,196-200: pem.EncodeToMemory of a CERTIFICATE block without headers */
size_t orc_pem_encode(const uint8_t* der, size_t n, char* out);
/* Issuer.ID() = b64url(SHA-256(RawSubjectPublicKeyInfo)); 44 chars + NUL. types.go:124-130 */
void orc_issuer_id(const uint8_t* spki, size_t n, char out[45]);
/* floor(unix/3600): NewExpDateFromTime = NotAfter.Truncate(time.Hour) types.go:339-346 */
int32_t orc_exp_hour(int64_t unix_seconds);
/* ExpDate.ID() "2006-01-02-15", 13 chars + NUL. types.go:379-384 */
void orc_exp_date_id(int32_t exp_hour, char out[16]);
/* "2006-01-02" of a unix time — FilesystemDatabase.markDirty filesystemdatabase.go:141-144 */
void orc_day_id(int64_t unix_seconds, char out[16]);

/* certIsFilteredOut (ct-fetch.go:44-70).  filter = *ctconfig.IssuerCNFilter verbatim
 * (comma-split, pieces NOT trimmed); now = time.Now() as unix seconds (H6: injected).
 * Returns ORC_ST_PASS or one of the FILTERED_* codes. */
int orc_cert_is_filtered_out(const uint8_t* der, const orc_cert* c, const char* filter,
                             size_t filter_len, int log_expired, int64_t now);

/* ---- the reduce: FilesystemDatabase.Store over a MockRemoteCache-like set store
 *      (filesystemdatabase.go:158-211, knowncertificates.go:38-55, mockcache.go:38-61) ---- */
typedef struct orc_engine orc_engine;

orc_engine* orc_engine_new(const char* filter, size_t filter_len, int log_expired, int64_t now);
void orc_engine_free(orc_engine*);

/* One iteration of insertCTWorker's loop body.  issuer_der==NULL ⇔ len(Chain)<1.
 * Returns the status; *was_unknown = knownCerts.WasUnknown(serial) when status==PASS.
 * exp_hour/serial are filled whenever the leaf parsed. */
int orc_engine_entry(orc_engine*, const uint8_t* leaf, size_t leaf_len, int entry_type, const uint8_t* issuer_der,
                     size_t issuer_len, int* was_unknown, int32_t* exp_hour,
                     const uint8_t** serial, uint32_t* serial_len);

/* Direct RemoteCache-style access to the same store (types.go:83-102 subset). */
int orc_set_insert(orc_engine*, const char* key, size_t key_len, const uint8_t* member, size_t n);
int orc_set_contains(orc_engine*, const char* key, size_t key_len, const uint8_t* member, size_t n);
int64_t orc_set_cardinality(orc_engine*, const char* key, size_t key_len);
int64_t orc_key_count(orc_engine*);                       /* number of distinct set keys */
/* i-th key in sorted order (bytewise); returns length, copies ≤cap bytes. */
size_t orc_key_at(orc_engine*, int64_t i, char* out, size_t cap);
/* expiry recorded by the first WasUnknown on a key (knowncertificates.go:44-47,98-104);
 * returns 0 if none recorded. */
int orc_key_expiry(orc_engine*, const char* key, size_t key_len, int64_t* unix_seconds);
/* Members of a set, sorted bytewise (mockcache keeps them sorted, mockcache.go:38-61).
 * Serialised as [u32 len][bytes]...; returns bytes needed; copies if cap suffices. */
size_t orc_set_members(orc_engine*, const char* key, size_t key_len, uint8_t* out, size_t cap);
/* Total members over all keys whose issuer part == issuer_id:
 * storage-statistics.go:44-53 (Σ_expDate SCARD(serials::expDate::issuer)). */
int64_t orc_issuer_count(orc_engine*, const char* issuer_id);
int64_t orc_total_count(orc_engine*);
/* Running `insertCTWorker.Inserted` counter (ct-fetch.go:235): PASS entries, dups included. */
int64_t orc_inserted(orc_engine*);

/* Batch driver over the packed layout: the same loop, used for the
 * cpu_baseline timing and for bulk parity.  issuer_idx[i]==0xFFFFFFFF ⇔ no chain.
 * out_status[n], out_unknown[n], out_exp_hour[n] may be NULL. */
/* entry_type[n]: 0 X509LogEntryType, 1 PrecertLogEntryType (decides what happens to non-fatal parse findings);
 * NULL = all X509. */
void orc_engine_batch(orc_engine*, const uint8_t* payload, const uint64_t* offsets,
                      const uint32_t* issuer_idx, const uint8_t* entry_type, uint64_t n, const uint8_t* issuer_payload,
                      const uint64_t* issuer_offsets, uint32_t n_issuers, uint8_t* out_status,
                      uint8_t* out_unknown, int32_t* out_exp_hour);

/* ---- what IssuerMetadata.Accumulate reads from a newly unknown certificate (storage/issuermetadata.go:92-138):
 *      aCert.Issuer (→ .String(), formatted by the host) as the RawIssuer Name TLV, and aCert.CRLDistributionPoints
 *      = the uniformResourceIdentifier [6] members of every DistributionPoint.distributionPoint.fullName of every
 *      extension 2.5.29.31, in order (Go x509 parseCertificate, RFC 5280 §4.2.1.13).  A DistributionPoints value
 *      that does not decode yields no URIs and bad_crl = 1 (Go would have rejected the certificate: outside the walk
 *      profile of DESIGN.md §3).  Returns 0 when the certificate itself does not parse. ---- */
#define ORC_MAX_CRL 16
typedef struct {
  uint32_t issuer_off, issuer_len;   /* full Name TLV */
  uint32_t n_crl;                    /* URIs found (only the first ORC_MAX_CRL are recorded) */
  uint32_t crl_off[ORC_MAX_CRL], crl_len[ORC_MAX_CRL];
  uint32_t n_crl_ext;                /* occurrences of extension 2.5.29.31 */
  int32_t bad_crl;
} orc_meta;
int orc_cert_meta(const uint8_t* der, size_t len, orc_meta* out);

/* ---- ct.LogEntryFromLeaf as far as the path consumes it (cmd/ct-fetch/ct-fetch.go:452; call sites of the
 *      result :198-204,:215,:221,:476).  The TLS structures are RFC 6962 §3.4 (MerkleTreeLeaf /
 *      TimestampedEntry) and §4.6 (extra_data: certificate_chain | PrecertChainEntry), with the field limits of
 *      certificate-transparency-go v1.1.0's struct tags.  PARITY UNPINNED: the reference holds no raw get-entries
 *      fixture and CT-go is not on this machine; the hand-built vectors of tests/test_entry_decode_cpu.py follow
 *      the RFC text.  Offsets are into the respective buffer. ---- */
typedef struct {
  int32_t ok;              /* 0 = LogEntryFromLeaf returns an error */
  int32_t entry_type;      /* 0 X509LogEntryType, 1 PrecertLogEntryType */
  uint64_t timestamp;      /* TimestampedEntry.Timestamp (ms) */
  int32_t cert_in_extra;   /* 1: the certificate insertCTWorker parses lies in extra_data (Precert.Submitted) */
  uint32_t cert_off, cert_len;
  uint32_t chain0_off, chain0_len;   /* in extra_data; len 0 = len(Chain) < 1 */
  uint32_t n_chain;
  uint32_t tbs_off, tbs_len;         /* precert: TBSCertificate in leaf_input */
} orc_entry;

void orc_decode_entry(const uint8_t* leaf_input, size_t leaf_len, const uint8_t* extra_data, size_t extra_len,
                      orc_entry* out);

/* The downloader + insertCTWorker over raw entries: blob/bounds layout of include/ctmr.h (leaf_input_i =
 * [bounds[2i], bounds[2i+1]), extra_data_i = [bounds[2i+1], bounds[2i+2])).  Entries LogEntryFromLeaf rejects get
 * ORC_ST_ENTRY_DECODE_ERROR.  out_timestamp may be NULL. */
/* 1: a precertificate entry whose leaf TBSCertificate does not parse is undecodable (ct.LogEntryFromLeaf, ct-fetch.go:452) */
void orc_engine_set_strict_leaf(orc_engine*, int on);
/* Go stdlib character-set rules for the string values of both Names, filed as non-fatal findings (default ON since round 6) */
void orc_engine_set_strict_strings(orc_engine*, int on);
/* parsePublicKey's verdict on the key (ON by default: the reference always parses the key); 0 = rounds 1-3 behaviour */
void orc_engine_set_strict_spki(orc_engine*, int on);
void orc_engine_set_strict_extensions(orc_engine*, int on);
void orc_engine_raw_batch(orc_engine*, const uint8_t* blob, const uint64_t* bounds, uint64_t n,
                          uint8_t* out_status, uint8_t* out_unknown, int32_t* out_exp_hour,
                          uint64_t* out_timestamp);

#ifdef __cplusplus
}
#endif
#endif
