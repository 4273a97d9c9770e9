/* Test-only: OpenSSL 3 as an INDEPENDENT X.509 field extractor.
 * Not the Go parser — a cross-check that the oracle's walk reads the right bytes. */
#include <openssl/asn1.h>
#include <openssl/err.h>
#include <openssl/objects.h>
#include <openssl/x509.h>
#include <openssl/x509v3.h>
#include <string.h>
#include <time.h>

typedef struct {
  int ok;
  long long not_before, not_after;
  int is_ca;          /* X509_check_ca() > 0 from basicConstraints */
  int has_bc;
  int serial_len;
  unsigned char serial[64];   /* magnitude bytes (OpenSSL normalises) */
  int serial_neg;
  int cn_len;
  unsigned char cn[256];      /* last issuer CN raw value bytes */
  int spki_len;
  unsigned char spki[1024];
} ossl_out;

int ossl_extract(const unsigned char* der, long len, ossl_out* o) {
  memset(o, 0, sizeof *o);
  const unsigned char* p = der;
  X509* x = d2i_X509(NULL, &p, len);
  if (!x || p != der + len) { if (x) X509_free(x); return 0; }
  struct tm tm;
  if (ASN1_TIME_to_tm(X509_get0_notBefore(x), &tm)) o->not_before = timegm(&tm);
  if (ASN1_TIME_to_tm(X509_get0_notAfter(x), &tm)) o->not_after = timegm(&tm);
  BASIC_CONSTRAINTS* bc = X509_get_ext_d2i(x, NID_basic_constraints, NULL, NULL);
  if (bc) { o->has_bc = 1; o->is_ca = bc->ca != 0; BASIC_CONSTRAINTS_free(bc); }
  const ASN1_INTEGER* s = X509_get0_serialNumber(x);
  o->serial_len = s->length < 64 ? s->length : 64;
  memcpy(o->serial, s->data, o->serial_len);
  o->serial_neg = (s->type & V_ASN1_NEG) != 0;
  X509_NAME* n = X509_get_issuer_name(x);
  int idx = -1, last = -1;
  while ((idx = X509_NAME_get_index_by_NID(n, NID_commonName, idx)) >= 0) last = idx;
  if (last >= 0) {
    ASN1_STRING* v = X509_NAME_ENTRY_get_data(X509_NAME_get_entry(n, last));
    o->cn_len = v->length < 256 ? v->length : 256;
    memcpy(o->cn, v->data, o->cn_len);
  }
  unsigned char* sp = o->spki;
  int sl = i2d_X509_PUBKEY(X509_get_X509_PUBKEY(x), NULL);
  if (sl > 0 && sl <= 1024) { i2d_X509_PUBKEY(X509_get_X509_PUBKEY(x), &sp); o->spki_len = sl; }
  o->ok = 1;
  X509_free(x);
  return 1;
}

/* Third opinion on the PUBLIC KEY (round 4): 1 = OpenSSL decodes the certificate AND its key (X509_get_pubkey: the
 * RSAPublicKey structure, an EC point on its named curve, …), 0 = the certificate decodes but the key does not,
 * -1 = the certificate does not decode.  ossl_extract above only re-encodes the raw SPKI and never looks inside. */
int ossl_pubkey_ok(const unsigned char* der, long len) {
  const unsigned char* p = der;
  X509* x = d2i_X509(NULL, &p, len);
  if (!x || p != der + len) { if (x) X509_free(x); return -1; }
  EVP_PKEY* k = X509_get_pubkey(x);
  int ok = k != NULL;
  if (k) EVP_PKEY_free(k);
  X509_free(x);
  return ok;
}

/* The whole third opinion on ACCEPT / REJECT (scripts/diff_openssl.py, round 4): which stage of OpenSSL's handling of the
 * certificate fails first, and why.  stage: 0 = everything decodes; 1 = d2i_X509; 2 = bytes left behind the certificate;
 * 3 = the public key (X509_get_pubkey); 4 = a validity time (ASN1_TIME_check); 5 = the body of an extension OpenSSL has a
 * decoder for (ext_nid says which).  reason = OpenSSL's reason string of the error it raised. */
typedef struct {
  int stage;
  int ext_nid;
  int n_ext;
  char reason[96];
} ossl_verdict_t;

static void take_reason(ossl_verdict_t* v) {
  /* the EARLIEST error of the queue is the root cause ("wrong tag", "invalid object encoding", …); the errors raised on
   * the way up carry "Field=…, Type=…" data: the innermost of those says where */
  unsigned long e;
  const char* data;
  int flags;
  char where[64] = "";
  const char* r = NULL;
  while ((e = ERR_get_error_all(NULL, NULL, NULL, &data, &flags)) != 0) {
    if (!r) r = ERR_reason_error_string(e);
    if (!where[0] && data && (flags & ERR_TXT_STRING) && data[0]) snprintf(where, sizeof where, "%s", data);
  }
  snprintf(v->reason, sizeof v->reason, "%s @ %s", r ? r : "(no reason)", where);
}

int ossl_verdict(const unsigned char* der, long len, ossl_verdict_t* v) {
  memset(v, 0, sizeof *v);
  ERR_clear_error();
  const unsigned char* p = der;
  X509* x = d2i_X509(NULL, &p, len);
  if (!x) { v->stage = 1; take_reason(v); return 1; }
  if (p != der + len) { v->stage = 2; snprintf(v->reason, sizeof v->reason, "trailing bytes"); X509_free(x); return 2; }
  EVP_PKEY* k = X509_get_pubkey(x);
  if (!k) { v->stage = 3; take_reason(v); X509_free(x); return 3; }
  EVP_PKEY_free(k);
  if (ASN1_TIME_check(X509_get0_notBefore(x)) != 1 || ASN1_TIME_check(X509_get0_notAfter(x)) != 1) {
    v->stage = 4; snprintf(v->reason, sizeof v->reason, "time check"); ERR_clear_error(); X509_free(x); return 4;
  }
  int n = X509_get_ext_count(x);
  v->n_ext = n;
  for (int i = 0; i < n; i++) {
    X509_EXTENSION* e = X509_get_ext(x, i);
    if (!X509V3_EXT_get(e)) continue;            /* no decoder for this extension: nothing to say */
    void* d = X509V3_EXT_d2i(e);
    if (!d) {
      v->stage = 5;
      v->ext_nid = OBJ_obj2nid(X509_EXTENSION_get_object(e));
      take_reason(v);
      X509_free(x);
      return 5;
    }
    const X509V3_EXT_METHOD* m = X509V3_EXT_get(e);
    if (m->it) ASN1_item_free(d, ASN1_ITEM_ptr(m->it)); else if (m->ext_free) m->ext_free(d);
  }
  ERR_clear_error();
  X509_free(x);
  return 0;
}
