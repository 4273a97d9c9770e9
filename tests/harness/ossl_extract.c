/* Test-only: OpenSSL 3 as an INDEPENDENT X.509 field extractor.
 * Not the Go parser — a cross-check that the oracle's walk reads the right bytes. */
#include <openssl/asn1.h>
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
