// test_storage.cpp — the C++ host mirror (include/ctmr_storage.hpp) tested the way the reference tests its
// storage package: every TEST cites the Go test it follows.  `test_storage` runs the suites that need no GPU
// (MockRemoteCache / MockBackend / LocalDiskBackend); `test_storage --gpu` additionally runs the cache-facing
// suites and the batched Store path against the HBM-backed GpuRemoteCache on device 0.
//   usage: test_storage [--gpu] [--golden DIR] [--tmp DIR]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ctmr_storage.hpp"
#include "ctmr_bench.h"          // the synthetic corpus (not part of the drop-in ABI)
#include "ctmr_storage_mocks.hpp"  // the reference's test doubles (MockRemoteCache, MockBackend)

#include <hip/hip_runtime_api.h>  // device buffers of the multi-GPU group test (host API only; g++ -D__HIP_PLATFORM_AMD__)

using namespace ctmr::storage;

static int g_fail = 0, g_checks = 0;
static std::string g_golden = "tests/golden", g_tmp = "/tmp";
#define CHECK(c)                                                              \
  do {                                                                        \
    g_checks++;                                                               \
    if (!(c)) {                                                               \
      g_fail++;                                                               \
      fprintf(stderr, "  FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);          \
    }                                                                         \
  } while (0)
#define CHECK_EQ(a, b)                                                                              \
  do {                                                                                              \
    g_checks++;                                                                                     \
    const auto va_ = (a);                                                                           \
    const auto vb_ = (b);                                                                           \
    if (!(va_ == vb_)) {                                                                            \
      g_fail++;                                                                                     \
      fprintf(stderr, "  FAIL %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b);                      \
    }                                                                                               \
  } while (0)
#define CHECK_THROWS(expr)                                                   \
  do {                                                                       \
    g_checks++;                                                              \
    bool threw_ = false;                                                     \
    try { (void)(expr); } catch (const std::exception&) { threw_ = true; }   \
    if (!threw_) {                                                           \
      g_fail++;                                                              \
      fprintf(stderr, "  FAIL %s:%d: expected throw: %s\n", __FILE__, __LINE__, #expr); \
    }                                                                        \
  } while (0)
#define RUN(fn)                               \
  do {                                        \
    const int before_ = g_fail;               \
    fn;                                       \
    printf("%s %s\n", g_fail == before_ ? "ok  " : "FAIL", #fn); \
  } while (0)

// ---- minimal DER builder (the reference's tests mint P-256 certificates with crypto/x509; the path under test
// never verifies signatures, so hand-built DER serves)
static std::string tlv(uint8_t tag, const std::string& c) {
  std::string o(1, (char)tag);
  if (c.size() < 0x80) o += (char)c.size();
  else if (c.size() < 0x100) { o += (char)0x81; o += (char)c.size(); }
  else { o += (char)0x82; o += (char)(c.size() >> 8); o += (char)c.size(); }
  return o + c;
}
static std::string rdn(uint8_t attr, uint8_t strtag, const std::string& v) {
  return tlv(0x31, tlv(0x30, tlv(0x06, std::string("\x55\x04", 2) + (char)attr) + tlv(strtag, v)));
}
static std::string make_cert(const std::string& serial, const std::string& issuer_name, const std::string& not_after,
                             const std::string& extensions = "") {
  const std::string alg = tlv(0x30, tlv(0x06, std::string("\x2a\x86\x48\xce\x3d\x04\x03\x02", 8)));
  const std::string validity = tlv(0x30, tlv(0x17, "000101000000Z") + tlv(0x17, not_after));
  const std::string spki = tlv(0x30, alg + tlv(0x03, std::string(1, '\0') + std::string(33, '\x04')));
  std::string tbs = tlv(0xa0, tlv(0x02, std::string(1, '\x02'))) + tlv(0x02, serial) + alg + issuer_name + validity +
                    tlv(0x30, rdn(3, 0x0c, "subject")) + spki;
  if (!extensions.empty()) tbs += tlv(0xa3, tlv(0x30, extensions));
  return tlv(0x30, tlv(0x30, tbs) + alg + tlv(0x03, std::string(1, '\0') + std::string(8, '\x01')));
}
static std::string b64std_decode(const std::string& s) {
  std::string t;
  for (char c : s) t += c == '+' ? '-' : (c == '/' ? '_' : c);
  return b64url_decode(t);
}
static std::string load_golden_der(const std::string& name, std::string* pem_text = nullptr) {
  std::string txt;
  if (!LocalDiskBackend::slurp(g_golden + "/" + name + ".pem", &txt)) throw Error("missing golden " + name);
  while (!txt.empty() && (txt.back() == '\n' || txt.back() == ' ')) txt.pop_back();
  txt += "\n";
  if (pem_text) *pem_text = txt;
  std::string body;
  size_t p = txt.find('\n') + 1;
  const size_t e = txt.find("-----END");
  for (; p < e; p++)
    if (txt[p] != '\n') body += txt[p];
  return b64std_decode(body);
}
static std::string byte1(int v) { return std::string(1, (char)v); }

// ================================================================= types_test.go
static void Test_Serial() {  // :59-79,81-170
  const Serial x = Serial::FromHex("DEADBEEF"), y = Serial::FromBytes("\xde\xad\xbe\xef");
  CHECK(x == y);
  CHECK_EQ(x.Cmp(y), 0);
  CHECK_EQ(y.String(), std::string("deadbeef"));
  CHECK(Serial::FromIDString(x.ID()) == x);
  CHECK_THROWS(Serial::FromIDString("not base64"));
  CHECK_THROWS(Serial::FromHex("zz"));  // NewSerialFromHex panics
  for (const char* h : {"ABCDEF", "001100", "ABCDEF0100101010010101010100101010", "00ABCDEF01001010101010101010010101",
                        "FFFFFFFFFFFFFF00F00FFFFFFFFFFFFFFF"}) {
    const Serial s = Serial::FromHex(h);
    CHECK(Serial::FromBinaryString(s.BinaryString()) == s);
    CHECK(Serial::UnmarshalJSON(s.MarshalJSON()) == s);
  }
  CHECK_EQ(Serial::FromHex("CAFEDEAD").AsBigIntDecimal(), std::string("3405700781"));
  CHECK_EQ(Serial::FromHex("00aa").ID(), std::string("AKo="));  // TestSerialFromCertWithLeadingZeroes :81-101
  CHECK_EQ(Serial::FromHex("00aa").HexString(), std::string("00aa"));
  CHECK_THROWS(Serial::UnmarshalJSON("deadbeef"));
}
static void Test_Log() {  // :172-201
  CertificateLog a, b;
  a.ShortURL = "log.example.com/2525";
  b.ShortURL = "yeti2021.ct.digicert.com/log/";
  CHECK_EQ(a.ID(), std::string("bG9nLmV4YW1wbGUuY29tLzI1MjU="));
  CHECK_EQ(b.ID(), std::string("eWV0aTIwMjEuY3QuZGlnaWNlcnQuY29tL2xvZy8="));
  a.MaxEntry = 9;
  a.LastEntryTime = Time::Date(2019, 8, 28, 18, 18, 26, 123000000);
  CHECK_EQ(a.MarshalJSON(), std::string("{\"ShortURL\":\"log.example.com/2525\",\"MaxEntry\":9,\"LastEntryTime\":"
                                        "\"2019-08-28T18:18:26.123Z\",\"LastUpdateTime\":\"0001-01-01T00:00:00Z\"}"));
  CHECK(CertificateLog::UnmarshalJSON(a.MarshalJSON()) == a);
}
static void Test_ExpDate() {  // :203-252
  for (const char* d : {"2004-01-19", "2004-01-19-04", "2004-01-19-23"}) CHECK_EQ(ExpDate::Parse(d).ID(), std::string(d));
  const ExpDate hourless = ExpDate::Parse("2004-01-19");
  CHECK(hourless.IsExpiredAt(Time::Date(2004, 1, 20)));
  CHECK(!hourless.IsExpiredAt(Time::Date(2004, 1, 19, 23, 59, 59)));
  const ExpDate four = ExpDate::Parse("2004-01-19-04");
  CHECK(four.IsExpiredAt(Time::Date(2004, 1, 19, 5)));
  CHECK(!four.IsExpiredAt(Time::Date(2004, 1, 19, 4, 59, 59)));
  const Time date = Time::Date(2004, 1, 20, 4, 22, 19);
  const ExpDate e = ExpDate::FromTime(date);
  CHECK(e.IsExpiredAt(date));
  CHECK(!e.IsExpiredAt(Time::Date(2004, 1, 20).AddMillis(-1)));
  CHECK_EQ(e.ID(), std::string("2004-01-20-04"));
  CHECK(e.ExpireTime().Equal(Time::Date(2004, 1, 20, 4)));
  CHECK_THROWS(ExpDate::Parse("garbage"));
  CHECK_THROWS(ExpDate::Parse("2004-02-30"));
  CHECK_EQ(ExpDate::FromHour(e.hour()).ID(), e.ID());
  CHECK_EQ(ExpDate::FromTime(Time::Unix(-1)).ID(), std::string("1969-12-31-23"));  // Truncate floors
}
static void Test_ParseUniqueCertIdentifier() {  // :254-269
  CHECK_THROWS(UniqueCertIdentifier::Parse("a::b"));
  const std::string expected = "2019-04-28-22::an issuer::AESq_w==";
  CHECK_EQ(UniqueCertIdentifier::Parse(expected).String(), expected);
  CHECK_EQ(IssuerAndDate::Parse("2019-04-28/an issuer").String(), std::string("2019-04-28/an issuer"));
}

// ================================================================= knowncertificates_test.go
static void Suite_KnownCertificates(RemoteCache* cache) {  // Test_Unknown :11-55, Test_KnownCertificatesKnown :57-83
  const Issuer testIssuer = Issuer::FromString("test issuer");
  KnownCertificates kc(ExpDate::Parse("2029-01-30"), testIssuer, cache);
  for (const char* h : {"01", "02", "03", "04"}) cache->SetInsert(kc.serialId(), Serial::FromHex(h).BinaryString());
  for (const char* h : {"01", "02", "03", "04"}) CHECK(!kc.WasUnknown(Serial::FromHex(h)));
  CHECK(kc.WasUnknown(Serial::FromHex("05")));
  CHECK(!kc.WasUnknown(Serial::FromHex("05")));
  // the reference compares the JSON of the set with the five escaped code points 1..5: members are the raw
  // one-byte serials
  const std::vector<std::string> want = {byte1(1), byte1(2), byte1(3), byte1(4), byte1(5)};
  CHECK(cache->SetList(kc.serialId()) == want);
  std::vector<Serial> known = kc.Known();
  std::sort(known.begin(), known.end());
  CHECK_EQ(known.size(), (size_t)5);
  for (size_t i = 0; i < known.size() && i < 5; i++) CHECK_EQ(known[i].BinaryString(), want[i]);
  CHECK_EQ(kc.Count(), (int64_t)5);
  KnownCertificates raw(ExpDate::Parse("2029-01-30-07"), testIssuer, cache);  // serials are byte strings with NULs
  CHECK(raw.WasUnknown(Serial::FromHex("00aa")));
  CHECK(!raw.WasUnknown(Serial::FromHex("00aa")));
  CHECK(raw.WasUnknown(Serial::FromHex("aa")));
  CHECK_EQ(raw.Count(), (int64_t)2);
}
static void Test_Unknown_Mock() {
  MockRemoteCache c;
  Suite_KnownCertificates(&c);
  c.Duplicate = 3;  // Redis SSCAN duplicates: Known() de-duplicates (:65-96)
  KnownCertificates kc(ExpDate::Parse("2029-01-30"), Issuer::FromString("test issuer"), &c);
  CHECK_EQ(kc.Known().size(), (size_t)5);
}
static void Test_ExpireAt_Mock() {  // :85-110
  MockRemoteCache c;
  const Time date = Time::Date(2004, 1, 20, 4, 22, 19);
  KnownCertificates kc(ExpDate::FromTime(date), Issuer::FromString("test issuer"), &c);
  CHECK(kc.WasUnknown(Serial::FromHex("05")));
  CHECK_EQ(kc.serialId(), std::string("serials::2004-01-20-04::test issuer"));
  CHECK_EQ(c.Expirations.size(), (size_t)1);
  CHECK(c.Expirations.count("serials::2004-01-20-04::test issuer") &&
        c.Expirations["serials::2004-01-20-04::test issuer"].Equal(Time::Date(2004, 1, 20, 4)));
}

// ================================================================= issuermetadata_test.go
static void Suite_DuplicateCRLs(RemoteCache* cache) {  // Test_DuplicateCRLs :16-60
  IssuerMetadata meta(Issuer::FromString("issuer"), cache);
  meta.addCRL("ldaps://ldap.crl");
  meta.addCRL("schema://192.168.1.1:129/file.crl");
  meta.addCRL("http://::1/file.crl");
  CHECK_EQ(meta.CRLs().size(), (size_t)1);
  for (const char* v : {"http://::1/file.crl", "http://::1/file.crl ", " http://::1/file.crl ", " http://::1/file.crl   "}) {
    meta.addCRL(v);
    CHECK_EQ(meta.CRLs().size(), (size_t)1);
  }
  meta.addCRL("HTTP://Example.com/a b.crl");  // url.String(): scheme lower-cased, path escaped
  std::vector<std::string> want = {"http://::1/file.crl", "http://Example.com/a%20b.crl"};
  std::vector<std::string> got = meta.CRLs();
  std::sort(got.begin(), got.end());
  CHECK(got == want);
  meta.addCRL("http://host:bad/x.crl");  // invalid port: url.Parse error, ignored
  meta.addCRL("http://host/%zz");        // invalid escape, ignored
  meta.addCRL("no-scheme/file.crl");
  CHECK_EQ(meta.CRLs().size(), (size_t)2);
}
static void Suite_Accumulate(RemoteCache* cache) {  // Test_Accumulate :100-136
  const std::string name = tlv(0x30, rdn(3, 0x0c, "My First Issuer (tm)"));
  const std::string c0 = make_cert(byte1(0), name, "010101000000Z");
  const std::string c1 = make_cert(byte1(1), name, "010101000000Z");
  IssuerMetadata meta(Issuer::FromString("issuer"), cache);
  const ExpDate e = ExpDate::FromTime(Time::Date(2001, 1, 1));
  CHECK(!meta.Accumulate(HostCert(c0), e));  // "Should not have seen this expiration date before"
  CHECK(meta.Accumulate(HostCert(c1), e));
  CHECK_EQ(meta.CRLs().size(), (size_t)0);
  CHECK(meta.Issuers() == std::vector<std::string>{"CN=My First Issuer (tm)"});
}
static void Test_HostCert() {
  // pkix.Name.String() and CRLDistributionPoints of the reference's own fixture (filesystemdatabase_test.go:35-64)
  const std::string real = load_golden_der("kRealSPKI");
  const HostCert rc(real);
  CHECK(rc.crlDistributionPoints == std::vector<std::string>{"http://public.wisekey.com/crl/wcidsg1ca.crl"});
  CHECK_EQ(rc.IssuerString(), std::string("CN=WISeKey CertifyID Standard G1 CA,OU=Copyright (c) 2005 WISeKey SA+"
                                          "OU=International,O=WISeKey,C=CH"));
  const std::string lz = load_golden_der("kLeadingZeroes");
  CHECK_EQ(hex_encode(HostCert(lz).serial), std::string("00aa"));  // types_test.go:81-101
  // escaping rules of RDNSequence.String()
  const std::string name = tlv(0x30, rdn(3, 0x0c, "a,b+c \"q\"") + rdn(10, 0x0c, " x "));
  const std::string esc_cert = make_cert(byte1(1), name, "010101000000Z");
  CHECK_EQ(HostCert(esc_cert).IssuerString(), std::string("CN=a\\,b\\+c \\\"q\\\",O=\\ x\\ "));
  // attribute types outside the nine FillFromRDNSequence knows (DC, emailAddress) and values that are no Go string
  // (BMPString) are dropped by the pinned CT-go's Name.String()
  {
    const std::string dc = tlv(0x31, tlv(0x30, tlv(0x06, std::string("\x09\x92\x26\x89\x93\xf2\x2c\x64\x01\x19", 10)) + tlv(0x16, "example")));
    const std::string mail = tlv(0x31, tlv(0x30, tlv(0x06, std::string("\x2a\x86\x48\x86\xf7\x0d\x01\x09\x01", 9)) + tlv(0x16, "ca@example.org")));
    const std::string odd = tlv(0x30, dc + rdn(6, 0x13, "DE") + rdn(10, 0x1e, std::string("\0O\0r\0g", 6)) + rdn(10, 0x0c, "Org") + mail + rdn(3, 0x0c, "The CA"));
    CHECK_EQ(HostCert(make_cert(byte1(1), odd, "010101000000Z")).IssuerString(), std::string("CN=The CA,O=Org,C=DE"));
  }
  // a synthetic leaf of the benchmark corpus
  ctmr_synth_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.seed = 20260921;
  cfg.n_issuers = 40;
  std::string leaf(4096, '\0');
  uint32_t iss = 0;
  uint8_t et = 0;
  leaf.resize(ctmr_synth_leaf(&cfg, 7, (uint8_t*)&leaf[0], 4096, &iss, &et));
  const HostCert sc(leaf);
  char want_dn[64], want_crl[64];
  snprintf(want_dn, sizeof want_dn, "CN=Synth Issuer %03u,O=Synth CA Org,C=US", iss);
  snprintf(want_crl, sizeof want_crl, "http://crl.synth-%03u.example/ca.crl", iss);
  CHECK_EQ(sc.IssuerString(), std::string(want_dn));
  CHECK(sc.crlDistributionPoints == std::vector<std::string>{want_crl});
  const std::string bad("\x30\x82\xff\xff", 4);
  CHECK_THROWS(HostCert(bad));
}

// ================================================================= filesystemdatabase_test.go
static void Suite_GetIssuerAndDatesFromCache(FilesystemDatabase* db) {  // :218-279
  CHECK_EQ(db->GetIssuerAndDatesFromCache().size(), (size_t)0);
  const Issuer issuer = Issuer::FromString("Honesty Issuer");
  db->GetKnownCertificates(ExpDate::Parse("2040-02-03-19"), issuer).WasUnknown(Serial::FromHex("FEEDBEEF"));
  auto l2 = db->GetIssuerAndDatesFromCache();
  CHECK(l2.size() == 1 && l2[0].expDates.size() == 1);
  db->GetKnownCertificates(ExpDate::Parse("2040-02-03"), issuer).WasUnknown(Serial::FromHex("BEEF"));
  auto l3 = db->GetIssuerAndDatesFromCache();
  CHECK(l3.size() == 1 && l3[0].expDates.size() == 2);
}
static void Suite_LogState(RemoteCache* cache, FilesystemDatabase* db) {  // :281-340
  CertificateLog log = db->GetLogState("go.pher", "");
  CHECK_EQ(log.ShortURL, std::string("go.pher"));
  log = db->GetLogState("log.ct", "/2019");
  CHECK(log.ShortURL == "log.ct/2019" && log.MaxEntry == 0 && log.LastEntryTime.IsZero());
  log.MaxEntry = 9;
  db->SaveLogState(log);
  CHECK(cache->LoadLogState(log.ShortURL) == log);
  const CertificateLog upd = db->GetLogState("log.ct", "/2019");
  CHECK(upd.MaxEntry == 9 && upd.LastEntryTime.IsZero());
}
static void Test_ListExpiration() {  // :132-216
  MockBackend be;
  MockRemoteCache c;
  FilesystemDatabase db(&be, &c);
  for (const char* d : {"2017-11-28", "2018-11-28", "2019-11-28"})
    be.AllocateExpDateAndIssuer(ExpDate::Parse(d), Issuer::FromString("test issuer"));
  auto ids = [&](const Time& t) {
    std::vector<std::string> v;
    for (auto& e : db.ListExpirationDates(t)) v.push_back(e.ID());
    std::sort(v.begin(), v.end());
    return v;
  };
  using V = std::vector<std::string>;
  CHECK(ids(Time::Date(2016, 11, 29, 15, 4, 5)) == (V{"2017-11-28", "2018-11-28", "2019-11-28"}));
  CHECK(ids(Time::Date(2018, 11, 29, 15, 4, 5)) == (V{"2019-11-28"}));
  CHECK(ids(Time::Date(2019, 11, 28, 1, 4, 5)) == (V{"2019-11-28"}));
  CHECK(ids(Time::Date(2020, 11, 29, 15, 4, 5)) == V{});
  CHECK(ids(Time::Date(2018, 11, 28, 23, 59, 59)) == (V{"2018-11-28", "2019-11-28"}));
}
static void Test_NoopBackend() {  // :355-377
  NoopBackend be;
  MockRemoteCache c;
  FilesystemDatabase db(&be, &c);
  db.markDirty(Time());
  CHECK_THROWS(db.ListExpirationDates(Time()));
  CHECK_THROWS(db.ListIssuersForExpirationDate(ExpDate::Parse("2040-02-03")));
}
static void Test_Mock_cache_suites() {
  { MockRemoteCache c; Suite_DuplicateCRLs(&c); }
  { MockRemoteCache c; Suite_Accumulate(&c); }
  { MockRemoteCache c; MockBackend b; FilesystemDatabase db(&b, &c); Suite_GetIssuerAndDatesFromCache(&db); }
  { MockRemoteCache c; MockBackend b; FilesystemDatabase db(&b, &c); Suite_LogState(&c, &db); }
  { MockRemoteCache c; NoopBackend b; FilesystemDatabase db(&b, &c); Suite_LogState(&c, &db); }
}

// ================================================================= localdiskbackend_test.go
static void Test_LocalDisk() {
  const std::string root = g_tmp + "/ctmr_host_root";
  (void)system(("rm -rf '" + root + "' '" + g_tmp + "/2019-11-28'").c_str());
  LocalDiskBackend db(0644, root);
  const Issuer issuer = Issuer::FromString("issuerAKI");
  db.StoreKnownCertificateList(issuer, {Serial::FromHex("01"), Serial::FromHex("02"), Serial::FromHex("03")});
  std::string body;
  CHECK(LocalDiskBackend::slurp(root + "/" + issuer.ID(), &body));
  CHECK_EQ(hex_encode(body), std::string("30310a30320a30330a"));  // Test_KnownCertificateList :60-85
  CertificateLog log = db.LoadLogState("log.ct/2019");       // Test_LogState :87-130
  CHECK(log.ShortURL == "log.ct/2019" && log.MaxEntry == 0 && log.LastEntryTime.IsZero());
  log.MaxEntry = 0xDEADBEEF;
  log.LastEntryTime = Time::Unix(1567016306, 0);
  db.StoreLogState(log);
  const CertificateLog got = db.LoadLogState("log.ct/2019");
  CHECK(got == log);
  CHECK(LocalDiskBackend::slurp(root + "/state/" + CertificateLogIDFromShortURL("log.ct/2019"), &body));
  // certificate path = root/expDateID/issuerID/serialID — no ".pem" suffix (localdiskbackend.go:194-199)
  const ExpDate e = ExpDate::Parse("2019-11-28-04");
  db.StoreCertificatePEM(Serial::FromHex("02"), e, issuer, "\xda\xda");
  CHECK(LocalDiskBackend::slurp(root + "/2019-11-28-04/issuerAKI/Ag==", &body) && body == "\xda\xda");
  CHECK_THROWS(db.LoadCertificatePEM(Serial::FromHex("02"), e, issuer));
  // MarkDirty is relative to the CURRENT directory (localdiskbackend.go:89-91)
  char cwd[4096];
  CHECK(getcwd(cwd, sizeof cwd) != nullptr);
  CHECK(chdir(g_tmp.c_str()) == 0);
  db.MarkDirty("2019-11-28");
  CHECK(chdir(cwd) == 0);
  CHECK(LocalDiskBackend::slurp(g_tmp + "/2019-11-28/dirty", &body) && body == std::string(1, '\0'));
}

// ================================================================= GPU-backed suites (--gpu)
static void Test_IssuerLazyInit_Gpu(GpuEngine&) {  // types_test.go:41-57 (SHA-256 on the GPU)
  const Issuer i = Issuer::FromSPKI(std::string(1, '\xff'));
  CHECK(!i.idIsSet());
  CHECK_EQ(i.ID(), std::string("qBAK5qoZQNC2Y7sxzUZhQuu9vVGHExuS2TgYmHgy64k="));
  CHECK(i.idIsSet());
  CHECK_EQ(i.MarshalJSON(), std::string("\"qBAK5qoZQNC2Y7sxzUZhQuu9vVGHExuS2TgYmHgy64k=\""));
}
static void Test_ExpireAt_Gpu(GpuEngine& eng) {  // knowncertificates_test.go:85-110 + Redis' lazy expiry
  GpuRemoteCache c(eng);
  const std::string spki_issuer = load_golden_der("kEmptySPKI");
  const uint32_t idx = eng.AddIssuer(spki_issuer);
  const Issuer issuer = Issuer::FromString(eng.IssuerInfo(idx).issuer_id);
  CHECK_EQ(issuer.ID(), std::string("VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8="));
  const Time date = Time::Date(2004, 1, 20, 4, 22, 19);
  KnownCertificates kc(ExpDate::FromTime(date), issuer, &c);
  CHECK(kc.WasUnknown(Serial::FromHex("05")));
  CHECK(c.Exists(kc.serialId()));
  uint64_t removed = 0;
  eng.ck(ctmr_expire_sweep(eng.handle(), Time::Date(2004, 1, 20, 3, 59, 59).sec, &removed));
  CHECK(removed == 0 && c.Exists(kc.serialId()));
  eng.ck(ctmr_expire_sweep(eng.handle(), Time::Date(2004, 1, 20, 4).sec, &removed));  // EXPIREAT = the truncated hour
  CHECK(removed == 1 && !c.Exists(kc.serialId()));
}
static void Test_StoreBatch_Gpu(GpuEngine& eng) {
  // filesystemdatabase_test.go Test_Store-style, batched: golden leaf kLeadingZeroes issued under kEmptySPKI
  // (key serials::2020-02-05-00::VCIl…, member 00 aa), plus a synthetic stream with duplicates.
  GpuRemoteCache cache(eng);
  MockBackend be;
  FilesystemDatabase db(&be, &cache, &eng);
  eng.SetFilter("", false, Time::Date(2019, 1, 1).sec);
  std::string pem_text;
  const std::string leaf = load_golden_der("kLeadingZeroes", &pem_text), ca = load_golden_der("kEmptySPKI");
  BatchResult r = db.Store(leaf, ca, "log.ct/2019", 1);
  CHECK(r.stats.n_new == 1 && r.records[0].status == CTMR_ST_PASS);
  const std::string key = "serials::2020-02-05-00::VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8=";
  CHECK(cache.SetList(key) == std::vector<std::string>{Serial::FromHex("00aa").BinaryString()});
  const Issuer issuer = Issuer::FromString("VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8=");
  CHECK_EQ(be.LoadCertificatePEM(Serial::FromHex("00aa"), ExpDate::Parse("2020-02-05-00"), issuer), pem_text);
  CHECK(be.dirty == std::vector<std::string>{"2020-02-05"});
  CHECK(be.allocations == std::vector<std::string>{"2020-02-05-00/" + issuer.ID()});
  CHECK(db.GetIssuerMetadata(issuer)->Issuers() == std::vector<std::string>{"CN=ca"});
  r = db.Store(leaf, ca, "log.ct/2019", 2);  // known now: no PEM, no Accumulate, but still marked dirty (:204-208)
  CHECK(r.stats.n_new == 0 && r.stats.n_dup == 1);
  CHECK_EQ(be.dirty.size(), (size_t)2);
  CHECK_EQ(be.allocations.size(), (size_t)1);
  // the CA certificate as a leaf is filtered (certIsFilteredOut: CA) and an entry without chain is skipped
  CtLogEntry ca_leaf; ca_leaf.leaf_der = ca; ca_leaf.chain0_der = ca;
  CtLogEntry orphan; orphan.leaf_der = leaf; orphan.has_chain = false;
  r = db.StoreBatch({ca_leaf, orphan});
  CHECK(r.records[0].status == CTMR_ST_FILTERED_CA && r.records[1].status == CTMR_ST_NO_ISSUER);
  CHECK_EQ(be.dirty.size(), (size_t)2);

  // synthetic stream through the entryChan consumer
  ctmr_synth_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.seed = 77; cfg.n_issuers = 5; cfg.dup_permille = 200; cfg.ca_permille = 30; cfg.expired_permille = 30;
  eng.SetFilter("Synth Issuer 00", false, 1767225600ll);
  std::vector<std::string> issuers(5);
  for (uint32_t k = 0; k < 5; k++) {
    issuers[k].resize(4096);
    issuers[k].resize(ctmr_synth_issuer(&cfg, k, (uint8_t*)&issuers[k][0], 4096));
  }
  BatchInserter ins(&db, 700);
  const size_t N = 3000;
  for (size_t i = 0; i < N; i++) {
    CtLogEntry e;
    e.leaf_der.resize(4096);
    uint32_t iss; uint8_t et;
    e.leaf_der.resize(ctmr_synth_leaf(&cfg, i, (uint8_t*)&e.leaf_der[0], 4096, &iss, &et));
    e.chain0_der = issuers[iss];
    e.precert = et == 1;
    e.index = (int64_t)i;
    ins.Push(std::move(e));
  }
  ins.Flush();
  const uint64_t stored = ins.counters["insertCTWorker.Inserted"], fresh = ins.counters["WasUnknown"];
  CHECK(stored > 0 && fresh > 0 && fresh < stored);  // duplicates were stored-but-known
  CHECK_EQ(stored + ins.counters["certIsFilteredOut.CA"] + ins.counters["certIsFilteredOut.expired"] +
               ins.counters["certIsFilteredOut.cn-filtered"], (uint64_t)N);
  // storage-statistics over the GPU sets: sum of SCARD per issuer == number of certificates that were unknown
  const StorageStatistics st = StorageStatistics::Collect(&db);
  CHECK_EQ((uint64_t)st.totalSerials, fresh + 1);  // + the golden leaf
  uint64_t pem_files = 0;
  for (auto& kv : be.store) pem_files += kv.first.compare(0, 3, "pem") == 0;
  CHECK_EQ(pem_files, fresh + 1);
  for (auto& is : st.issuers) {
    if (is.issuerID == issuer.ID()) continue;
    CHECK_EQ(is.crls.size(), (size_t)1);
    CHECK(is.dns.size() == 1 && is.dns[0].compare(0, 17, "CN=Synth Issuer 0") == 0);
  }
}

// N2 + N3 through the host mirror: raw get-entries buffers + the IssuerMetadata memo on the GPU must leave the
// cache and the backend in the same state as the packed batch with per-certificate Accumulate on the host.
static void Test_StoreRawBatch_DeviceMeta_Gpu(int chain0_mode = CTMR_CHAIN0_EXACT) {
  ctmr_synth_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.seed = 78; cfg.n_issuers = 7; cfg.dup_permille = 150; cfg.ca_permille = 30; cfg.expired_permille = 30;
  const size_t N = 2500;
  std::vector<std::string> issuers(7);
  for (uint32_t k = 0; k < 7; k++) {
    issuers[k].resize(4096);
    issuers[k].resize(ctmr_synth_issuer(&cfg, k, (uint8_t*)&issuers[k][0], 4096));
  }
  GpuEngine ea(0, 1 << 14, 1 << 12), eb(0, 1 << 14, 1 << 12, 0, /*collect_meta=*/true);
  GpuRemoteCache ca(ea), cb(eb);
  MockBackend ba, bb;
  FilesystemDatabase da(&ba, &ca, &ea), db(&bb, &cb, &eb);
  ea.SetFilter("Synth Issuer 00", false, 1767225600ll);
  eb.SetFilter("Synth Issuer 00", false, 1767225600ll);
  eb.SetChain0Match(chain0_mode);  // trusted-log identification of Chain[0]: the same results on what a log serves
  std::vector<uint8_t> status_a, status_b;
  uint64_t new_a = 0, new_b = 0;
  for (size_t first = 0; first < N; first += 900) {
    const size_t n = std::min<size_t>(900, N - first);
    std::vector<CtLogEntry> entries(n);
    for (size_t i = 0; i < n; i++) {
      CtLogEntry& e = entries[i];
      e.leaf_der.resize(4096);
      uint32_t iss; uint8_t et;
      e.leaf_der.resize(ctmr_synth_leaf(&cfg, first + i, (uint8_t*)&e.leaf_der[0], 4096, &iss, &et));
      e.chain0_der = issuers[iss];
      e.precert = et == 1;
    }
    const BatchResult ra = da.StoreBatch(entries);
    RawEntries raw;
    raw.bounds.resize(2 * n + 1);
    raw.blob.resize(n * 6144);
    raw.blob.resize(ctmr_synth_entries_host(&cfg, first, n, raw.bounds.data(), (uint8_t*)&raw.blob[0], raw.blob.size()));
    const BatchResult rb = db.StoreRawBatch(raw);
    CHECK_EQ(rb.decode.n_x509 + rb.decode.n_precert, (uint64_t)n);
    CHECK(rb.timestamps[0] == (uint64_t)(1767225600ll * 1000 + (long long)first));
    for (size_t i = 0; i < n; i++) {
      status_a.push_back(ra.records[i].status);
      status_b.push_back(rb.records[i].status);
      CHECK(raw.Certificate(i) == entries[i].leaf_der);
    }
    CHECK(ra.new_idx == rb.new_idx);
    new_a += ra.stats.n_new;
    new_b += rb.stats.n_new;
  }
  CHECK(status_a == status_b);
  CHECK(new_a == new_b && new_a > 0);
  const StorageStatistics sa = StorageStatistics::Collect(&da), sb = StorageStatistics::Collect(&db);
  CHECK_EQ(sa.totalSerials, sb.totalSerials);
  CHECK_EQ(sa.issuers.size(), sb.issuers.size());
  for (auto& ia : sa.issuers)
    for (auto& ib : sb.issuers)
      if (ia.issuerID == ib.issuerID) {
        CHECK(ia.crls == ib.crls && ia.dns == ib.dns && ia.crls.size() == 1 && ia.dns.size() == 1);
        CHECK_EQ(ia.serials, ib.serials);
      }
  CHECK(ba.store == bb.store);   // same PEM files, byte for byte
  auto as_set = [](const std::vector<std::string>& v) { return std::set<std::string>(v.begin(), v.end()); };
  CHECK(as_set(ba.allocations) == as_set(bb.allocations));
  CHECK(as_set(ba.dirty) == as_set(bb.dirty));
  // the engine registered the issuers itself, the same seven
  CHECK_EQ(eb.IssuerCounts().size(), (size_t)7);
}

// ---- the entry points a cgo host would add in round 2, bound from C++ as it would bind them
// asynchronous ingestion (ctmr_submit_batch / ctmr_wait): get-entries-sized batches in flight, answers as the
// synchronous call gives them
static void Test_Pipeline_Gpu() {
  GpuEngine sync_e(0, 1 << 18, 1 << 14), async_e(0, 1 << 18, 1 << 14);
  ctmr_synth_config sc;
  memset(&sc, 0, sizeof sc);
  sc.seed = 4711; sc.n_issuers = 8; sc.dup_permille = 300; sc.ca_permille = 20; sc.expired_permille = 20;
  std::string blob;
  std::vector<uint64_t> ioff(1, 0);
  for (uint32_t k = 0; k < 8; k++) {
    std::string c(4096, '\0');
    c.resize(ctmr_synth_issuer(&sc, k, (uint8_t*)&c[0], 4096));
    blob += c;
    ioff.push_back(blob.size());
  }
  for (GpuEngine* e : {&sync_e, &async_e}) {
    uint32_t first;
    e->ck(ctmr_add_issuers(e->handle(), (const uint8_t*)blob.data(), ioff.data(), 8, &first));
    e->ck(ctmr_set_filter(e->handle(), "", 0, 1, 0));
  }
  const uint64_t per = 1001, batches = 40;
  struct B { std::vector<uint8_t> pay; std::vector<uint64_t> off; std::vector<uint32_t> iss; std::vector<uint8_t> et; ctmr_ticket t; };
  std::vector<B> bs(batches);
  for (uint64_t k = 0; k < batches; k++) {
    B& b = bs[k];
    b.off.resize(per + 1); b.iss.resize(per); b.et.resize(per);
    const uint64_t need = ctmr_synth_host(&sc, k * per, per, b.off.data(), nullptr, 0, b.iss.data(), b.et.data());
    b.pay.resize(need + 64);
    ctmr_synth_host(&sc, k * per, per, b.off.data(), b.pay.data(), need + 64, b.iss.data(), b.et.data());
    async_e.ck(ctmr_submit_batch(async_e.handle(), b.pay.data(), b.off.data(), b.iss.data(), b.et.data(), per, &b.t));
  }
  uint64_t dup_total = 0;
  for (uint64_t k = 0; k < batches; k++) {
    B& b = bs[k];
    std::vector<ctmr_record> ra(per), rs(per);
    std::vector<uint64_t> na(per), ns(per);
    ctmr_batch_stats sa, ss;
    async_e.ck(ctmr_wait(async_e.handle(), b.t, ra.data(), na.data(), &sa));
    sync_e.ck(ctmr_map_batch(sync_e.handle(), b.pay.data(), b.off.data(), b.iss.data(), b.et.data(), per, rs.data(), ns.data(), &ss));
    CHECK(memcmp(ra.data(), rs.data(), per * sizeof(ctmr_record)) == 0);
    CHECK_EQ(sa.n_new, ss.n_new);
    CHECK_EQ(sa.n_dup, ss.n_dup);
    CHECK(sa.n_new == 0 || memcmp(na.data(), ns.data(), sa.n_new * 8) == 0);
    dup_total += sa.n_dup;
  }
  CHECK(dup_total > 1000);  // duplicates across batches that were in flight together
  ctmr_batch_stats st;
  CHECK_EQ(ctmr_wait(async_e.handle(), bs[0].t, nullptr, nullptr, &st), (int)CTMR_E_NOTFOUND);  // collected once
  uint64_t ta, ts;
  async_e.ck(ctmr_total_count(async_e.handle(), &ta));
  sync_e.ck(ctmr_total_count(sync_e.handle(), &ts));
  CHECK_EQ(ta, ts);
}

// the same pipeline fed with RAW get-entries responses (ctmr_submit_entries / ctmr_wait_entries): every ticket answers
// like one synchronous ctmr_map_entries over the same stream — records, NEW list, timestamps, decode statistics; the
// engines register the issuers themselves
static void Test_PipelineRaw_Gpu() {
  GpuEngine sync_e(0, 1 << 18, 1 << 14), async_e(0, 1 << 18, 1 << 14);
  ctmr_synth_config sc;
  memset(&sc, 0, sizeof sc);
  sc.seed = 4713; sc.n_issuers = 8; sc.dup_permille = 300; sc.ca_permille = 20; sc.expired_permille = 20;
  for (GpuEngine* e : {&sync_e, &async_e}) e->ck(ctmr_set_filter(e->handle(), "", 0, 1, 0));
  const uint64_t per = 1001, batches = 24;
  struct B { std::vector<uint8_t> blob; std::vector<uint64_t> bounds; ctmr_ticket t; };
  std::vector<B> bs(batches);
  for (uint64_t k = 0; k < batches; k++) {
    B& b = bs[k];
    b.bounds.resize(2 * per + 1);
    const uint64_t need = ctmr_synth_entries_host(&sc, k * per, per, b.bounds.data(), nullptr, 0);
    b.blob.resize(need + 64);
    ctmr_synth_entries_host(&sc, k * per, per, b.bounds.data(), b.blob.data(), need + 64);
    async_e.ck(ctmr_submit_entries(async_e.handle(), b.blob.data(), b.bounds.data(), per, &b.t));
  }
  uint64_t dup_total = 0, x509 = 0, pre = 0;
  for (uint64_t k = 0; k < batches; k++) {
    B& b = bs[k];
    std::vector<ctmr_record> ra(per), rs(per);
    std::vector<uint64_t> na(per), ns(per), ta(per), ts(per);
    ctmr_batch_stats sa, ss;
    ctmr_decode_stats da, ds;
    async_e.ck(ctmr_wait_entries(async_e.handle(), b.t, ra.data(), na.data(), ta.data(), &da, &sa));
    sync_e.ck(ctmr_map_entries(sync_e.handle(), b.blob.data(), b.bounds.data(), per, rs.data(), ns.data(), ts.data(), &ds, &ss));
    CHECK(memcmp(ra.data(), rs.data(), per * sizeof(ctmr_record)) == 0);
    CHECK(ta == ts);
    CHECK_EQ(sa.n_new, ss.n_new);
    CHECK_EQ(sa.n_dup, ss.n_dup);
    CHECK(sa.n_new == 0 || memcmp(na.data(), ns.data(), sa.n_new * 8) == 0);
    CHECK_EQ(da.n, ds.n);
    CHECK_EQ(da.n_x509, ds.n_x509);
    CHECK_EQ(da.n_precert, ds.n_precert);
    CHECK_EQ(da.n_decode_error, ds.n_decode_error);
    CHECK_EQ(da.blob_bytes, ds.blob_bytes);
    dup_total += sa.n_dup; x509 += da.n_x509; pre += da.n_precert;
  }
  CHECK(dup_total > 500 && x509 > 1000 && pre > 1000);
  uint32_t ia = 0, is = 0;
  async_e.ck(ctmr_issuer_count(async_e.handle(), &ia));
  sync_e.ck(ctmr_issuer_count(sync_e.handle(), &is));
  CHECK(ia == is && ia > 0 && ia <= 8);
  uint64_t ca, cs;
  async_e.ck(ctmr_total_count(async_e.handle(), &ca));
  sync_e.ck(ctmr_total_count(sync_e.handle(), &cs));
  CHECK_EQ(ca, cs);
}

// multi-GPU groups (ctmr_group_*): two engines on one device as a local group; the owner-computes and the Bloom
// exchange give the same global answer as one engine over the whole stream
static void Test_Group_Gpu() {
  ctmr_synth_config sc;
  memset(&sc, 0, sizeof sc);
  sc.seed = 4712; sc.n_issuers = 8; sc.dup_permille = 300; sc.ca_permille = 20; sc.expired_permille = 20;
  std::string blob;
  std::vector<uint64_t> ioff(1, 0);
  for (uint32_t k = 0; k < 8; k++) {
    std::string c(4096, '\0');
    c.resize(ctmr_synth_issuer(&sc, k, (uint8_t*)&c[0], 4096));
    blob += c;
    ioff.push_back(blob.size());
  }
  const uint64_t n_total = 6000, half = n_total / 2;
  auto setup = [&](GpuEngine& e) {
    uint32_t first;
    e.ck(ctmr_add_issuers(e.handle(), (const uint8_t*)blob.data(), ioff.data(), 8, &first));
    e.ck(ctmr_set_filter(e.handle(), "", 0, 1, 0));
  };
  struct Dev { uint8_t* pay; uint64_t* off; uint32_t* iss; uint8_t* et; ctmr_record* rec; uint64_t* nw; uint64_t n; };
  auto make_shard = [&](GpuEngine& e, uint64_t first, uint64_t n) {
    Dev d;
    d.n = n;
    CHECK(hipMalloc((void**)&d.off, (n + 1) * 8) == hipSuccess);
    CHECK(hipMalloc((void**)&d.iss, n * 4) == hipSuccess);
    CHECK(hipMalloc((void**)&d.et, n) == hipSuccess);
    CHECK(hipMalloc((void**)&d.rec, n * sizeof(ctmr_record)) == hipSuccess);
    CHECK(hipMalloc((void**)&d.nw, n * 8) == hipSuccess);
    uint64_t bytes = 0;
    e.ck(ctmr_synth_device(e.handle(), &sc, first, n, d.off, nullptr, 0, nullptr, nullptr, &bytes));
    CHECK(hipMalloc((void**)&d.pay, bytes + 64) == hipSuccess);
    e.ck(ctmr_synth_device(e.handle(), &sc, first, n, d.off, d.pay, bytes + 64, d.iss, d.et, &bytes));
    return d;
  };
  auto to_shard = [](const Dev& d, uint64_t base) {
    ctmr_shard s;
    memset(&s, 0, sizeof s);
    s.d_payload = d.pay; s.d_offsets = d.off; s.d_issuer_idx = d.iss; s.d_entry_type = d.et; s.n = d.n;
    s.order_base = base; s.d_records = d.rec; s.d_new_idx = d.nw;
    return s;
  };
  // the reference answer: one engine, the whole stream
  GpuEngine one(0, 1 << 16, 1 << 12);
  setup(one);
  Dev whole = make_shard(one, 0, n_total);
  ctmr_batch_stats st_one;
  one.ck(ctmr_map_batch_device(one.handle(), whole.pay, whole.off, whole.iss, whole.et, n_total, whole.rec, whole.nw, &st_one));
  std::vector<ctmr_record> want(n_total);
  CHECK(hipMemcpy(want.data(), whole.rec, n_total * sizeof(ctmr_record), hipMemcpyDeviceToHost) == hipSuccess);
  std::vector<uint64_t> counts_one(8);
  one.ck(ctmr_issuer_counts(one.handle(), counts_one.data(), 8));
  for (int mode : {(int)CTMR_DEDUP_OWNER, (int)CTMR_DEDUP_BLOOM}) {
    GpuEngine a(0, 1 << 16, 1 << 12), b(0, 1 << 16, 1 << 12);
    setup(a); setup(b);
    ctmr_engine* engines[2] = {a.handle(), b.handle()};
    ctmr_group* g = nullptr;
    CHECK_EQ(ctmr_group_create_local(engines, 2, &g), (int)CTMR_OK);
    if (mode == CTMR_DEDUP_BLOOM) CHECK_EQ(ctmr_group_bloom_config(g, 1ull << 16), (int)CTMR_OK);
    Dev d0 = make_shard(a, 0, half), d1 = make_shard(b, half, n_total - half);
    ctmr_shard sh[2] = {to_shard(d0, 0), to_shard(d1, half)};
    ctmr_batch_stats st[2];
    const int rc = ctmr_group_map_batch(g, mode, sh, st);
    if (rc != CTMR_OK) fprintf(stderr, "group: %s\n", ctmr_group_last_error(g));
    CHECK_EQ(rc, (int)CTMR_OK);
    std::vector<ctmr_record> got(n_total);
    CHECK(hipMemcpy(got.data(), d0.rec, half * sizeof(ctmr_record), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(hipMemcpy(got.data() + half, d1.rec, (n_total - half) * sizeof(ctmr_record), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(memcmp(got.data(), want.data(), n_total * sizeof(ctmr_record)) == 0);   // WAS_UNKNOWN flags are global
    CHECK_EQ(st[0].n_new + st[1].n_new, st_one.n_new);
    std::vector<uint64_t> counts(8);
    CHECK_EQ(ctmr_group_issuer_counts(g, counts.data(), 8), (int)CTMR_OK);
    CHECK(counts == counts_one);
    uint64_t total = 0;
    CHECK_EQ(ctmr_group_total_count(g, &total), (int)CTMR_OK);
    CHECK_EQ(total, st_one.n_new);
    ctmr_group_stats gi;
    CHECK_EQ(ctmr_group_info(g, &gi), (int)CTMR_OK);
    CHECK(gi.world == 2 && gi.n_local == 2 && gi.transport == CTMR_TRANSPORT_LOCAL && gi.keys_sent > 0);
    ctmr_group_destroy(g);
    for (Dev* d : {&d0, &d1}) { hipFree(d->pay); hipFree(d->off); hipFree(d->iss); hipFree(d->et); hipFree(d->rec); hipFree(d->nw); }
  }
  hipFree(whole.pay); hipFree(whole.off); hipFree(whole.iss); hipFree(whole.et); hipFree(whole.rec); hipFree(whole.nw);
}

int main(int argc, char** argv) {
  bool gpu = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--gpu") gpu = true;
    else if (a == "--golden" && i + 1 < argc) g_golden = argv[++i];
    else if (a == "--tmp" && i + 1 < argc) g_tmp = argv[++i];
  }
  RUN(Test_Serial());
  RUN(Test_Log());
  RUN(Test_ExpDate());
  RUN(Test_ParseUniqueCertIdentifier());
  RUN(Test_Unknown_Mock());
  RUN(Test_ExpireAt_Mock());
  RUN(Test_HostCert());
  RUN(Test_Mock_cache_suites());
  RUN(Test_ListExpiration());
  RUN(Test_NoopBackend());
  RUN(Test_LocalDisk());
  if (gpu) {
    try {
      GpuEngine eng(0, 1 << 16, 1 << 14);
      RUN(Test_IssuerLazyInit_Gpu(eng));
      { GpuRemoteCache c(eng); RUN(Suite_KnownCertificates(&c)); }
      { GpuRemoteCache c(eng); RUN(Suite_DuplicateCRLs(&c)); }
      { GpuEngine e2(0, 1 << 12, 1 << 10); GpuRemoteCache c(e2); RUN(Suite_Accumulate(&c)); }
      { GpuEngine e2(0, 1 << 12, 1 << 10); GpuRemoteCache c(e2); MockBackend b; FilesystemDatabase db(&b, &c, &e2);
        RUN(Suite_GetIssuerAndDatesFromCache(&db)); }
      { GpuEngine e2(0, 1 << 12, 1 << 10); GpuRemoteCache c(e2); MockBackend b; FilesystemDatabase db(&b, &c, &e2);
        RUN(Suite_LogState(&c, &db)); }
      RUN(Test_ExpireAt_Gpu(eng));
      GpuEngine eng3(0, 1 << 16, 1 << 14);
      RUN(Test_StoreBatch_Gpu(eng3));
      RUN(Test_StoreRawBatch_DeviceMeta_Gpu());
      RUN(Test_StoreRawBatch_DeviceMeta_Gpu(CTMR_CHAIN0_TRUSTED_LOG));
      RUN(Test_Pipeline_Gpu());
      RUN(Test_PipelineRaw_Gpu());
      RUN(Test_Group_Gpu());
    } catch (const std::exception& ex) {
      fprintf(stderr, "GPU suites aborted: %s\n", ex.what());
      g_fail++;
    }
  } else {
    CHECK_THROWS(Issuer::FromSPKI("x").ID());  // no engine, no digest: the host has no SHA-256 fallback
  }
  printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail ? 1 : 0;
}
