// ctmr_storage.hpp — C++ host mirror of the reference's `storage` package (and of the entryChan consumer of
// cmd/ct-fetch) above the C ABI of libctmr (include/ctmr.h).
//
// The reference is Go; this image has no Go toolchain, so the host side a Go maintainer would write with cgo
//  is written here in C++ with the reference's own names, argument meaning and error behaviour,
// so that tests/host/test_storage.cpp reads like storage/*_test.go:
//
//   types            SPKI, Issuer, Serial, ExpDate, UniqueCertIdentifier, IssuerAndDate, CertificateLog   storage/types.go
//   RemoteCache      interface (types.go:83-102); GpuRemoteCache = libctmr  (the reference's test double MockRemoteCache:
//                    tests/host/ctmr_storage_mocks.hpp)
//   KnownCertificates                                                                                     storage/knowncertificates.go
//   IssuerMetadata                                                                                        storage/issuermetadata.go
//   StorageBackend   interface (types.go:46-68); NoopBackend, LocalDiskBackend                            storage/{noop,localdisk}backend.go
//                    (MockBackend: tests/host/ctmr_storage_mocks.hpp)
//   FilesystemDatabase  Store + StoreBatch (the batched insertCTWorker → Store)                           storage/filesystemdatabase.go
//   BatchInserter    the entryChan consumer: collects CtLogEntry values, flushes one GPU batch            cmd/ct-fetch/ct-fetch.go:180-246
//   StorageStatistics                                                                                    cmd/storage-statistics/storage-statistics.go:28-82
//
// Go (value, error) returns become return values + exceptions (storage::Error); Go panics (NewSerialFromHex) throw
// std::logic_error.  Channels become callbacks/vectors; the callee-closes-channel contract has no equivalent.
// The per-entry hot path (parse, filters, WasUnknown, PEM encoding) runs on the GPU through the C ABI; nothing in
// this header parses certificates per entry — HostCert below is used for NEWLY UNKNOWN certificates only, exactly
// where the reference calls IssuerMetadata.Accumulate (filesystemdatabase.go:183-201).
#pragma once
#include <algorithm>
#include <array>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "ctmr.h"

namespace ctmr {
namespace storage {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------------------------------ encodings
inline std::string b64url_encode(const std::string& in) {  // base64.URLEncoding (padded)
  static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-_";
  std::string o;
  size_t i = 0, n = in.size();
  auto b = [&](size_t k) { return (uint32_t)(uint8_t)in[k]; };
  for (; i + 3 <= n; i += 3) {
    const uint32_t v = (b(i) << 16) | (b(i + 1) << 8) | b(i + 2);
    o += A[(v >> 18) & 63]; o += A[(v >> 12) & 63]; o += A[(v >> 6) & 63]; o += A[v & 63];
  }
  if (n - i == 1) {
    const uint32_t v = b(i) << 16;
    o += A[(v >> 18) & 63]; o += A[(v >> 12) & 63]; o += "==";
  } else if (n - i == 2) {
    const uint32_t v = (b(i) << 16) | (b(i + 1) << 8);
    o += A[(v >> 18) & 63]; o += A[(v >> 12) & 63]; o += A[(v >> 6) & 63]; o += '=';
  }
  return o;
}

inline std::string b64url_decode(const std::string& s) {  // base64.URLEncoding.DecodeString: strict, padded
  auto val = [](char c) -> int {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '-') return 62;
    if (c == '_') return 63;
    return -1;
  };
  if (s.size() % 4 != 0) throw Error("illegal base64 data at input byte " + std::to_string(s.size() - s.size() % 4));
  std::string o;
  for (size_t i = 0; i < s.size(); i += 4) {
    int v[4], pad = 0;
    for (int k = 0; k < 4; k++) {
      const char c = s[i + k];
      if (c == '=') {
        if (i + 4 != s.size() || k < 2) throw Error("illegal base64 data at input byte " + std::to_string(i + k));
        v[k] = 0;
        pad++;
      } else {
        if (pad) throw Error("illegal base64 data at input byte " + std::to_string(i + k));
        v[k] = val(c);
        if (v[k] < 0) throw Error("illegal base64 data at input byte " + std::to_string(i + k));
      }
    }
    const uint32_t w = (v[0] << 18) | (v[1] << 12) | (v[2] << 6) | v[3];
    o += (char)(w >> 16);
    if (pad < 2) o += (char)(w >> 8);
    if (pad < 1) o += (char)w;
  }
  return o;
}

inline std::string hex_encode(const std::string& in) {
  static const char H[] = "0123456789abcdef";
  std::string o;
  for (unsigned char c : in) { o += H[c >> 4]; o += H[c & 15]; }
  return o;
}
inline bool hex_decode(const std::string& s, std::string* out) {
  if (s.size() % 2) return false;
  out->clear();
  auto v = [](char c) -> int {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
  };
  for (size_t i = 0; i < s.size(); i += 2) {
    const int a = v(s[i]), b = v(s[i + 1]);
    if (a < 0 || b < 0) return false;
    *out += (char)((a << 4) | b);
  }
  return true;
}

// ------------------------------------------------------------------------------------------ time
// Go time.Time restricted to UTC: seconds since the Unix epoch + nanoseconds.  The zero Time is 0001-01-01.
struct Time {
  int64_t sec = -62135596800ll;
  int32_t nsec = 0;
  static Time Unix(int64_t s, int32_t ns = 0) { Time t; t.sec = s; t.nsec = ns; return t; }
  static int64_t days_from_civil(int64_t y, unsigned m, unsigned d) {
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m > 2 ? m - 3 : m + 9) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
  }
  static Time Date(int y, unsigned mo, unsigned d, unsigned h = 0, unsigned mi = 0, unsigned s = 0, int32_t ns = 0) {
    return Unix(days_from_civil(y, mo, d) * 86400 + h * 3600 + mi * 60 + s, ns);
  }
  void civil(int64_t* y, unsigned* m, unsigned* d, unsigned* hh, unsigned* mm, unsigned* ss) const {
    int64_t days = sec / 86400, rem = sec % 86400;
    if (rem < 0) { rem += 86400; days -= 1; }
    int64_t z = days + 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const unsigned doe = (unsigned)(z - era * 146097);
    const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const unsigned mp = (5 * doy + 2) / 153;
    *d = doy - (153 * mp + 2) / 5 + 1;
    *m = mp < 10 ? mp + 3 : mp - 9;
    *y = (int64_t)yoe + era * 400 + (*m <= 2);
    *hh = (unsigned)(rem / 3600); *mm = (unsigned)(rem % 3600 / 60); *ss = (unsigned)(rem % 60);
  }
  bool IsZero() const { return sec == -62135596800ll && nsec == 0; }
  bool Before(const Time& o) const { return sec < o.sec || (sec == o.sec && nsec < o.nsec); }
  bool Equal(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
  Time AddMillis(int64_t ms) const {
    int64_t ns = (int64_t)nsec + (ms % 1000) * 1000000ll, s = sec + ms / 1000;
    if (ns < 0) { ns += 1000000000ll; s -= 1; }
    if (ns >= 1000000000ll) { ns -= 1000000000ll; s += 1; }
    return Unix(s, (int32_t)ns);
  }
  Time TruncateHour() const {  // t.Truncate(time.Hour): floor for every representable time
    int64_t h = sec / 3600;
    if (sec % 3600 < 0) h -= 1;
    return Unix(h * 3600, 0);
  }
  std::string Format(bool with_hour) const {  // "2006-01-02" / "2006-01-02-15"
    int64_t y; unsigned m, d, hh, mm, ss;
    civil(&y, &m, &d, &hh, &mm, &ss);
    char b[40];
    if (with_hour) snprintf(b, sizeof b, "%04lld-%02u-%02u-%02u", (long long)y, m, d, hh);
    else snprintf(b, sizeof b, "%04lld-%02u-%02u", (long long)y, m, d);
    return b;
  }
  std::string RFC3339Nano() const {  // what encoding/json writes for a time.Time in UTC
    int64_t y; unsigned m, d, hh, mm, ss;
    civil(&y, &m, &d, &hh, &mm, &ss);
    char b[64];
    snprintf(b, sizeof b, "%04lld-%02u-%02uT%02u:%02u:%02u", (long long)y, m, d, hh, mm, ss);
    std::string o = b;
    if (nsec) {
      char f[16];
      snprintf(f, sizeof f, ".%09d", nsec);
      std::string fs = f;
      while (fs.back() == '0') fs.pop_back();
      o += fs;
    }
    return o + "Z";
  }
  static bool ParseRFC3339(const std::string& s, Time* out) {
    int y; unsigned m, d, hh, mm, ss;
    int used = 0;
    if (sscanf(s.c_str(), "%4d-%2u-%2uT%2u:%2u:%2u%n", &y, &m, &d, &hh, &mm, &ss, &used) != 6) return false;
    int32_t ns = 0;
    size_t p = (size_t)used;
    if (p < s.size() && s[p] == '.') {
      p++;
      int digits = 0;
      while (p < s.size() && s[p] >= '0' && s[p] <= '9') {
        if (digits < 9) { ns = ns * 10 + (s[p] - '0'); digits++; }
        p++;
      }
      while (digits++ < 9) ns *= 10;
    }
    if (p >= s.size() || s[p] != 'Z' || p + 1 != s.size()) return false;  // UTC only
    *out = Date(y, m, d, hh, mm, ss, ns);
    return true;
  }
  static bool ParseDate(const std::string& s, bool with_hour, Time* out) {  // time.Parse of the two layouts
    const size_t want = with_hour ? 13 : 10;
    if (s.size() != want) return false;
    for (size_t i = 0; i < want; i++) {
      const bool dash = i == 4 || i == 7 || i == 10;
      if (dash ? s[i] != '-' : (s[i] < '0' || s[i] > '9')) return false;
    }
    const int y = atoi(s.substr(0, 4).c_str());
    const unsigned m = (unsigned)atoi(s.substr(5, 2).c_str()), d = (unsigned)atoi(s.substr(8, 2).c_str());
    const unsigned h = with_hour ? (unsigned)atoi(s.substr(11, 2).c_str()) : 0;
    static const unsigned dim[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const bool leap = (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0);
    if (m < 1 || m > 12 || d < 1 || d > dim[m - 1] + (m == 2 && leap ? 1u : 0u) || h > 23) return false;
    *out = Date(y, m, d, h);
    return true;
  }
};

// ------------------------------------------------------------------------------------------ types.go
constexpr const char* kExpirationFormat = "2006-01-02";
constexpr const char* kExpirationFormatWithHour = "2006-01-02-15";

// SHA-256 provider for Issuer.ID() of an Issuer built from SPKI bytes: the GPU (GpuEngine installs
// ctmr_sha256).  There is no host fallback: without an engine the call throws.
using Sha256Fn = std::function<std::array<uint8_t, 32>(const std::string&)>;
inline Sha256Fn& sha256_provider() {
  static Sha256Fn f;
  return f;
}

class SPKI {  // types.go:142-159
 public:
  SPKI() = default;
  explicit SPKI(std::string bytes) : spki_(std::move(bytes)) {}
  std::string ID() const { return b64url_encode(spki_); }
  std::string String() const { return hex_encode(spki_); }
  std::string Sha256DigestURLEncodedBase64() const {
    if (!sha256_provider()) throw Error("no SHA-256 provider: construct a GpuEngine first (no CPU fallback)");
    const auto d = sha256_provider()(spki_);
    return b64url_encode(std::string((const char*)d.data(), 32));
  }
  const std::string& bytes() const { return spki_; }

 private:
  std::string spki_;
};

class Issuer {  // types.go:104-140
 public:
  Issuer() = default;
  static Issuer FromSPKI(std::string rawSubjectPublicKeyInfo) {  // NewIssuer(aCert)
    Issuer i;
    i.spki_ = SPKI(std::move(rawSubjectPublicKeyInfo));
    return i;
  }
  static Issuer FromString(std::string id) {  // NewIssuerFromString
    Issuer i;
    i.id_ = std::make_shared<std::string>(std::move(id));
    return i;
  }
  const std::string& ID() const {  // lazily memoised
    if (!id_) id_ = std::make_shared<std::string>(spki_.Sha256DigestURLEncodedBase64());
    return *id_;
  }
  bool idIsSet() const { return (bool)id_; }
  std::string MarshalJSON() const { return "\"" + ID() + "\""; }

 private:
  mutable std::shared_ptr<std::string> id_;
  SPKI spki_;
};

class Serial {  // types.go:161-244: the RAW INTEGER content octets
 public:
  Serial() = default;
  static Serial FromBytes(std::string b) { Serial s; s.serial_ = std::move(b); return s; }
  static Serial FromHex(const std::string& h) {  // NewSerialFromHex panics on bad input
    Serial s;
    if (!hex_decode(h, &s.serial_)) throw std::logic_error("encoding/hex: invalid byte");
    return s;
  }
  static Serial FromIDString(const std::string& id) { return FromBytes(b64url_decode(id)); }
  static Serial FromBinaryString(const std::string& b) { return FromBytes(b); }
  std::string ID() const { return b64url_encode(serial_); }
  std::string String() const { return HexString(); }
  std::string BinaryString() const { return serial_; }
  std::string HexString() const { return hex_encode(serial_); }
  int Cmp(const Serial& o) const { return serial_ < o.serial_ ? -1 : (serial_ == o.serial_ ? 0 : 1); }
  std::string MarshalJSON() const { return "\"" + HexString() + "\""; }
  static Serial UnmarshalJSON(const std::string& data) {
    if (data.size() < 2 || data.front() != '"' || data.back() != '"') throw Error("Expected surrounding quotes");
    Serial s;
    if (!hex_decode(data.substr(1, data.size() - 2), &s.serial_)) throw Error("encoding/hex: invalid byte");
    return s;
  }
  // big.Int.SetBytes, as a decimal string (tests compare small values)
  std::string AsBigIntDecimal() const {
    std::vector<uint8_t> digits{0};
    for (unsigned char c : serial_) {
      unsigned carry = c;
      for (auto& d : digits) { const unsigned v = d * 256u + carry; d = (uint8_t)(v % 10); carry = v / 10; }
      while (carry) { digits.push_back((uint8_t)(carry % 10)); carry /= 10; }
    }
    std::string o;
    for (auto it = digits.rbegin(); it != digits.rend(); ++it) o += (char)('0' + *it);
    return o;
  }
  bool operator==(const Serial& o) const { return serial_ == o.serial_; }
  bool operator<(const Serial& o) const { return serial_ < o.serial_; }

 private:
  std::string serial_;
};

class ExpDate {  // types.go:333-384
 public:
  ExpDate() = default;
  static ExpDate FromTime(const Time& t) {  // NewExpDateFromTime
    ExpDate e;
    e.date_ = t.TruncateHour();
    e.lastGood_ = e.date_.AddMillis(-1);
    e.hourResolution_ = true;
    return e;
  }
  static ExpDate FromHour(int32_t exp_hour) { return FromTime(Time::Unix((int64_t)exp_hour * 3600)); }  // device encoding
  static ExpDate Parse(const std::string& s) {  // NewExpDate
    ExpDate e;
    Time t;
    if (s.size() > 10 && Time::ParseDate(s, true, &t)) {
      e.date_ = t; e.lastGood_ = t.AddMillis(3600 * 1000 - 1); e.hourResolution_ = true;
      return e;
    }
    if (Time::ParseDate(s, false, &t)) {
      e.date_ = t; e.lastGood_ = t.AddMillis(24ll * 3600 * 1000 - 1); e.hourResolution_ = false;
      return e;
    }
    throw Error("parsing time \"" + s + "\" as \"" + kExpirationFormat + "\": cannot parse");
  }
  bool IsExpiredAt(const Time& t) const { return lastGood_.Before(t); }
  Time ExpireTime() const { return date_; }
  std::string ID() const { return date_.Format(hourResolution_); }
  std::string String() const { return ID(); }
  int32_t hour() const { return (int32_t)(date_.sec / 3600 - (date_.sec % 3600 < 0)); }
  bool operator<(const ExpDate& o) const { return date_.Before(o.date_); }

 private:
  Time date_, lastGood_;
  bool hourResolution_ = false;
};

struct UniqueCertIdentifier {  // types.go:279-315
  ExpDate expDate;
  Issuer issuer;
  Serial serialNum;
  static UniqueCertIdentifier Parse(const std::string& s) {
    std::vector<std::string> parts;
    size_t p = 0;
    for (;;) {
      const size_t q = s.find("::", p);
      parts.push_back(s.substr(p, q == std::string::npos ? std::string::npos : q - p));
      if (q == std::string::npos) break;
      p = q + 2;
    }
    if (parts.size() != 3) throw Error("Expected 3 parts, got " + std::to_string(parts.size()));
    return {ExpDate::Parse(parts[0]), Issuer::FromString(parts[1]), Serial::FromIDString(parts[2])};
  }
  std::string String() const { return expDate.ID() + "::" + issuer.ID() + "::" + serialNum.ID(); }
};

struct IssuerAndDate {  // types.go:317-340
  ExpDate expDate;
  Issuer issuer;
  static IssuerAndDate Parse(const std::string& s) {
    const size_t q = s.find('/');
    if (q == std::string::npos || s.find('/', q + 1) != std::string::npos)
      throw Error("Unexpected number of parts from " + s);
    return {ExpDate::Parse(s.substr(0, q)), Issuer::FromString(s.substr(q + 1))};
  }
  std::string String() const { return expDate.ID() + "/" + issuer.ID(); }
};

struct IssuerDate {  // types.go:402-405
  Issuer issuer;
  std::vector<ExpDate> expDates;
};

inline std::string CertificateLogIDFromShortURL(const std::string& shortURL) { return b64url_encode(shortURL); }

inline std::string json_escape(const std::string& s) {
  std::string o;
  for (unsigned char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
    else o += (char)c;
  }
  return o;
}

struct CertificateLog {  // types.go:25-44; JSON field names as encoding/json derives them
  std::string ShortURL;
  int64_t MaxEntry = 0;
  Time LastEntryTime, LastUpdateTime;
  std::string ID() const { return CertificateLogIDFromShortURL(ShortURL); }
  std::string MarshalJSON() const {
    return "{\"ShortURL\":\"" + json_escape(ShortURL) + "\",\"MaxEntry\":" + std::to_string(MaxEntry) +
           ",\"LastEntryTime\":\"" + LastEntryTime.RFC3339Nano() + "\",\"LastUpdateTime\":\"" +
           LastUpdateTime.RFC3339Nano() + "\"}";
  }
  static CertificateLog UnmarshalJSON(const std::string& j) {
    auto str_field = [&](const char* name) -> std::string {
      const std::string k = std::string("\"") + name + "\":\"";
      const size_t p = j.find(k);
      if (p == std::string::npos) throw Error(std::string("json: missing field ") + name);
      std::string o;
      for (size_t i = p + k.size(); i < j.size() && j[i] != '"'; i++) {
        if (j[i] == '\\' && i + 1 < j.size()) i++;
        o += j[i];
      }
      return o;
    };
    CertificateLog l;
    l.ShortURL = str_field("ShortURL");
    const size_t p = j.find("\"MaxEntry\":");
    if (p == std::string::npos) throw Error("json: missing field MaxEntry");
    l.MaxEntry = strtoll(j.c_str() + p + 11, nullptr, 10);
    if (!Time::ParseRFC3339(str_field("LastEntryTime"), &l.LastEntryTime) ||
        !Time::ParseRFC3339(str_field("LastUpdateTime"), &l.LastUpdateTime))
      throw Error("json: cannot parse time");
    return l;
  }
  bool operator==(const CertificateLog& o) const {
    return ShortURL == o.ShortURL && MaxEntry == o.MaxEntry && LastEntryTime.Equal(o.LastEntryTime) &&
           LastUpdateTime.Equal(o.LastUpdateTime);
  }
};

// ------------------------------------------------------------------------------------------ RemoteCache
// storage/types.go:83-102.  Set members and keys are byte strings (serials contain NULs).
class RemoteCache {
 public:
  virtual ~RemoteCache() = default;
  virtual bool Exists(const std::string& key) = 0;
  virtual bool SetInsert(const std::string& key, const std::string& entry) = 0;
  virtual bool SetRemove(const std::string& key, const std::string& entry) = 0;
  virtual bool SetContains(const std::string& key, const std::string& entry) = 0;
  virtual std::vector<std::string> SetList(const std::string& key) = 0;
  virtual void SetToChan(const std::string& key, const std::function<void(const std::string&)>& c) = 0;
  virtual int SetCardinality(const std::string& key) = 0;
  virtual void ExpireAt(const std::string& key, const Time& t) = 0;
  virtual void ExpireIn(const std::string& key, int64_t millis) = 0;
  virtual int64_t Queue(const std::string& key, const std::string& identifier) = 0;
  virtual std::string Pop(const std::string& key) = 0;
  virtual int64_t QueueLength(const std::string& key) = 0;
  virtual std::string BlockingPopCopy(const std::string& key, const std::string& dest, int64_t timeout_ms) = 0;
  virtual void ListRemove(const std::string& key, const std::string& value) = 0;
  virtual std::string TrySet(const std::string& k, const std::string& v, int64_t life_ms) = 0;
  virtual void KeysToChan(const std::string& pattern, const std::function<void(const std::string&)>& c) = 0;
  virtual void StoreLogState(const CertificateLog& log) = 0;
  virtual CertificateLog LoadLogState(const std::string& shortUrl) = 0;
};

inline Time now() { return Time::Unix((int64_t)::time(nullptr)); }

// filepath.Match (mockcache.go:152-166): '*', '?', '[class]', backslash escape; no separator in our keys matters
inline bool path_match(const std::string& pat, const std::string& s) {
  size_t pi = 0, si = 0, star_p = std::string::npos, star_s = 0;
  const size_t pn = pat.size(), sn = s.size();
  while (si < sn) {
    bool adv = false;
    if (pi < pn) {
      char c = pat[pi];
      if (c == '*') { star_p = pi++; star_s = si; continue; }
      if (c == '?') { if (s[si] != '/') { pi++; si++; adv = true; } }
      else if (c == '[') {
        size_t q = pi + 1;
        bool neg = false, hit = false;
        if (q < pn && pat[q] == '^') { neg = true; q++; }
        while (q < pn && pat[q] != ']') {
          char lo = pat[q];
          if (lo == '\\' && q + 1 < pn) lo = pat[++q];
          char hi = lo;
          if (q + 2 < pn && pat[q + 1] == '-' && pat[q + 2] != ']') { hi = pat[q + 2]; q += 2; }
          if (s[si] >= lo && s[si] <= hi) hit = true;
          q++;
        }
        if (hit != neg) { pi = q < pn ? q + 1 : q; si++; adv = true; }
      } else {
        if (c == '\\' && pi + 1 < pn) c = pat[++pi];
        if (c == s[si]) { pi++; si++; adv = true; }
      }
    }
    if (adv) continue;
    if (star_p == std::string::npos || s[star_s] == '/') return false;
    pi = star_p + 1;
    si = ++star_s;
  }
  while (pi < pn && pat[pi] == '*') pi++;
  return pi == pn;
}

class GpuEngine {
 public:
  // collect_meta: IssuerMetadata's memo maps live on the GPU (ctmr_meta_new) and StoreBatch only sees first sightings
  explicit GpuEngine(int device = 0, uint64_t table_slots = 0, uint64_t pair_slots = 0, uint32_t max_issuers = 0,
                     bool collect_meta = false)
      : collect_meta_(collect_meta) {
    ctmr_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.device = device;
    cfg.table_slots = table_slots;
    cfg.pair_slots = pair_slots;
    cfg.max_issuers = max_issuers;
    cfg.collect_meta = collect_meta ? 1 : 0;
    const int rc = ctmr_create(&cfg, &h_);
    if (rc != CTMR_OK) throw Error("ctmr_create failed (" + std::to_string(rc) + "): no usable HIP device — libctmr has no CPU fallback");
    ctmr_engine* h = h_;
    sha256_provider() = [h](const std::string& m) {
      std::array<uint8_t, 32> d{};
      if (ctmr_sha256(h, (const uint8_t*)m.data(), m.size(), d.data()) != CTMR_OK) throw Error(ctmr_last_error(h));
      return d;
    };
  }
  ~GpuEngine() {
    sha256_provider() = nullptr;
    ctmr_destroy(h_);
  }
  GpuEngine(const GpuEngine&) = delete;
  GpuEngine& operator=(const GpuEngine&) = delete;
  ctmr_engine* handle() const { return h_; }
  bool collect_meta() const { return collect_meta_; }
  void ck(int rc) const {
    if (rc != CTMR_OK) throw Error(std::string("libctmr: ") + ctmr_last_error(h_) + " (" + std::to_string(rc) + ")");
  }
  // *ctconfig.IssuerCNFilter, *ctconfig.LogExpiredEntries, time.Now() of certIsFilteredOut (ct-fetch.go:44-70)
  void SetFilter(const std::string& issuerCNFilter, bool logExpiredEntries, int64_t now_unix) {
    ck(ctmr_set_filter(h_, issuerCNFilter.data(), issuerCNFilter.size(), logExpiredEntries ? 1 : 0, now_unix));
  }
  // how raw-entry calls identify Chain[0] (ct-fetch.go:221): CTMR_CHAIN0_EXACT (default) | CTMR_CHAIN0_TRUSTED_LOG
  void SetChain0Match(int mode) { ck(ctmr_set_chain0_match(h_, mode)); }
  uint32_t AddIssuer(const std::string& chain0_der) {
    const uint64_t off[2] = {0, chain0_der.size()};
    uint32_t first = 0;
    ck(ctmr_add_issuers(h_, (const uint8_t*)chain0_der.data(), off, 1, &first));
    return first;
  }
  ctmr_issuer_info IssuerInfo(uint32_t idx) const {
    ctmr_issuer_info info;
    ck(ctmr_issuer_info_get(h_, idx, &info));
    return info;
  }
  std::vector<uint64_t> IssuerCounts() const {
    uint32_t n = 0;
    ck(ctmr_issuer_count(h_, &n));
    std::vector<uint64_t> v(n);
    if (n) ck(ctmr_issuer_counts(h_, v.data(), n));
    return v;
  }

 private:
  ctmr_engine* h_ = nullptr;
  bool collect_meta_ = false;
};

// storage.RemoteCache over libctmr: "serials::<expDate>::<issuerID>" sets live in HBM, every other key in the
// library's host-side store — the drop-in for RedisCache (storage/rediscache.go).
class GpuRemoteCache : public RemoteCache {
 public:
  explicit GpuRemoteCache(GpuEngine& e) : e_(e) {}
  GpuEngine& engine() { return e_; }
  bool Exists(const std::string& key) override {
    int r = 0;
    e_.ck(ctmr_exists(e_.handle(), key.data(), key.size(), &r));
    return r != 0;
  }
  bool SetInsert(const std::string& key, const std::string& entry) override {
    int r = 0;
    e_.ck(ctmr_set_insert(e_.handle(), key.data(), key.size(), (const uint8_t*)entry.data(), entry.size(), &r));
    return r != 0;
  }
  bool SetRemove(const std::string& key, const std::string& entry) override {
    int r = 0;
    e_.ck(ctmr_set_remove(e_.handle(), key.data(), key.size(), (const uint8_t*)entry.data(), entry.size(), &r));
    return r != 0;
  }
  bool SetContains(const std::string& key, const std::string& entry) override {
    int r = 0;
    e_.ck(ctmr_set_contains(e_.handle(), key.data(), key.size(), (const uint8_t*)entry.data(), entry.size(), &r));
    return r != 0;
  }
  std::vector<std::string> SetList(const std::string& key) override {
    return listing([&](uint8_t* o, size_t cap, size_t* need, uint64_t* cnt) {
      return ctmr_set_members(e_.handle(), key.data(), key.size(), o, cap, need, cnt);
    });
  }
  void SetToChan(const std::string& key, const std::function<void(const std::string&)>& c) override {
    for (auto& m : SetList(key)) c(m);
  }
  int SetCardinality(const std::string& key) override {
    int64_t n = 0;
    e_.ck(ctmr_set_cardinality(e_.handle(), key.data(), key.size(), &n));
    return (int)n;
  }
  void ExpireAt(const std::string& key, const Time& t) override {
    e_.ck(ctmr_expire_at(e_.handle(), key.data(), key.size(), t.sec));
  }
  void ExpireIn(const std::string& key, int64_t ms) override { ExpireAt(key, now().AddMillis(ms)); }
  // list / lock keys (rediscache.go:122-178): RPUSH/LPOP/LLEN/BRPOPLPUSH/LREM/SETNX kept host side
  int64_t Queue(const std::string& key, const std::string& id) override {
    lists_[key].push_back(id);
    return (int64_t)lists_[key].size();
  }
  std::string Pop(const std::string& key) override {
    auto& l = lists_[key];
    if (l.empty()) throw Error("redis: nil");
    std::string v = l.front();
    l.erase(l.begin());
    return v;
  }
  int64_t QueueLength(const std::string& key) override { return (int64_t)lists_[key].size(); }
  std::string BlockingPopCopy(const std::string& key, const std::string& dest, int64_t) override {
    auto& l = lists_[key];
    if (l.empty()) throw Error("redis: nil");
    std::string v = l.back();
    l.pop_back();
    auto& d = lists_[dest];
    d.insert(d.begin(), v);
    return v;
  }
  void ListRemove(const std::string& key, const std::string& value) override {
    auto& l = lists_[key];
    l.erase(std::remove(l.begin(), l.end(), value), l.end());
  }
  std::string TrySet(const std::string& k, const std::string& v, int64_t life_ms) override {
    const std::string key = "lock::" + k;
    auto cur = SetList(key);
    if (!cur.empty()) return cur[0];
    SetInsert(key, v);
    ExpireIn(key, life_ms);
    return v;
  }
  void KeysToChan(const std::string& pattern, const std::function<void(const std::string&)>& c) override {
    for (auto& k : listing([&](uint8_t* o, size_t cap, size_t* need, uint64_t* cnt) {
           return ctmr_keys(e_.handle(), pattern.data(), pattern.size(), o, cap, need, cnt);
         }))
      c(k);
  }
  void StoreLogState(const CertificateLog& log) override {  // rediscache.go:180-190: key "log::<shortURL>"
    const std::string key = "log::" + log.ShortURL;
    for (auto& old : SetList(key)) SetRemove(key, old);
    SetInsert(key, log.MarshalJSON());
  }
  CertificateLog LoadLogState(const std::string& shortUrl) override {
    auto d = SetList("log::" + shortUrl);
    if (d.empty()) throw Error("Log state not found");
    return CertificateLog::UnmarshalJSON(d[0]);
  }

 private:
  template <class F>
  std::vector<std::string> listing(F call) {
    size_t need = 0;
    uint64_t cnt = 0;
    int rc = call(nullptr, 0, &need, &cnt);
    if (rc != CTMR_OK && rc != CTMR_E_RANGE) e_.ck(rc);
    std::vector<uint8_t> buf(need ? need : 1);
    e_.ck(call(buf.data(), buf.size(), &need, &cnt));
    std::vector<std::string> out;
    size_t p = 0;
    for (uint64_t i = 0; i < cnt; i++) {
      uint32_t l;
      memcpy(&l, buf.data() + p, 4);
      out.emplace_back((const char*)buf.data() + p + 4, l);
      p += 4 + l;
    }
    return out;
  }
  GpuEngine& e_;
  std::map<std::string, std::vector<std::string>> lists_;
};

// ------------------------------------------------------------------------------------------ KnownCertificates
constexpr const char* kSerials = "serials";
constexpr const char* kCrls = "crl";
constexpr const char* kIssuers = "issuer";

class KnownCertificates {  // storage/knowncertificates.go
 public:
  KnownCertificates(ExpDate expDate, Issuer issuer, RemoteCache* cache)
      : expDate_(std::move(expDate)), issuer_(std::move(issuer)), cache_(cache) {}
  std::string id(const std::string& params = "") const { return expDate_.ID() + params + "::" + issuer_.ID(); }
  std::string serialId(const std::string& params = "") const { return std::string(kSerials) + "::" + id(params); }
  // true if this serial was unknown; subsequent calls with the same serial return false (:36-55)
  bool WasUnknown(const Serial& s) {
    const bool result = cache_->SetInsert(serialId(), s.BinaryString());
    if (!expirySet_) {
      cache_->ExpireAt(serialId(), expDate_.ExpireTime());
      expirySet_ = true;
    }
    return result;
  }
  int64_t Count() const { return cache_->SetCardinality(serialId()); }
  std::vector<Serial> Known() const {  // de-duplicates what SetToChan yields (:65-96)
    std::set<std::string> seen;
    cache_->SetToChan(serialId(), [&](const std::string& s) { seen.insert(s); });
    std::vector<Serial> out;
    for (auto& s : seen) out.push_back(Serial::FromBinaryString(s));
    return out;
  }

 private:
  ExpDate expDate_;
  Issuer issuer_;
  RemoteCache* cache_;
  bool expirySet_ = false;
};

// ------------------------------------------------------------------------------------------ host certificate view
// The fields the certWasUnknown branch needs from a NEWLY UNKNOWN certificate the GPU walk already accepted
// (issuermetadata.go:92-138): pkix.Name.String() of the issuer and CRLDistributionPoints; plus the raw serial.
class HostCert {
 public:
  explicit HostCert(const std::string& der) : d_(der) {
    size_t cs, ce, ts, te;
    uint8_t tag;
    tlv(0, d_.size(), &tag, &cs, &ce);
    tlv(cs, ce, &tag, &ts, &te);
    std::vector<std::array<size_t, 3>> kids = children(ts, te);
    size_t k = (!kids.empty() && kids[0][0] == 0xa0) ? 1 : 0;
    if (kids.size() < k + 6) throw Error("x509: malformed tbsCertificate");
    serial = d_.substr(kids[k][1], kids[k][2] - kids[k][1]);
    parseName(kids[k + 2][1], kids[k + 2][2]);
    for (size_t i = k + 6; i < kids.size(); i++) {
      if (kids[i][0] != 0xa3) continue;
      size_t es, ee;
      tlv(kids[i][1], kids[i][2], &tag, &es, &ee);
      for (auto& x : children(es, ee)) {
        auto parts = children(x[1], x[2]);
        if (parts.empty() || d_.compare(parts[0][1], parts[0][2] - parts[0][1], "\x55\x1d\x1f") != 0) continue;
        // asn1.Unmarshal(value, &[]distributionPoint) as the device walk restates it (csrc/der_walk.h crl_dps): the WHOLE
        // value must unmarshal — one SEQUENCE filling it, every distributionPoint a SEQUENCE whose three optional fields
        // are taken in order (a header must parse at each field's position unless the contents are used up; a field of
        // another tag is skipped unconsumed; a matching one must fit; Reason is a BIT STRING) — or parseCertificate fails
        // and there are NO distribution points (round 6, ADVICE r05: this mirror used to keep the URIs of a value whose
        // later part was malformed, where the device — under CTMR_PROFILE_FAST, which lets such a certificate through —
        // reports none).
        std::vector<std::string> uris;
        if (crlDps(parts.back()[1], parts.back()[2], &uris))
          for (auto& u : uris) crlDistributionPoints.push_back(u);
      }
    }
  }
  struct ATV { std::string oid; uint8_t tag; std::string value; size_t rdn; };
  // pkix.Name.String() of a bare Name TLV — the bytes a CTMR_MK_DN item carries (ctmr.h, N3)
  static std::string NameString(const std::string& name_tlv) {
    HostCert h(name_tlv, 0);
    return h.IssuerString();
  }
  std::string serial;
  std::vector<ATV> issuer_atvs;
  std::vector<std::string> crlDistributionPoints;

  // pkix.Name.String() of certificate-transparency-go v1.1.0 (go.mod:10) = ToRDNSequence().String(): the Name is
  // rebuilt from the nine typed fields FillFromRDNSequence fills (C, ST, L, STREET, POSTALCODE, O, OU, CN,
  // SERIALNUMBER; values of the types Go's asn1 decodes to a `string` only) plus ExtraNames, which a parsed certificate
  // never has — every other attribute type (DC, emailAddress, organizationIdentifier, jurisdiction*, …) is DROPPED.
  // (Go >= 1.15's crypto/x509/pkix appends them as `oid=#hex`; the pinned fork predates that.)  The sequence is
  // printed reversed, the values of one type joined by '+'.
  std::string IssuerString() const {
    static const std::map<uint8_t, const char*> names = {{6, "C"}, {10, "O"}, {11, "OU"}, {3, "CN"}, {5, "SERIALNUMBER"},
                                                         {7, "L"}, {8, "ST"}, {9, "STREET"}, {17, "POSTALCODE"}};
    static const uint8_t order[] = {6, 8, 7, 9, 17, 10, 11, 3, 5};
    std::map<uint8_t, std::vector<std::string>> named;
    std::vector<std::string> rdns;
    for (auto& a : issuer_atvs) {
      const bool go_string = a.tag == 0x0c || a.tag == 0x12 || a.tag == 0x13 || a.tag == 0x14 || a.tag == 0x16;
      if (a.oid.size() == 3 && a.oid[0] == 0x55 && a.oid[1] == 0x04 && names.count((uint8_t)a.oid[2]) && go_string)
        named[(uint8_t)a.oid[2]].push_back(a.value);
    }
    for (uint8_t t : order) {
      auto f = named.find(t);
      if (f == named.end()) continue;
      std::string r;
      for (size_t i = 0; i < f->second.size(); i++) {
        if (i) r += "+";
        r += std::string(names.at(t)) + "=" + escape(f->second[i]);
      }
      rdns.push_back(r);
    }
    std::string out;
    for (auto it = rdns.rbegin(); it != rdns.rend(); ++it) {
      if (it != rdns.rbegin()) out += ",";
      out += *it;
    }
    return out;
  }

 private:
  HostCert(const std::string& name_tlv, int) : d_(name_tlv) {
    uint8_t tag;
    size_t cs, ce;
    tlv(0, d_.size(), &tag, &cs, &ce);
    parseName(cs, ce);
  }
  void parseName(size_t s, size_t e) {
    for (auto& rdn : children(s, e))
      for (auto& atv : children(rdn[1], rdn[2])) {
        auto parts = children(atv[1], atv[2]);
        if (parts.size() < 2) throw Error("x509: malformed AttributeTypeAndValue");
        issuer_atvs.push_back({d_.substr(parts[0][1], parts[0][2] - parts[0][1]), (uint8_t)parts[1][0],
                               d_.substr(parts[1][1], parts[1][2] - parts[1][1]), rdn[1]});
      }
  }
  static std::string escape(const std::string& v) {
    std::string o;
    for (size_t k = 0; k < v.size(); k++) {
      const char c = v[k];
      const bool esc = strchr(",+\"\\<>;", c) != nullptr || (k == 0 && (c == ' ' || c == '#')) ||
                       (k + 1 == v.size() && c == ' ');
      if (esc && c != 0) o += '\\';
      o += c;
    }
    return o;
  }
  // Go encoding/asn1 parseTagAndLength at p inside [p, end): low tag numbers only (a high-tag-number element matches no
  // field here), definite minimal lengths.  fit: the contents must lie inside as well.  false = the header does not parse.
  bool goHdr(size_t p, size_t end, bool fit, uint8_t* tag, size_t* cs, size_t* ce) const {
    if (p + 2 > end) return false;
    *tag = (uint8_t)d_[p];
    if ((*tag & 0x1f) == 0x1f) return false;                      // (parses in Go, matches nothing below: treated as a failure
    const uint8_t b = (uint8_t)d_[p + 1];                         //  only where a field MUST match — see crlDps)
    size_t len, hl;
    if (b < 0x80) { len = b; hl = 2; }
    else {
      const size_t n = b & 0x7f;
      if (n == 0 || n > 4 || p + 2 + n > end || d_[p + 2] == 0) return false;
      len = 0;
      for (size_t i = 0; i < n; i++) len = (len << 8) | (uint8_t)d_[p + 2 + i];
      if (len < 0x80 || len > 0x7fffffff) return false;
      hl = 2 + n;
    }
    *cs = p + hl;
    *ce = p + hl + len;
    return !fit || *ce <= end;
  }
  bool bitStringOk(size_t c, size_t e) const {                      // parseBitString
    if (e == c) return false;
    const uint8_t pad = (uint8_t)d_[c];
    if (pad > 7 || (e - c == 1 && pad != 0)) return false;
    return pad == 0 || ((uint8_t)d_[e - 1] & ((1u << pad) - 1)) == 0;
  }
  bool crlDps(size_t ov, size_t oe, std::vector<std::string>* uris) const {
    uint8_t t, td, tf, tg, tn;
    size_t v, ve, p, pe, off, end, fc, fe, gc, ge, u, ue;
    v = ov; ve = oe;                                                   // [ov, oe) = the extnValue's contents
    if (!goHdr(v, ve, true, &t, &p, &pe) || t != 0x30 || pe != ve) return false;
    while (p < ve) {
      if (!goHdr(p, ve, true, &td, &off, &end) || td != 0x30) return false;
      tf = 0;
      if (off < end && !goHdr(off, end, false, &tf, &fc, &fe)) return false;
      if (off < end && tf == 0xa0) {                                  // DistributionPoint distributionPointName `optional,tag:0`
        if (fe > end) return false;
        size_t n = fc;
        tg = 0;
        if (n < fe && !goHdr(n, fe, false, &tg, &gc, &ge)) return false;
        if (n < fe && tg == 0xa0) {                                   // FullName []asn1.RawValue `optional,tag:0`
          if (ge > fe) return false;
          for (size_t q = gc; q < ge; q = ue) {
            if (!goHdrAny(q, ge, &tn, &u, &ue)) return false;
            if ((tn & 0x1f) == 6 && (tn & 0x1f) != 0x1f) uris->push_back(d_.substr(u, ue - u));   // Tag == 6: the NUMBER alone
          }
          n = ge;
          tg = 0;
          if (n < fe && !goHdr(n, fe, false, &tg, &gc, &ge)) return false;
        }
        if (n < fe && tg == 0xa1 && ge > fe) return false;            // RelativeName: must fit (its RDNs are the walk's business)
        off = fe;
        tf = 0;
        if (off < end && !goHdr(off, end, false, &tf, &fc, &fe)) return false;
      }
      if (off < end && tf == 0x81) {                                  // Reason asn1.BitString `optional,tag:1`
        if (fe > end || !bitStringOk(fc, fe)) return false;
        off = fe;
        tf = 0;
        if (off < end && !goHdr(off, end, false, &tf, &fc, &fe)) return false;
      }
      if (off < end && (tf == 0x82 || tf == 0xa2) && fe > end) return false;   // CRLIssuer asn1.RawValue `optional,tag:2`
      p = end;
    }
    return true;
  }
  // a RawValue element: any TLV that fits, the high-tag-number form included (its number is never 6)
  bool goHdrAny(size_t p, size_t end, uint8_t* tag, size_t* cs, size_t* ce) const {
    if (p >= end) return false;
    if (((uint8_t)d_[p] & 0x1f) != 0x1f) return goHdr(p, end, true, tag, cs, ce);
    size_t o = p + 1, k = 0;
    unsigned long long val = 0;
    for (;; k++) {
      if (o >= end || k == 5) return false;
      const uint8_t b = (uint8_t)d_[o++];
      if (k == 0 && b == 0x80) return false;
      val = (val << 7) | (b & 0x7f);
      if (!(b & 0x80)) break;
    }
    if (val > 0x7fffffffull || val < 0x1f) return false;
    // the length octets behind the tag: reuse goHdr on a header whose identifier octet is one low-tag octet earlier
    if (o >= end) return false;
    const uint8_t b = (uint8_t)d_[o];
    size_t len, hl;
    if (b < 0x80) { len = b; hl = 1; }
    else {
      const size_t n = b & 0x7f;
      if (n == 0 || n > 4 || o + 1 + n > end || d_[o + 1] == 0) return false;
      len = 0;
      for (size_t i = 0; i < n; i++) len = (len << 8) | (uint8_t)d_[o + 1 + i];
      if (len < 0x80 || len > 0x7fffffff) return false;
      hl = 1 + n;
    }
    *tag = 0x1f;
    *cs = o + hl;
    *ce = o + hl + len;
    return *ce <= end;
  }
  void tlv(size_t p, size_t end, uint8_t* tag, size_t* cs, size_t* ce) const {
    if (p + 2 > end) throw Error("asn1: truncated");
    *tag = (uint8_t)d_[p];
    const uint8_t b = (uint8_t)d_[p + 1];
    size_t len, hl;
    if (b < 0x80) { len = b; hl = 2; }
    else {
      const size_t n = b & 0x7f;
      if (n == 0 || n > 4 || p + 2 + n > end) throw Error("asn1: bad length");
      len = 0;
      for (size_t i = 0; i < n; i++) len = (len << 8) | (uint8_t)d_[p + 2 + i];
      hl = 2 + n;
    }
    if (p + hl + len > end) throw Error("asn1: length past end");
    *cs = p + hl;
    *ce = p + hl + len;
  }
  std::vector<std::array<size_t, 3>> children(size_t s, size_t e) const {
    std::vector<std::array<size_t, 3>> out;
    while (s < e) {
      uint8_t tag;
      size_t cs, ce;
      tlv(s, e, &tag, &cs, &ce);
      out.push_back({tag, cs, ce});
      s = ce;
    }
    return out;
  }
  const std::string& d_;
};

// ------------------------------------------------------------------------------------------ IssuerMetadata
// url.Parse(strings.TrimSpace(s)) + url.String() for the shapes a CRL distribution point takes: scheme
// (lower-cased) "://" authority path ["?" query] ["#" fragment]; control characters and malformed %-escapes
// are parse errors; path bytes outside Go's unescaped set are %XX-escaped by String().
inline bool go_url_normalise(const std::string& raw, std::string* scheme, std::string* out) {
  size_t a = 0, b = raw.size();
  auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  while (a < b && ws(raw[a])) a++;
  while (b > a && ws(raw[b - 1])) b--;
  const std::string s = raw.substr(a, b - a);
  for (unsigned char c : s)
    if (c < 0x20 || c == 0x7f) return false;  // "invalid control character in URL"
  size_t i = 0;
  scheme->clear();
  for (; i < s.size(); i++) {
    const char c = s[i];
    const bool alpha = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    if (alpha || (i > 0 && ((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.'))) continue;
    break;
  }
  if (i == 0 || i >= s.size() || s[i] != ':') { *out = s; return true; }  // no scheme
  for (size_t k = 0; k < i; k++) *scheme += (char)tolower((unsigned char)s[k]);
  std::string rest = s.substr(i + 1), frag, query;
  bool has_frag = false, has_query = false;
  size_t f = rest.find('#');
  if (f != std::string::npos) { frag = rest.substr(f + 1); rest.resize(f); has_frag = true; }
  size_t q = rest.find('?');
  if (q != std::string::npos) { query = rest.substr(q + 1); rest.resize(q); has_query = true; }
  auto bad_escape = [](const std::string& t) {
    for (size_t k = 0; k < t.size(); k++)
      if (t[k] == '%' && (k + 2 >= t.size() + 0 || !isxdigit((unsigned char)t[k + 1]) || !isxdigit((unsigned char)t[k + 2])))
        return true;
    return false;
  };
  if (bad_escape(rest) || bad_escape(frag)) return false;
  std::string authority, path = rest;
  if (rest.compare(0, 2, "//") == 0) {
    const size_t sl = rest.find('/', 2);
    authority = rest.substr(2, sl == std::string::npos ? std::string::npos : sl - 2);
    path = sl == std::string::npos ? "" : rest.substr(sl);
    const size_t at = authority.rfind('@');
    const std::string host = at == std::string::npos ? authority : authority.substr(at + 1);
    if (host.empty() || host[0] != '[') {  // validOptionalPort on what follows the last ':'
      const size_t colon = host.rfind(':');
      if (colon != std::string::npos)
        for (size_t k = colon + 1; k < host.size(); k++)
          if (host[k] < '0' || host[k] > '9') return false;
    }
  }
  auto esc = [](const std::string& t, bool is_path) {
    static const char H[] = "0123456789ABCDEF";
    std::string o;
    for (unsigned char c : t) {
      const bool keep = isalnum(c) || strchr("-_.~$&+,/:;=@%!*'()", c) != nullptr || (!is_path && c == '?');
      if (keep && c != 0) o += (char)c;
      else { o += '%'; o += H[c >> 4]; o += H[c & 15]; }
    }
    return o;
  };
  *out = *scheme + ":";
  if (rest.compare(0, 2, "//") == 0) *out += "//" + authority;
  *out += esc(path, true);
  if (has_query) *out += "?" + query;
  if (has_frag) *out += "#" + esc(frag, false);
  return true;
}

class IssuerMetadata {  // storage/issuermetadata.go
 public:
  IssuerMetadata(Issuer issuer, RemoteCache* cache) : issuer_(std::move(issuer)), cache_(cache) {}
  std::string id() const { return issuer_.ID(); }
  std::string crlId() const { return std::string(kCrls) + "::" + id(); }
  std::string issuersId() const { return std::string(kIssuers) + "::" + id(); }
  void addCRL(const std::string& aCRL) {  // :48-73
    std::string scheme, norm;
    if (!go_url_normalise(aCRL, &scheme, &norm)) return;  // "Not a valid CRL DP URL"
    if (scheme == "ldap" || scheme == "ldaps") return;
    if (scheme != "http" && scheme != "https") return;    // "Ignoring unknown CRL scheme"
    cache_->SetInsert(crlId(), norm);
  }
  void addIssuerDN(const std::string& dn) { cache_->SetInsert(issuersId(), dn); }  // :75-87
  // Must tolerate duplicate information.  Returns "this issuer's expiry date (hour) was seen before" (:92-138).
  bool Accumulate(const HostCert& cert, const ExpDate& expDate) {
    const std::string dn = cert.IssuerString();
    const bool seenExpDateBefore = knownExpDates_.count(expDate.ID()) != 0;
    const bool seenIssuerDn = knownIssuerDNs_.count(dn) != 0;
    if (!seenExpDateBefore) knownExpDates_.insert(expDate.ID());
    for (auto& dp : cert.crlDistributionPoints)
      if (knownCrlDPs_.insert(dp).second) addCRL(dp);
    if (!seenIssuerDn) {
      knownIssuerDNs_.insert(dn);
      addIssuerDN(dn);
    }
    return seenExpDateBefore;
  }
  // first sightings reported by the GPU memo (ctmr_meta_new): the same three steps of Accumulate, one at a time
  void firstExpDate(const ExpDate& expDate) { knownExpDates_.insert(expDate.ID()); }
  void firstCRL(const std::string& dp) { if (knownCrlDPs_.insert(dp).second) addCRL(dp); }
  void firstIssuerDN(const std::string& dn) { if (knownIssuerDNs_.insert(dn).second) addIssuerDN(dn); }
  std::vector<std::string> Issuers() const { return cache_->SetList(issuersId()); }
  std::vector<std::string> CRLs() const { return cache_->SetList(crlId()); }

 private:
  Issuer issuer_;
  RemoteCache* cache_;
  std::set<std::string> knownCrlDPs_, knownIssuerDNs_, knownExpDates_;
};

// ------------------------------------------------------------------------------------------ StorageBackend
class StorageBackend {  // storage/types.go:46-68 (ctx parameters dropped: no cancellation on this path)
 public:
  virtual ~StorageBackend() = default;
  virtual void MarkDirty(const std::string& id) = 0;
  virtual void StoreCertificatePEM(const Serial& serial, const ExpDate& expDate, const Issuer& issuer,
                                   const std::string& pem) = 0;
  virtual void StoreLogState(const CertificateLog& log) = 0;
  virtual void StoreKnownCertificateList(const Issuer& issuer, const std::vector<Serial>& serials) = 0;
  virtual std::string LoadCertificatePEM(const Serial& serial, const ExpDate& expDate, const Issuer& issuer) = 0;
  virtual CertificateLog LoadLogState(const std::string& logURL) = 0;
  virtual void AllocateExpDateAndIssuer(const ExpDate& expDate, const Issuer& issuer) = 0;
  virtual std::vector<ExpDate> ListExpirationDates(const Time& notBefore) = 0;
  virtual std::vector<Issuer> ListIssuersForExpirationDate(const ExpDate& expDate) = 0;
  virtual std::vector<Serial> ListSerialsForExpirationDateAndIssuer(const ExpDate& expDate, const Issuer& issuer) = 0;
  virtual void StreamSerialsForExpirationDateAndIssuer(const ExpDate& expDate, const Issuer& issuer,
                                                       const std::function<void(const UniqueCertIdentifier&)>& stream) = 0;
};

class NoopBackend : public StorageBackend {  // storage/noopbackend.go
 public:
  void MarkDirty(const std::string&) override {}
  void StoreCertificatePEM(const Serial&, const ExpDate&, const Issuer&, const std::string&) override {}
  void StoreLogState(const CertificateLog&) override {}
  void StoreKnownCertificateList(const Issuer&, const std::vector<Serial>&) override {}
  std::string LoadCertificatePEM(const Serial&, const ExpDate&, const Issuer&) override { throw err(); }
  CertificateLog LoadLogState(const std::string&) override { throw err(); }
  void AllocateExpDateAndIssuer(const ExpDate&, const Issuer&) override {}
  std::vector<ExpDate> ListExpirationDates(const Time&) override { throw err(); }
  std::vector<Issuer> ListIssuersForExpirationDate(const ExpDate&) override { throw err(); }
  std::vector<Serial> ListSerialsForExpirationDateAndIssuer(const ExpDate&, const Issuer&) override { throw err(); }
  void StreamSerialsForExpirationDateAndIssuer(const ExpDate&, const Issuer&,
                                               const std::function<void(const UniqueCertIdentifier&)>&) override { throw err(); }

 private:
  static Error err() { return Error("Unable to load from the NoopBackend."); }
};

class LocalDiskBackend : public StorageBackend {
 public:
  static constexpr const char* kStateDirName = "state";
  static constexpr const char* kDirtyMarker = "dirty";
  LocalDiskBackend(mode_t perms, std::string path) : perms_(perms), rootPath_(std::move(path)) {}
  void MarkDirty(const std::string& id) override { store_(join(id, kDirtyMarker), std::string(1, '\0')); }
  void AllocateExpDateAndIssuer(const ExpDate& e, const Issuer& i) override {
    mkdirs(dirname(join(join(rootPath_, e.ID()), i.ID())));  // makeDirectoryIfNotExist splits off the last element
  }
  void StoreCertificatePEM(const Serial& s, const ExpDate& e, const Issuer& i, const std::string& pem) override {
    AllocateExpDateAndIssuer(e, i);
    store_(join(join(join(rootPath_, e.ID()), i.ID()), s.ID()), pem);
  }
  void StoreLogState(const CertificateLog& log) override {
    store_(join(join(rootPath_, kStateDirName), log.ID()), log.MarshalJSON());
  }
  void StoreKnownCertificateList(const Issuer& i, const std::vector<Serial>& serials) override {
    std::string body;
    for (auto& s : serials) body += s.HexString() + "\n";
    store_(join(rootPath_, i.ID()), body);
  }
  std::string LoadCertificatePEM(const Serial&, const ExpDate&, const Issuer&) override { throw Error("Unimplemented"); }
  CertificateLog LoadLogState(const std::string& logURL) override {
    const std::string path = join(join(rootPath_, kStateDirName), CertificateLogIDFromShortURL(logURL));
    std::string body;
    if (!slurp(path, &body)) {  // a log we have not seen: fresh state (:244-252)
      CertificateLog l;
      l.ShortURL = logURL;
      return l;
    }
    return CertificateLog::UnmarshalJSON(body);
  }
  std::vector<ExpDate> ListExpirationDates(const Time&) override { throw Error("Unimplemented"); }
  std::vector<Issuer> ListIssuersForExpirationDate(const ExpDate&) override { throw Error("Unimplemented"); }
  std::vector<Serial> ListSerialsForExpirationDateAndIssuer(const ExpDate&, const Issuer&) override { throw Error("Unimplemented"); }
  void StreamSerialsForExpirationDateAndIssuer(const ExpDate&, const Issuer&,
                                               const std::function<void(const UniqueCertIdentifier&)>&) override { throw Error("Unimplemented"); }
  static bool slurp(const std::string& path, std::string* out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    out->clear();
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    fclose(f);
    return true;
  }

 private:
  static std::string join(const std::string& a, const std::string& b) {
    if (a.empty()) return b;
    return a.back() == '/' ? a + b : a + "/" + b;
  }
  static std::string dirname(const std::string& p) {
    const size_t q = p.rfind('/');
    return q == std::string::npos ? "" : p.substr(0, q);
  }
  static void mkdirs(const std::string& d) {
    if (d.empty()) return;
    for (size_t i = 1; i <= d.size(); i++)
      if (i == d.size() || d[i] == '/') {
        const std::string part = d.substr(0, i);
        if (mkdir(part.c_str(), 0777) != 0 && errno != EEXIST) throw Error("mkdir " + part + ": " + strerror(errno));
      }
  }
  void store_(const std::string& path, const std::string& data) {
    mkdirs(dirname(path));
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, perms_);
    if (fd < 0) throw Error("open " + path + ": " + strerror(errno));
    size_t w = 0;
    while (w < data.size()) {
      const ssize_t r = write(fd, data.data() + w, data.size() - w);
      if (r < 0) { close(fd); throw Error("write " + path + ": " + strerror(errno)); }
      w += (size_t)r;
    }
    close(fd);
  }
  mode_t perms_;
  std::string rootPath_;
};

// ------------------------------------------------------------------------------------------ CtLogEntry / batch result
// What insertCTWorker takes from one ct.LogEntry (ct-fetch.go:77-80,191-225): the DER of the certificate it
// parses (the X509 leaf, or Precert.Submitted.Data) and Chain[0].Data if the chain is not empty.
struct CtLogEntry {
  std::string leaf_der;
  std::string chain0_der;
  bool has_chain = true;   // len(ep.LogEntry.Chain) >= 1
  bool precert = false;    // ct.PrecertLogEntryType
  int64_t index = 0;       // ep.LogEntry.Index
  std::string logURL;
};

struct BatchResult {
  std::vector<ctmr_record> records;
  std::vector<uint64_t> new_idx;
  std::vector<uint64_t> timestamps;  // raw-entry batches: TimestampedEntry.Timestamp (ms)
  ctmr_batch_stats stats{};
  ctmr_decode_stats decode{};
};

// Raw get-entries batch (ctmr.h, N2): blob = leaf_input_0 ‖ extra_data_0 ‖ leaf_input_1 ‖ …, bounds[2n+1].
// What the downloader holds before ct.LogEntryFromLeaf (ct-fetch.go:446-452), base64 already decoded.
struct RawEntries {
  std::string blob;
  std::vector<uint64_t> bounds{0};
  uint64_t size() const { return (bounds.size() - 1) / 2; }
  void Append(const std::string& leaf_input, const std::string& extra_data) {
    blob += leaf_input;
    bounds.push_back(blob.size());
    blob += extra_data;
    bounds.push_back(blob.size());
  }
  std::string LeafInput(uint64_t i) const { return blob.substr(bounds[2 * i], bounds[2 * i + 1] - bounds[2 * i]); }
  std::string ExtraData(uint64_t i) const { return blob.substr(bounds[2 * i + 1], bounds[2 * i + 2] - bounds[2 * i + 1]); }
  // the certificate insertCTWorker parses (ct-fetch.go:198-204) of an entry the GPU decode accepted
  std::string Certificate(uint64_t i) const {
    const std::string leaf = LeafInput(i);
    auto be = [](const std::string& s, size_t p, int k) { size_t v = 0; for (int j = 0; j < k; j++) v = (v << 8) | (uint8_t)s[p + j]; return v; };
    if (be(leaf, 10, 2) == 0) return leaf.substr(15, be(leaf, 12, 3));
    const std::string extra = ExtraData(i);
    return extra.substr(3, be(extra, 0, 3));
  }
};

// ------------------------------------------------------------------------------------------ FilesystemDatabase
class FilesystemDatabase {  // storage/filesystemdatabase.go
 public:
  // aCache may be any RemoteCache; the batched GPU path needs aEngine (normally the engine behind a GpuRemoteCache)
  FilesystemDatabase(StorageBackend* aBackend, RemoteCache* aExtCache, GpuEngine* aEngine = nullptr)
      : backend_(aBackend), extCache_(aExtCache), engine_(aEngine) {}

  IssuerMetadata* GetIssuerMetadata(const Issuer& aIssuer) {  // :40-57
    auto f = meta_.find(aIssuer.ID());
    if (f == meta_.end()) f = meta_.emplace(aIssuer.ID(), std::make_unique<IssuerMetadata>(aIssuer, extCache_)).first;
    return f->second.get();
  }
  KnownCertificates GetKnownCertificates(const ExpDate& aExpDate, const Issuer& aIssuer) {  // :213-240
    return KnownCertificates(aExpDate, aIssuer, extCache_);
  }
  std::vector<IssuerDate> GetIssuerAndDatesFromCache() {  // :59-100
    std::map<std::string, IssuerDate> issuerMap;
    std::vector<std::string> order;
    extCache_->KeysToChan(std::string(kSerials) + "::*", [&](const std::string& entry) {
      std::vector<std::string> parts;
      size_t p = 0;
      for (;;) {
        const size_t q = entry.find("::", p);
        parts.push_back(entry.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 2;
      }
      if (parts.size() != 3) throw Error("Unexpected key format: " + entry);
      ExpDate e;
      try { e = ExpDate::Parse(parts[1]); } catch (const Error&) { return; }  // "Couldn't parse expiration date"
      auto f = issuerMap.find(parts[2]);
      if (f == issuerMap.end()) {
        f = issuerMap.emplace(parts[2], IssuerDate{Issuer::FromString(parts[2]), {}}).first;
        order.push_back(parts[2]);
      }
      f->second.expDates.push_back(e);
    });
    std::vector<IssuerDate> out;
    for (auto& k : order) out.push_back(issuerMap[k]);
    return out;
  }
  std::vector<ExpDate> ListExpirationDates(const Time& aNotBefore) { return backend_->ListExpirationDates(aNotBefore); }
  std::vector<Issuer> ListIssuersForExpirationDate(const ExpDate& e) { return backend_->ListIssuersForExpirationDate(e); }
  void SaveLogState(const CertificateLog& aLogObj) {  // :110-118
    try { extCache_->StoreLogState(aLogObj); } catch (const Error&) {}
    backend_->StoreLogState(aLogObj);
  }
  CertificateLog GetLogState(const std::string& host, const std::string& path) {  // :120-139 (url.Host + url.Path)
    const std::string shortUrl = host + path;
    try { return extCache_->LoadLogState(shortUrl); } catch (const Error&) {}
    try { return backend_->LoadLogState(shortUrl); } catch (const Error&) {}
    CertificateLog l;
    l.ShortURL = shortUrl;
    return l;
  }
  void markDirty(const Time& aExpiration) { backend_->MarkDirty(aExpiration.Format(false)); }  // :141-144
  void Cleanup() {}

  // The batched insertCTWorker → Store (ct-fetch.go:191-235 + filesystemdatabase.go:158-211): one GPU batch, then
  // for the NEWLY UNKNOWN certificates exactly what Store does after WasUnknown — IssuerMetadata.Accumulate,
  // AllocateExpDateAndIssuer when the (issuer, expDate) is new to this process, StoreCertificatePEM (PEM encoded
  // on the GPU) — and markDirty for every entry that reached Store.  Entries are processed in order; within one
  // batch the lowest log index of a new key is the one that "was unknown" (the reference with numThreads = 1).
  BatchResult StoreBatch(const std::vector<CtLogEntry>& entries) {
    if (!engine_) throw Error("StoreBatch needs a GpuEngine (no CPU fallback)");
    const uint64_t n = entries.size();
    BatchResult res;
    if (n == 0) return res;
    std::vector<uint64_t> offsets(n + 1, 0);
    std::vector<uint32_t> issuer_idx(n);
    std::vector<uint8_t> entry_type(n);
    std::string payload;
    for (uint64_t i = 0; i < n; i++) {
      const CtLogEntry& en = entries[i];
      payload += en.leaf_der;
      offsets[i + 1] = payload.size();
      entry_type[i] = en.precert ? 1 : 0;
      issuer_idx[i] = en.has_chain ? issuerIndex(en.chain0_der) : CTMR_NO_ISSUER;
    }
    payload.append(CTMR_PAYLOAD_PAD + 16, '\0');
    res.records.resize(n);
    res.new_idx.resize(n);
    engine_->ck(ctmr_map_batch(engine_->handle(), (const uint8_t*)payload.data(), offsets.data(), issuer_idx.data(),
                               entry_type.data(), n, res.records.data(), res.new_idx.data(), &res.stats));
    res.new_idx.resize(res.stats.n_new);
    afterMap(res, [&](uint64_t i) { return entries[i].leaf_der; });
    return res;
  }
  // The same from raw get-entries buffers (N2): ct.LogEntryFromLeaf, the choice of certificate and Chain[0] and the
  // issuer registration happen on the GPU (ctmr_map_entries); entries LogEntryFromLeaf rejects get
  // CTMR_ST_ENTRY_DECODE_ERROR (the downloader would have dropped them, ct-fetch.go:452-459).
  BatchResult StoreRawBatch(const RawEntries& raw) {
    if (!engine_) throw Error("StoreRawBatch needs a GpuEngine (no CPU fallback)");
    const uint64_t n = raw.size();
    BatchResult res;
    if (n == 0) return res;
    res.records.resize(n);
    res.new_idx.resize(n);
    res.timestamps.resize(n);
    // host variant: the library stages the blob itself and pads the device copy — no copy, no padding here
    engine_->ck(ctmr_map_entries(engine_->handle(), (const uint8_t*)raw.blob.data(), raw.bounds.data(), n, res.records.data(),
                                 res.new_idx.data(), res.timestamps.data(), &res.decode, &res.stats));
    res.new_idx.resize(res.stats.n_new);
    afterMap(res, [&](uint64_t i) { return raw.Certificate(i); });
    return res;
  }

 private:
  // FilesystemDatabase.Store after WasUnknown (filesystemdatabase.go:183-205) for a whole batch.  cert_of(i) is only
  // called for the certificates the host has to parse itself.
  template <class F>
  void afterMap(BatchResult& res, F cert_of) {
    const uint64_t n = res.records.size();
    // PEM blocks of the newly unknown certificates, encoded on the GPU
    size_t need = 0;
    uint64_t count = 0;
    int rc = ctmr_pem_new(engine_->handle(), nullptr, 0, nullptr, &need, &count);
    if (rc != CTMR_OK && rc != CTMR_E_RANGE) engine_->ck(rc);
    std::string pems(need, '\0');
    std::vector<uint64_t> pem_off(count + 1, 0);
    if (count) engine_->ck(ctmr_pem_new(engine_->handle(), (uint8_t*)&pems[0], pems.size(), pem_off.data(), &need, &count));
    if (count != res.new_idx.size()) throw Error("ctmr_pem_new: count mismatch");
    std::map<uint32_t, Issuer> issuers;
    auto issuer_of = [&](uint32_t idx) -> const Issuer& {
      auto f = issuers.find(idx);
      if (f == issuers.end()) f = issuers.emplace(idx, Issuer::FromString(engine_->IssuerInfo(idx).issuer_id)).first;
      return f->second;
    };
    auto serial_of = [&](uint64_t i) {
      const ctmr_record& r = res.records[i];
      if (r.serial_len <= 20) return Serial::FromBytes(std::string((const char*)r.serial, r.serial_len));
      const std::string der = cert_of(i);
      return Serial::FromBytes(HostCert(der).serial);
    };
    if (engine_->collect_meta()) {
      // N3: IssuerMetadata's memo maps live on the GPU; only first sightings come back (issuermetadata.go:92-138)
      uint64_t ni = 0;
      size_t nb = 0;
      rc = ctmr_meta_new(engine_->handle(), nullptr, 0, nullptr, 0, &ni, &nb);
      if (rc != CTMR_OK && rc != CTMR_E_RANGE) engine_->ck(rc);
      std::vector<ctmr_meta_item> items(ni);
      std::string bytes(nb, '\0');
      if (ni) engine_->ck(ctmr_meta_new(engine_->handle(), items.data(), ni, (uint8_t*)&bytes[0], nb, &ni, &nb));
      size_t at = 0;
      for (const ctmr_meta_item& it : items) {
        const Issuer& issuer = issuer_of(it.issuer_idx);
        IssuerMetadata* md = GetIssuerMetadata(issuer);
        const ExpDate expDate = ExpDate::FromHour(it.exp_hour);
        const std::string b = bytes.substr(at, it.len);
        at += it.len;
        if (it.kind == CTMR_MK_EXPDATE) {          // seenExpDateBefore == false (filesystemdatabase.go:189-195)
          md->firstExpDate(expDate);
          backend_->AllocateExpDateAndIssuer(expDate, issuer);
        } else if (it.kind == CTMR_MK_CRL) {
          md->firstCRL(b);
        } else if (it.kind == CTMR_MK_DN) {
          md->firstIssuerDN(HostCert::NameString(b));
        } else {                                   // CTMR_MK_HOST: the reference's own per-certificate route
          const std::string der = cert_of(it.entry);
          if (!md->Accumulate(HostCert(der), expDate)) backend_->AllocateExpDateAndIssuer(expDate, issuer);
        }
      }
      for (uint64_t k = 0; k < count; k++) {
        const uint64_t i = res.new_idx[k];
        const ctmr_record& r = res.records[i];
        backend_->StoreCertificatePEM(serial_of(i), ExpDate::FromHour(r.exp_hour), issuer_of(r.issuer_idx),
                                      pems.substr(pem_off[k], pem_off[k + 1] - pem_off[k]));
      }
    } else {
      for (uint64_t k = 0; k < count; k++) {  // certWasUnknown branch, filesystemdatabase.go:183-201, per certificate
        const uint64_t i = res.new_idx[k];
        const ctmr_record& r = res.records[i];
        const Issuer& issuer = issuer_of(r.issuer_idx);
        const ExpDate expDate = ExpDate::FromHour(r.exp_hour);
        const std::string der = cert_of(i);
        const HostCert cert(der);
        const bool issuerDateSeenBefore = GetIssuerMetadata(issuer)->Accumulate(cert, expDate);
        if (!issuerDateSeenBefore) backend_->AllocateExpDateAndIssuer(expDate, issuer);
        backend_->StoreCertificatePEM(Serial::FromBytes(cert.serial), expDate, issuer,
                                      pems.substr(pem_off[k], pem_off[k + 1] - pem_off[k]));
      }
    }
    std::set<int32_t> days;  // :204-208 — every entry that reached Store marks its day dirty (the marker is idempotent)
    for (uint64_t i = 0; i < n; i++)
      if (res.records[i].status == CTMR_ST_PASS) {
        const int32_t h = res.records[i].exp_hour;
        days.insert(h >= 0 ? h / 24 : -((-h + 23) / 24));
      }
    for (int32_t d : days) markDirty(Time::Unix((int64_t)d * 86400));
  }

 public:
  // Store(aCert, aIssuer, aLogURL, aEntryId) for one entry, DER in (types.go:74-75)
  BatchResult Store(const std::string& cert_der, const std::string& issuer_der, const std::string& aLogURL = "",
                    int64_t aEntryId = 0) {
    CtLogEntry e;
    e.leaf_der = cert_der;
    e.chain0_der = issuer_der;
    e.logURL = aLogURL;
    e.index = aEntryId;
    return StoreBatch({e});
  }

 private:
  uint32_t issuerIndex(const std::string& chain0_der) {  // exact: keyed by the full DER bytes
    auto f = issuerIdx_.find(chain0_der);
    if (f != issuerIdx_.end()) return f->second;
    const uint32_t idx = engine_->AddIssuer(chain0_der);
    issuerIdx_.emplace(chain0_der, idx);
    return idx;
  }
  StorageBackend* backend_;
  RemoteCache* extCache_;
  GpuEngine* engine_;
  std::map<std::string, std::unique_ptr<IssuerMetadata>> meta_;
  std::unordered_map<std::string, uint32_t> issuerIdx_;
};

// ------------------------------------------------------------------------------------------ the entryChan consumer
// LogSyncEngine.insertCTWorker (ct-fetch.go:180-246) with the per-entry body replaced by batches: Push() is the
// `for ep := range ld.entryChan` receive, Flush() the GPU batch.  Counters carry the reference's metric names.
class BatchInserter {
 public:
  BatchInserter(FilesystemDatabase* db, size_t batch_entries = 16384) : db_(db), cap_(batch_entries) {}  // entryChan cap :132
  void Push(CtLogEntry e) {
    pending_.push_back(std::move(e));
    if (pending_.size() >= cap_) Flush();
  }
  void Flush() {
    if (pending_.empty()) return;
    const BatchResult r = db_->StoreBatch(pending_);
    counters["certIsFilteredOut.CA"] += r.stats.by_status[CTMR_ST_FILTERED_CA];
    counters["certIsFilteredOut.expired"] += r.stats.by_status[CTMR_ST_FILTERED_EXPIRED];
    counters["certIsFilteredOut.cn-filtered"] += r.stats.by_status[CTMR_ST_FILTERED_CN];
    counters["insertCTWorker.Inserted"] += r.stats.by_status[CTMR_ST_PASS];  // counts duplicates too (:235)
    counters["insertCTWorker.ProblemDecodingCertificate"] += r.stats.by_status[CTMR_ST_PARSE_ERROR];
    counters["insertCTWorker.NoIssuerKnown"] += r.stats.by_status[CTMR_ST_NO_ISSUER];
    counters["insertCTWorker.ProblemDecodingIssuingCertificate"] += r.stats.by_status[CTMR_ST_ISSUER_PARSE_ERROR];
    counters["WasUnknown"] += r.stats.n_new;
    pending_.clear();
  }
  std::map<std::string, uint64_t> counters;

 private:
  FilesystemDatabase* db_;
  size_t cap_;
  std::vector<CtLogEntry> pending_;
};

// ------------------------------------------------------------------------------------------ storage-statistics
struct IssuerStatistics {
  std::string issuerID;
  size_t expDates = 0;
  int64_t serials = 0;
  std::vector<std::string> crls, dns;
};
struct StorageStatistics {  // cmd/storage-statistics/storage-statistics.go:28-82
  std::vector<IssuerStatistics> issuers;
  int64_t totalSerials = 0;
  size_t totalCRLs = 0;
  static StorageStatistics Collect(FilesystemDatabase* db) {
    StorageStatistics s;
    for (auto& issuerObj : db->GetIssuerAndDatesFromCache()) {
      IssuerStatistics is;
      is.issuerID = issuerObj.issuer.ID();
      is.expDates = issuerObj.expDates.size();
      IssuerMetadata* md = db->GetIssuerMetadata(issuerObj.issuer);
      is.crls = md->CRLs();
      is.dns = md->Issuers();
      for (auto& e : issuerObj.expDates) is.serials += db->GetKnownCertificates(e, issuerObj.issuer).Count();  // :44-53
      s.totalSerials += is.serials;
      s.totalCRLs += is.crls.size();
      s.issuers.push_back(std::move(is));
    }
    return s;
  }
};

}  // namespace storage
}  // namespace ctmr
