// ctmr_storage_mocks.hpp — the reference's TEST DOUBLES (storage/mockcache.go, storage/mockbackend.go) restated for the
// C++ host mirror's test suites (tests/host/test_storage.cpp).  Test scaffolding: they lived in the public header
// include/ctmr_storage.hpp until round 4 and have no business in what a host links against.
#pragma once
#include "../../include/ctmr_storage.hpp"

namespace ctmr {
namespace storage {

class MockRemoteCache : public RemoteCache {  // storage/mockcache.go (sorted slices = sets)
 public:
  std::map<std::string, std::vector<std::string>> Data;
  std::map<std::string, Time> Expirations;
  int Duplicate = 0;

  void CleanupExpiry() {
    const Time n = now();
    for (auto it = Expirations.begin(); it != Expirations.end();) {
      if (it->second.Before(n)) { Data.erase(it->first); it = Expirations.erase(it); }
      else ++it;
    }
  }
  bool SetInsert(const std::string& key, const std::string& entry) override {
    auto& v = Data[key];
    auto it = std::lower_bound(v.begin(), v.end(), entry);
    if (it != v.end() && *it == entry) return false;
    v.insert(it, entry);
    return true;
  }
  bool SetRemove(const std::string& key, const std::string& entry) override {
    CleanupExpiry();
    auto f = Data.find(key);
    if (f == Data.end()) return false;
    auto it = std::lower_bound(f->second.begin(), f->second.end(), entry);
    if (it == f->second.end() || *it != entry) return false;
    f->second.erase(it);  // (the reference's append(s[:idx], s[idx:]...) is a no-op bug; sets need the removal)
    return true;
  }
  bool SetContains(const std::string& key, const std::string& entry) override {
    CleanupExpiry();
    auto f = Data.find(key);
    return f != Data.end() && std::binary_search(f->second.begin(), f->second.end(), entry);
  }
  std::vector<std::string> SetList(const std::string& key) override {
    CleanupExpiry();
    auto f = Data.find(key);
    return f == Data.end() ? std::vector<std::string>{} : f->second;
  }
  void SetToChan(const std::string& key, const std::function<void(const std::string&)>& c) override {
    CleanupExpiry();
    auto f = Data.find(key);
    if (f == Data.end()) return;
    for (int i = 0; i < Duplicate + 1; i++)
      for (auto& v : f->second) c(v);
  }
  int SetCardinality(const std::string& key) override {
    auto f = Data.find(key);
    return f == Data.end() ? 0 : (int)f->second.size();
  }
  bool Exists(const std::string& key) override { CleanupExpiry(); return Data.count(key) != 0; }
  void ExpireAt(const std::string& key, const Time& t) override { Expirations[key] = t; }
  void ExpireIn(const std::string& key, int64_t ms) override { Expirations[key] = now().AddMillis(ms); }
  int64_t Queue(const std::string&, const std::string&) override { throw Error("Queue unimplemented"); }
  std::string Pop(const std::string&) override { throw Error("Pop unimplemented"); }
  int64_t QueueLength(const std::string&) override { throw Error("QueueLength unimplemented"); }
  std::string BlockingPopCopy(const std::string& key, const std::string& dest, int64_t) override {
    const std::string v = Pop(key);
    Queue(dest, v);
    return v;
  }
  void ListRemove(const std::string& key, const std::string& value) override { SetRemove(key, value); }
  std::string TrySet(const std::string& key, const std::string& v, int64_t life_ms) override {
    auto f = Data.find(key);
    if (f != Data.end() && !f->second.empty()) return f->second[0];
    Data[key] = {v};
    ExpireAt(key, now().AddMillis(life_ms));
    return v;
  }
  void KeysToChan(const std::string& pattern, const std::function<void(const std::string&)>& c) override {
    for (auto& kv : Data)
      if (path_match(pattern, kv.first)) c(kv.first);
  }
  void StoreLogState(const CertificateLog& log) override { Data[log.ShortURL] = {log.MarshalJSON()}; }
  CertificateLog LoadLogState(const std::string& shortUrl) override {
    auto f = Data.find(shortUrl);
    if (f == Data.end()) throw Error("Log state not found");
    if (f->second.size() != 1) throw Error("Unexpected number of log states");
    return CertificateLog::UnmarshalJSON(f->second[0]);
  }
};

// ------------------------------------------------------------------------------------------ the GPU engine
// RAII handle on a libctmr engine.  Fails loudly without a usable MI355X: there is no CPU fallback.
class MockBackend : public StorageBackend {  // storage/mockbackend.go
 public:
  std::map<std::string, std::vector<Issuer>> expDateToIssuer;
  std::map<std::string, std::vector<Serial>> expDateIssuerIDToSerials;
  std::map<std::string, std::string> store;
  std::vector<std::string> dirty;
  std::vector<std::string> allocations;  // call log: "<expDateID>/<issuerID>"

  void MarkDirty(const std::string& id) override { dirty.push_back(id); }
  void AllocateExpDateAndIssuer(const ExpDate& e, const Issuer& i) override {
    allocations.push_back(e.ID() + "/" + i.ID());
    auto& l = expDateToIssuer[e.ID()];
    for (auto& x : l)
      if (x.ID() == i.ID()) return;
    l.push_back(i);
    std::sort(l.begin(), l.end(), [](const Issuer& a, const Issuer& b) { return a.ID() < b.ID(); });
  }
  void StoreCertificatePEM(const Serial& s, const ExpDate& e, const Issuer& i, const std::string& pem) override {
    store["pem" + e.ID() + i.ID() + s.ID()] = pem;
    expDateIssuerIDToSerials[e.ID() + i.ID()].push_back(s);
  }
  void StoreLogState(const CertificateLog& log) override { store["logstate" + log.ShortURL] = log.MarshalJSON(); }
  void StoreKnownCertificateList(const Issuer& i, const std::vector<Serial>& serials) override {
    std::string j = "[";
    for (size_t k = 0; k < serials.size(); k++) j += (k ? "," : "") + serials[k].MarshalJSON();
    store[i.ID()] = j + "]";
  }
  std::string LoadCertificatePEM(const Serial& s, const ExpDate& e, const Issuer& i) override {
    auto f = store.find("pem" + e.ID() + i.ID() + s.ID());
    if (f == store.end()) throw Error("Couldn't find");
    return f->second;
  }
  CertificateLog LoadLogState(const std::string& logURL) override {
    auto f = store.find("logstate" + logURL);
    if (f != store.end()) return CertificateLog::UnmarshalJSON(f->second);
    CertificateLog l;
    l.ShortURL = logURL;
    return l;
  }
  std::vector<ExpDate> ListExpirationDates(const Time& notBefore) override {
    const Time day = Time::Unix((notBefore.sec / 86400 - (notBefore.sec % 86400 < 0)) * 86400);
    std::vector<ExpDate> out;
    for (auto& kv : expDateToIssuer) {
      const ExpDate e = ExpDate::Parse(kv.first);
      Time t;
      Time::ParseDate(kv.first.substr(0, 10), false, &t);
      if (!t.Before(day)) out.push_back(e);
    }
    return out;
  }
  std::vector<Issuer> ListIssuersForExpirationDate(const ExpDate& e) override { return expDateToIssuer[e.ID()]; }
  std::vector<Serial> ListSerialsForExpirationDateAndIssuer(const ExpDate& e, const Issuer& i) override {
    return expDateIssuerIDToSerials[e.ID() + i.ID()];
  }
  void StreamSerialsForExpirationDateAndIssuer(const ExpDate& e, const Issuer& i,
                                               const std::function<void(const UniqueCertIdentifier&)>& stream) override {
    for (auto& s : expDateIssuerIDToSerials[e.ID() + i.ID()]) stream({e, i, s});
  }
};

// storage/localdiskbackend.go — including its quirks: certificates are written WITHOUT the ".pem" suffix the
// listers look for (:194-199 vs :131,170), and MarkDirty writes "<id>/dirty" relative to the CURRENT
// directory, not rootPath (:89-91).
}  // namespace storage
}  // namespace ctmr
