// Opt-in kernel timing for bench.py: HIP events recorded on the LAUNCH stream around every kernel
// launch of this library, aggregated per kernel.  Off by default (then a launch costs one relaxed
// bool load).  Bench-only facility: enable/read are not thread safe.
#include "ctk_common.h"
#include "ctk_profile.h"

#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {
struct Rec {
  hipEvent_t a, b;
  const char* name;
  double flops, bytes;
};
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
bool g_on = false;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

CtkProfScope::CtkProfScope(const char* name, double flops, double bytes, hipStream_t s) : idx_(-1), s_(s) {
  if (!g_on) return;
  Rec r;
  r.a = get_event();
  r.b = get_event();
  r.name = name;
  r.flops = flops;
  r.bytes = bytes;
  (void)hipEventRecord(r.a, s);
  idx_ = (long)g_recs.size();
  g_recs.push_back(r);
}

CtkProfScope::~CtkProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].b, s_);
}

extern "C" int ctk_profile_enable(int on) {
  for (auto& r : g_recs) {
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_on = on != 0;
  return CTK_OK;
}

extern "C" int ctk_profile_read(ctk_profile_row* rows, int max_rows, int* nrows) {
  if (!rows || !nrows) return CTK_E_NULL;
  std::map<std::string, ctk_profile_row> agg;
  for (auto& r : g_recs) {
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return (int)e;
    auto it = agg.find(r.name);
    if (it == agg.end()) {
      ctk_profile_row row;
      std::memset(&row, 0, sizeof(row));
      std::strncpy(row.name, r.name, sizeof(row.name) - 1);
      it = agg.emplace(r.name, row).first;
    }
    it->second.launches += 1;
    it->second.total_ms += ms;
    it->second.flops += r.flops;
    it->second.bytes += r.bytes;
  }
  int n = 0;
  for (auto& kv : agg) {
    if (n >= max_rows) break;
    rows[n++] = kv.second;
  }
  *nrows = n;
  return CTK_OK;
}
