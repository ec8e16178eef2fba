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
  char name[32];  // copied: launchers build per-shape names on the stack
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

bool ctk_profile_is_on() { return g_on; }

CtkProfScope::CtkProfScope(const char* name, double flops, double bytes, hipStream_t s) : idx_(-1), s_(s) {
  if (!g_on) return;
  Rec r;
  r.a = get_event();
  r.b = get_event();
  std::strncpy(r.name, name, sizeof(r.name) - 1);
  r.name[sizeof(r.name) - 1] = 0;
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

// ---- MFMA peak probes (tools/bench_peak.py): register-only MFMA loops, 2 waves per SIMD ----------
namespace {
typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef short pbf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void probe_mfma_f32_kernel(int iters, float* out) {
  pf32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + blockIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void probe_mfma_bf16_kernel(int iters, float* out) {
  pf32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  pbf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (short)(0x3f80 + threadIdx.x + e);
    b[e] = (short)(0x3f00 + blockIdx.x % 64 + e);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}

// kind 2: the instruction the split-half kernels issue (v_mfma_f32_32x32x16_f16) on pseudo-random normal-range halves,
// four operand pairs in rotation, so operand-bus and multiplier toggling is that of real data (the constant operands
// of kind 1 draw less power and hold a higher clock): the MFMA rate this chip SUSTAINS under its power cap.
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void probe_mfma_f16_random_kernel(int iters, float* out) {
  pf32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  pf16x8 a[4], b[4];
  unsigned x = 0x9E3779B9u * (threadIdx.x + 257u * blockIdx.x + 1u);
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      x = x * 1664525u + 1013904223u;
      const unsigned short ha = (unsigned short)(((x >> 16) & 0x83ffu) | 0x3800u);  // sign + 10 random mantissa bits, |v| in [0.5, 1)
      x = x * 1664525u + 1013904223u;
      const unsigned short hb = (unsigned short)(((x >> 16) & 0x83ffu) | 0x3400u);  // |v| in [0.25, 0.5)
      a[i][e] = __builtin_bit_cast(_Float16, ha);
      b[i][e] = __builtin_bit_cast(_Float16, hb);
    }
  for (int it = 0; it < iters; it += 4) {  // (iters is rounded up to a multiple of 4 by the launcher)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + j) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
}  // namespace

// kind 0: v_mfma_f32_32x32x2_f32, kind 1: v_mfma_f32_32x32x16_bf16 (constant operands), kind 2: v_mfma_f32_32x32x16_f16 on
// pseudo-random operands.  Returns the flop count launched.
extern "C" int ctk_probe_mfma(int kind, int iters, float* scratch, double* flops, void* stream) {
  if (!scratch || !flops || iters <= 0) return CTK_E_NULL;
  const int blocks = 256 * 2;  // 2 workgroups of 4 waves per CU
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kind == 0) {
    hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(blocks), dim3(256), 0, s, iters, scratch);
    *flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * (2.0 * 32 * 32 * 2);
  } else if (kind == 1) {
    hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(blocks), dim3(256), 0, s, iters, scratch);
    *flops = (double)blocks * 4 * iters * 4.0 * (2.0 * 32 * 32 * 16);
  } else if (kind == 2) {
    iters = (iters + 3) & ~3;
    hipLaunchKernelGGL(probe_mfma_f16_random_kernel, dim3(blocks), dim3(256), 0, s, iters, scratch);
    *flops = (double)blocks * 4 * iters * 4.0 * (2.0 * 32 * 32 * 16);
  } else {
    return CTK_E_SHAPE;
  }
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
