// Shared device helpers for the gfx950 kernels (wave64, MFMA f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ctk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CTK_WAVE 64

#define CTK_HIP_CHECK_LAUNCH()                 \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)

static inline bool ctk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// XCD-aware remap: the dispatcher places workgroup b on XCD b % 8 (observed, speed only).
// Give consecutive LOGICAL tile ids to the same XCD so tiles that share operand rows hit the
// same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned ctk_xcd_remap(unsigned pid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = pid & 7u, idx = pid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float ctk_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float ctk_gelu_erf(float x) {  // nn.GELU() (exact), blocks.py:48
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float ctk_gelu_tanh(float x) {  // nn.GELU(approximate="tanh"), blocks.py:418
  // 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3): 8 VALU with the raw v_exp_f32 /
  // v_rcp_f32 (1 ulp each; exp -> inf or 0 at the extremes gives the exact limits -0 and x).  ocml tanhf costs ~40.
  const float t = x * fmaf(x * x, 0.044715f, 1.0f);
  const float e = __builtin_amdgcn_exp2f(t * -2.3022081986f);  // -2 sqrt(2/pi) log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---------------------------------------------------------------------------------------
// Coordinate pipeline of bilinear_sampler + ATen grid_sampler (align_corners=True, border):
//   model_utils.py:242-251:  g = c * f32(2/max(size-1,1));  g -= 1
//   ATen GridSampler.h:27-36,58-60:  u = ((g+1)/2)*(size-1);  u = clip(u, 0, size-1)
// Every step is a separately rounded float32 op (explicit _rn intrinsics: no FMA contraction),
// because the round trip is NOT the identity and floor(u) must match the reference bit-for-bit.
// ---------------------------------------------------------------------------------------
struct CtkTap {
  int i0, i1;    // floor index and its (clamped) upper neighbour
  float w0, w1;  // weights of i0 / i1
};

__device__ __forceinline__ CtkTap ctk_tap(float c, int size, float scale /* f32(2/max(size-1,1)) */) {
  float g = __fmul_rn(c, scale);
  g = __fsub_rn(g, 1.0f);
  float u = __fmul_rn(__fadd_rn(g, 1.0f), 0.5f);  // (g+1)/2 : exact halving
  const float hi = (float)(size - 1);
  u = __fmul_rn(u, hi);
  u = fminf(hi, fmaxf(u, 0.0f));
  const float f = floorf(u);
  CtkTap t;
  t.i0 = (int)f;
  t.i1 = min(t.i0 + 1, size - 1);  // out-of-range corner carries weight 0 (u == size-1)
  t.w1 = __fsub_rn(u, f);
  t.w0 = __fsub_rn(__fadd_rn(f, 1.0f), u);
  return t;
}

static inline float ctk_sampler_scale(int size) {  // python double 2/max(size-1,1) -> float32
  return (float)(2.0 / (double)(size - 1 > 1 ? size - 1 : 1));
}
