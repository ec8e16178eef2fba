// Launch parameters shared by the GEMM back ends (gemm.hip: exact f32 MFMA; gemm_f16x3.hip: split-half MFMA)
// and the split-half ("SH") storage format helpers.
//
// SH format of a matrix X[M][K] (K % 32 == 0): IEEE halves [M][K/32][2][32] -- for every row and
// 32-column tile one 128-byte line holding the 32 hi halves then the 32 lo halves, x = hi + lo with
// hi = rn16(x), lo = rn16(x - hi).  Same bytes as f32.  It is what the split-half GEMM reads with
// direct-to-LDS loads, so producers (LayerNorm, attention, GEMM epilogues, token assembly) write it.
#pragma once
#include "ctk_common.h"

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// x = hi + lo, both IEEE half, round-to-nearest-even (v_cvt_pk_f16_f32).
__device__ __forceinline__ void ctk_split4(const f32x4 v, f16x4& hi, f16x4& lo) {
  hi = __builtin_convertvector(v, f16x4);
  const f32x4 r = v - __builtin_convertvector(hi, f32x4);  // exact in f32
  lo = __builtin_convertvector(r, f16x4);
}
__device__ __forceinline__ void ctk_split2(const f32x2 v, f16x2& hi, f16x2& lo) {
  hi = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(hi, f32x2);
  lo = __builtin_convertvector(r, f16x2);
}
// halves offset of column c inside an SH row (hi plane; lo plane is +32)
__device__ __host__ __forceinline__ long ctk_sh_col(int c) { return (long)(c >> 5) * 64 + (c & 31); }

__device__ __forceinline__ f16x8 ctk_cat8(const f16x4 a, const f16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// 8 consecutive f32 (two float4) -> hi / lo f16x8
__device__ __forceinline__ void ctk_split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
  f16x4 ah, al, bh, bl;
  ctk_split4(a, ah, al);
  ctk_split4(b, bh, bl);
  hi = ctk_cat8(ah, bh);
  lo = ctk_cat8(al, bl);
}
__device__ __forceinline__ f32x16 ctk_mma3(const f16x8 ah, const f16x8 al, const f16x8 bh, const f16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);  // small terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  return acc;
}

struct CtkGemmP {
  const void* A; long lda; int M;   // f32 [M][lda] or SH halves (lda = halves per row) when a_split
  const float* W; long ldw; int N; int K;
  const unsigned short* Wp;         // packed split-half weights (gemm_f16x3.hip) or null
  void* C; long ldc;                // f32 or SH halves when c_split
  const float* bias;
  const float* bias_rows; int bias_period;
  const float* resid; long ldr;
  int act;
  int batch; long a_bs; long c_bs;  // batch strides, in elements of the respective format (floats / halves)
  int a_split, c_split;
  int mblocks, nblocks;
};

// gemm_f16x3.hip
int ctk_launch_gemm_f16x3(CtkGemmP& g, double flops, double bytes, hipStream_t s);
int ctk_launch_gemm_sh64(CtkGemmP& g, double flops, double bytes, hipStream_t s);  // 64 x 64 tiles (SH operands)
// gemm_pp.hip: persistent ping-pong kernels; returns -1 when the shape is not theirs (caller falls back)
int ctk_launch_gemm_pp(CtkGemmP& g, double flops, double bytes, hipStream_t s);
