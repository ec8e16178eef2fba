// Shared device helpers of the persistent ping-pong kernels (gemm_pp.hip: Linear layers; conv_pp.hip: the encoder's
// convolutions as implicit GEMMs): waits / barriers, LDS-DMA, the LDS-transposed full-line epilogue, residual preload.
#pragma once
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int PP_HDR_BYTES = 64;
constexpr int PP_BIAS_BYTES = 8192;  // the whole bias vector (N <= 2048) staged in LDS once per workgroup
// Every workgroup claims the CU's whole LDS (160 KiB) although it needs 136-152 KiB: with less, a workgroup of ANOTHER
// kernel (another process sharing the GPU: tests/test_sharding.py runs two ranks on one device) can be placed beside it,
// this workgroup's LDS then starts at a non-zero base, and the LDS-DMA ring -- whose offsets reach 128 KiB -- went wrong
// in exactly that situation (first seen as garbage tracks when two processes ran the predictor at the same time; the
// kernels were bit-exact alone and against each other).  Owning the whole LDS pins the base to 0.
constexpr int PP_LDS_ALL = 163840;

#define PP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_WAIT_VM(N)                                              \
  do {                                                             \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");       \
    PP_SCHED_FENCE();                                              \
  } while (0)
#define PP_WAIT_LGKM0()                                            \
  do {                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    PP_SCHED_FENCE();                                              \
  } while (0)
#define PP_BARRIER()                                               \
  do {                                                             \
    PP_SCHED_FENCE();                                              \
    pp_jitter<DBG>(dbg, wave, jctr);                               \
    __builtin_amdgcn_s_barrier();                                  \
    pp_jitter<DBG>(dbg, wave, jctr);                               \
    PP_SCHED_FENCE();                                              \
  } while (0)

// experiment (dbg bit 3): pseudo-random per-wave delays at the phase boundaries -- a protocol that is correct must stay
// bit-exact under any timing
template <bool DBG>
__device__ __forceinline__ void pp_jitter(const int dbg, const int wave, unsigned& ctr) {
  if (DBG && (dbg & 8)) {
    ctr = ctr * 1664525u + 1013904223u + (unsigned)wave * 2654435761u;
    const unsigned h = ctr >> 16;
    if ((h & 3) == 0) {
      const int n = (h >> 2) & 7;
      for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(20);
    }
  }
}

typedef const __attribute__((address_space(1))) void* pp_gptr;
typedef __attribute__((address_space(3))) void* pp_lptr;

// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KiB at lds_dst (wave-uniform) + lane * 16
__device__ __forceinline__ void pp_dma16(const unsigned char* base /*uniform*/, unsigned voff, unsigned char* lds_dst /*uniform*/) {
  __builtin_amdgcn_global_load_lds((pp_gptr)(base + voff), (pp_lptr)lds_dst, 16, 0, 0);
}

// ---- epilogue ----------------------------------------------------------------------------------
// EPI bit layout as gemm_f16x3.hip: act (bits 0-1), residual (2), SH output (3), per-row bias table (4), bias (5).
constexpr int pp_epi(int act, bool res, bool split, bool brows, bool bias) {
  return act | (res ? 4 : 0) | (split ? 8 : 0) | (brows ? 16 : 0) | (bias ? 32 : 0);
}

// acc[mi][ni] is the swapped-operand 32x32 accumulator D'[n][m]: lane = output row r32 (+ row_of(mi)), register quad q =
// output columns col_of(ni) + 8q + 4*half + 0..3.
// Stores: written straight from that layout a store instruction touches 32 rows x 32 bytes, and a CU retires such an
// instruction only every ~100 cycles -- measured in tools/gemm_lab.cpp (profiles/r03_gemm_lab_store_experiments.txt): the
// SAME bytes as one contiguous KiB per instruction cost a third.  So every 32x32 sub-tile (f32: 32 rows x 128 B; SH: the
// row's 128-byte line = 32 hi | 32 lo halves) goes through a wave-private 4 KiB LDS image (16-byte chunk c of row r at
// position c ^ (r & 7): conflict-free writes) and leaves as 4 dwordx4 stores of 8 full 128-byte lines each.
// The bias vector lives in LDS too (bias_lds, staged once per workgroup): an ordinary global load in the epilogue would
// make hipcc drain the whole LDS-DMA queue (vmcnt(0)) at its first use AND at the top of the next tile.  The residual
// (EPI bit 2) is not added here: it is in the accumulators already (256 x 192: rides on the tile's first K-tiles).
template <int EPI, int MI, int NI, class RowOf, class ColOf>
__device__ __forceinline__ void pp_epilogue(const CtkGemmP& g, f32x16 (&acc)[MI][NI], const int lane, const int bz, const float unscale,
                                            const float* bias_lds, unsigned char* scratch /* 4 KiB, this wave's */, RowOf row_of,
                                            ColOf col_of, const bool no_store = false, const int n_valid = 0x7fffffff) {
  constexpr int ACT = EPI & 3;
  constexpr bool SPLIT = (EPI & 8) != 0, BROWS = (EPI & 16) != 0, BIAS = (EPI & 32) != 0;
  unsigned char* const c0 = static_cast<unsigned char*>(g.C) + (long)bz * g.c_bs * (SPLIT ? 2 : 4);  // row 0, column 0 of the destination
  const int r32 = lane & 31, half = lane >> 5;
  constexpr bool BV_REGS = (EPI & 4) == 0;  // residual kernels are short of registers here: bias straight from LDS
  f32x4 bv[NI][4];
  if (BIAS && BV_REGS) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[ni][q] = *reinterpret_cast<const f32x4*>(bias_lds + col_of(ni) + q * 8 + half * 4);
  }
  unsigned char* wr = scratch + r32 * 128;  // my row of the image
  const int wsw = r32 & 7;
  const int rrow = lane >> 3, rchunk = lane & 7;  // read-back: 8 lanes per row, 8 rows per instruction
  const unsigned char* rd = scratch + rrow * 128 + ((rchunk ^ rrow) << 4);
  const bool full = row_of(MI - 1) + 32 <= g.M;   // wave-uniform: no row of this wave's tile is beyond M
  const long row_step = (long)8 * g.ldc * (SPLIT ? 2 : 4);
  // software pipeline over the MI x NI sub-tiles: the 4 read-backs of sub-tile k are issued right behind its writes (the LDS
  // serves a wave's accesses in order) and stored one sub-tile later, behind the next sub-tile's arithmetic
  f32x4 pend[4];
  unsigned char* pend_dst = nullptr;
  int pend_row0 = 0;
  auto flush = [&]() {
    if (full) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(pend_dst + i * row_step) = pend[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (pend_row0 + 8 * i < g.M) *reinterpret_cast<f32x4*>(pend_dst + i * row_step) = pend[i];
    }
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int rowc = min(row_of(mi) + r32, g.M - 1);
    const float* bp = BROWS ? g.bias_rows + (long)(rowc % g.bias_period) * g.N + half * 4 : nullptr;
    unsigned char* crow = c0 + (long)(row_of(mi) + rrow) * g.ldc * (SPLIT ? 2 : 4) + rchunk * 16;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      if (col_of(ni) >= n_valid) continue;  // zero-padded output columns (the encoder's 64 / 96-channel convolutions on 128-column tiles)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][q * 4 + e] * unscale;
        if (BIAS) v += BV_REGS ? bv[ni][q] : *reinterpret_cast<const f32x4*>(bias_lds + col_of(ni) + q * 8 + half * 4);
        if (BROWS) v += *reinterpret_cast<const f32x4*>(bp + col_of(ni) + q * 8);
        if (ACT == CTK_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_erf(v[e]);
        } else if (ACT == CTK_ACT_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_tanh(v[e]);
        }
        if (SPLIT) {  // line = 8 chunks: hi halves of columns 8c..8c+7 in chunk c, lo halves in chunk 4 + c
          f16x4 hi, lo;
          ctk_split4(v, hi, lo);
          *reinterpret_cast<f16x4*>(wr + ((q ^ wsw) << 4) + half * 8) = hi;
          *reinterpret_cast<f16x4*>(wr + (((4 + q) ^ wsw) << 4) + half * 8) = lo;
        } else {      // line = 32 floats: columns 4c..4c+3 in chunk c = 2q + half
          *reinterpret_cast<f32x4*>(wr + (((2 * q + half) ^ wsw) << 4)) = v;
        }
      }
      if (pend_dst != nullptr && !no_store) flush();
#pragma unroll
      for (int i = 0; i < 4; ++i) pend[i] = *reinterpret_cast<const f32x4*>(rd + i * 1024);
      // column offset of this sub-tile inside the output row: SH = (col/32) lines of 128 B, f32 = col * 4 B -- the same number
      pend_dst = crow + (long)col_of(ni) * 4;
      pend_row0 = row_of(mi) + rrow;
    }
  }
  if (pend_dst != nullptr && !no_store) flush();
}

template <int EPI>
__device__ __forceinline__ void pp_stage_bias(const CtkGemmP& g, unsigned char* dst, const int tid) {
  if ((EPI & 32) != 0) {
    for (int i = tid; i < g.N / 4; i += 512) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(g.bias)[i];
    __syncthreads();  // before any LDS-DMA is in flight: a plain barrier with full waits
  }
}

// Accumulator start values of a tile: 0, or (EPI bit 2) the residual tile times the weight scale s -- the epilogue's
// "* 1/s" then returns it exactly (s is a power of two).  Like the stores, the residual is moved in full 128-byte lines
// (4 dwordx4 loads of 8 rows each per 32x32 sub-tile, all sub-tiles requested before the first is used) and transposed to
// the accumulator layout through the wave's 4 KiB LDS image: fetched in that layout directly (32 rows x 32 B per
// instruction) the preload cost as much as the stores did (to_out 154 us against 115 us for the same Linear without it).
template <int MI, int NI>
struct PPResid {
  f32x4 line[MI][NI][4];
};

// issue the residual loads of a tile (rows / columns given by row_of / col_of); nothing waits for them here
template <int EPI, int MI, int NI, class RowOf, class ColOf>
__device__ __forceinline__ void pp_resid_issue(const CtkGemmP& g, PPResid<MI, NI>& r, const int lane, const int bz, RowOf row_of, ColOf col_of) {
  if ((EPI & 4) == 0) return;
  const int rrow = lane >> 3, rchunk = lane & 7;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowc = min(row_of(mi) + rrow + 8 * i, g.M - 1);
      const float* rp = g.resid + (long)bz * g.c_bs + (long)rowc * g.ldr + rchunk * 4;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) r.line[mi][ni][i] = *reinterpret_cast<const f32x4*>(rp + col_of(ni));
    }
  }
}

template <int EPI, int MI, int NI>
__device__ __forceinline__ void pp_init_acc(f32x16 (&acc)[MI][NI], const PPResid<MI, NI>& r, const int lane, const float scale, unsigned char* scratch) {
  constexpr bool RES = (EPI & 4) != 0;
  if (!RES) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
    return;
  }
  const int r32 = lane & 31, half = lane >> 5;
  const int rrow = lane >> 3, rchunk = lane & 7;
  unsigned char* wr = scratch + rrow * 128 + ((rchunk ^ rrow) << 4);  // row r = rrow + 8 i, chunk c at position c ^ (r & 7)
  const unsigned char* rd = scratch + r32 * 128;
  const int rsw = r32 & 7;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(wr + i * 1024) = r.line[mi][ni][i] * scale;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rd + (((2 * q + half) ^ rsw) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][ni][q * 4 + e] = v[e];
      }
    }
}

}  // namespace
