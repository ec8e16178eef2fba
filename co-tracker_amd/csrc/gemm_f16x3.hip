// Split-half GEMM:  C = act(A @ W^T + bias + bias_rows) + resid  at ~fp32 accuracy on the f16 MFMA pipe.
//
// gfx950 has no TF32/xf32; exact-f32 MFMA (gemm.hip) tops out at 157 TF, 1/16 of the f16/bf16 matrix
// rate.  Here every f32 operand is split into two IEEE halves, x = hi + lo with hi = rn16(x),
// lo = rn16(x - hi) (|x - hi - lo| <= 2^-22 |x|), and the product is formed from three
// v_mfma_f32_32x32x16_f16 with f32 accumulation:  lo_w*hi_a + hi_w*lo_a + hi_w*hi_a  (the dropped
// lo*lo term is <= 2^-22 of the product).  Per-product relative error ~7e-7 vs 6e-8 for f32 -- inside
// the fp32 reduction-order noise of the reference at model level (tools/sim_split_bf16.py: f16 x3
// reproduces the f32 goldens to 1.4e-4 px / 2.4e-5 logit where exact f32 gives 1.0e-4 / 1.3e-5;
// bf16 x3, 16 significant bits, misses the 1e-4 logit bar) -- at 16/3 = 5.3x the f32-MFMA ceiling
// (2.5 PF / 3 = 833 TF "f32-equivalent").
//
//  * W is split once at load time (ctk_pack_weight): scaled by a power of two s so that
//    max|W| lands in [2^13, 2^14) (keeps lo out of the f16 subnormal range; exact, undone by 1/s in
//    the epilogue), stored [N][K/32][2][32] halves = one 128-byte line per (row, K-tile): hi(32) lo(32).
//  * A (activations, f32 in HBM) is split on the fly while it is staged global -> VGPR -> LDS
//    (v_cvt_pk_f16_f32 x2 + widen + subtract per float4): ~10 VALU per float4 against 24 MFMAs per
//    wave per K-tile.  |A| must stay below 65504 (f16 max); the path's activations are O(1..100).
//  * Tiling as gemm.hip: 4 waves (2x2), block tile (64*MR) x (64*NR) x 32, operands swapped
//    (D' = W_tile . A_tile^T) so a lane owns 4 consecutive output columns per register quad; LDS rows are
//    40 halves (80 B): the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte bank groups.
//    Double-buffered LDS, one barrier per K-tile, next tile's global loads issued before the MFMAs.
#include "ctk_common.h"
#include "ctk_options.h"
#include "ctk_profile.h"
#include "gemm_params.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BK = 32;
constexpr int PITCH = BK + 8;      // halves per LDS row (80 bytes)
constexpr int HDR_BYTES = 64;      // packed blob header: float s, float 1/s

// ---- shared epilogue -------------------------------------------------------------------------
// The swapped-operand 32x32 accumulator holds D'[n][m]: lane = output row m = lane & 31, register quad q
// = output columns 8q + 4*(lane>>5) + 0..3.  v = act(acc/s + bias + bias_rows) + resid, written either as
// f32 (float4 per quad) or in SH form (4 hi halves + 4 lo halves per quad: feeds the next GEMM's DMA).
template <int MR, int NR>
__device__ __forceinline__ void gemm_epilogue(const CtkGemmP& g, f32x16 (&acc)[MR][NR], const int m_base,
                                              const int n_base, const int r32, const int half, const int bz) {
  const float unscale = reinterpret_cast<const float*>(g.Wp)[1];
  const bool has_res = g.resid != nullptr;
  const float* Rz = has_res ? g.resid + (long)bz * g.c_bs : nullptr;
  f32x4 res[MR][NR][4];
#pragma unroll
  for (int mi = 0; mi < MR; ++mi) {
    const int row = min(m_base + mi * 32 + r32, g.M - 1);
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n_base + ni * 32 + q * 8 + half * 4;
        res[mi][ni][q] = has_res ? *reinterpret_cast<const f32x4*>(Rz + (long)row * g.ldr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  float* Cf = static_cast<float*>(g.C) + (long)bz * g.c_bs;
  _Float16* Ch = static_cast<_Float16*>(g.C) + (long)bz * g.c_bs;
#pragma unroll
  for (int mi = 0; mi < MR; ++mi) {
    const int row = m_base + mi * 32 + r32;
    const float* brow = g.bias_rows ? g.bias_rows + (long)(min(row, g.M - 1) % g.bias_period) * g.N : nullptr;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n_base + ni * 32 + q * 8 + half * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][q * 4 + e] * unscale;
        if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
        if (brow) v += *reinterpret_cast<const f32x4*>(brow + col);
        if (g.act == CTK_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_erf(v[e]);
        } else if (g.act == CTK_ACT_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_tanh(v[e]);
        }
        v += res[mi][ni][q];
        if (row < g.M) {
          if (g.c_split) {
            f16x4 hi, lo;
            ctk_split4(v, hi, lo);
            _Float16* dst = Ch + (long)row * g.ldc + ctk_sh_col(col);
            *reinterpret_cast<f16x4*>(dst) = hi;
            *reinterpret_cast<f16x4*>(dst + 32) = lo;
          } else {
            *reinterpret_cast<f32x4*>(Cf + (long)row * g.ldc + col) = v;
          }
        }
      }
    }
  }
}

// ---- compile-time epilogues of the hot launches -----------------------------------------------------
// For the 12-K-tile Linears of the transformer (K = 384) the epilogue, not the MFMA loop, is the bulk of a wave's
// instruction stream (SQ counters, profiles/r01_gemm_sh_sq_counters.txt: fc1 issues ~2350 VALU per wave there vs
// ~530 in its main loop), and an epilogue wave keeps its SIMD's MFMA pipe idle unless the co-resident workgroup is
// in its main loop.  EPI encodes the flags as constants -- act (bits 0-1), residual (2), SH output (3), per-row
// bias table (4), bias (5) -- so the runtime branches, the zero residual adds and the per-quad 64-bit addressing
// of the generic epilogue disappear: two row base pointers per lane, every column offset an instruction immediate.
constexpr int EPI_GENERIC = -1;
constexpr int epi_code(int act, bool res, bool split, bool brows, bool bias) {
  return act | (res ? 4 : 0) | (split ? 8 : 0) | (brows ? 16 : 0) | (bias ? 32 : 0);
}

template <int MR, int NR, int EPI>
__device__ __forceinline__ void gemm_epilogue_c(const CtkGemmP& g, f32x16 (&acc)[MR][NR], const int m_base,
                                                const int n_base, const int r32, const int half, const int bz) {
  constexpr int ACT = EPI & 3;
  constexpr bool RES = (EPI & 4) != 0, SPLIT = (EPI & 8) != 0, BROWS = (EPI & 16) != 0, BIAS = (EPI & 32) != 0;
  const float unscale = reinterpret_cast<const float*>(g.Wp)[1];
  const int c0 = n_base + half * 4;  // my first column; register quad (ni, q) adds ni*32 + q*8 (compile-time)
  f32x4 bv[NR][4];
  if (BIAS) {
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[ni][q] = *reinterpret_cast<const f32x4*>(g.bias + c0 + ni * 32 + q * 8);
  }
#pragma unroll
  for (int mi = 0; mi < MR; ++mi) {
    const int row = m_base + mi * 32 + r32;
    const int rowc = min(row, g.M - 1);
    const float* rp = RES ? g.resid + (long)bz * g.c_bs + (long)rowc * g.ldr + c0 : nullptr;
    const float* bp = BROWS ? g.bias_rows + (long)(rowc % g.bias_period) * g.N + c0 : nullptr;
    f32x4 rv[NR][4];
    if (RES) {
#pragma unroll
      for (int ni = 0; ni < NR; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[ni][q] = *reinterpret_cast<const f32x4*>(rp + ni * 32 + q * 8);
    }
    float* cf = static_cast<float*>(g.C) + (long)bz * g.c_bs + (long)row * g.ldc + c0;
    // SH row: column c sits at (c >> 5) * 64 + (c & 31) halves; n_base is a multiple of 32 and (c & 31) = q*8 + half*4
    _Float16* ch = static_cast<_Float16*>(g.C) + (long)bz * g.c_bs + (long)row * g.ldc + (n_base >> 5) * 64 + half * 4;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][q * 4 + e] * unscale;
        if (BIAS) v += bv[ni][q];
        if (BROWS) v += *reinterpret_cast<const f32x4*>(bp + ni * 32 + q * 8);
        if (ACT == CTK_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_erf(v[e]);
        } else if (ACT == CTK_ACT_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_tanh(v[e]);
        }
        if (RES) v += rv[ni][q];
        if (row < g.M) {
          if (SPLIT) {
            f16x4 hi, lo;
            ctk_split4(v, hi, lo);
            *reinterpret_cast<f16x4*>(ch + ni * 64 + q * 8) = hi;
            *reinterpret_cast<f16x4*>(ch + ni * 64 + q * 8 + 32) = lo;
          } else {
            *reinterpret_cast<f32x4*>(cf + ni * 32 + q * 8) = v;
          }
        }
      }
  }
}

template <int MR, int NR>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(CtkGemmP g) {
  constexpr int BM = 64 * MR, BN = 64 * NR;
  constexpr int A_LD = BM / 32, W_LD = BN / 32;  // 16-byte loads per thread per K-tile
  constexpr int STAGE = (2 * BM + 2 * BN) * PITCH;  // halves
  constexpr int A_HI = 0, A_LO = BM * PITCH, W_HI = 2 * BM * PITCH, W_LO = (2 * BM + BN) * PITCH;
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];

  const unsigned nblk = gridDim.x;
  unsigned tile = ctk_xcd_remap(blockIdx.x, nblk);
  const int nb = tile % g.nblocks;
  tile /= g.nblocks;
  const int mb = tile % g.mblocks;
  const int bz = tile / g.mblocks;

  const float* A = static_cast<const float*>(g.A) + (long)bz * g.a_bs;
  const int m0 = mb * BM, n0 = nb * BN;
  const int KT = g.K / BK;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  // staging assignment: thread -> (row lr + 32*i, 16-byte column c)
  const int lr = tid >> 3, c8 = tid & 7;
  const float* a_src[A_LD];
  const unsigned short* w_src[W_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const int row = min(m0 + lr + 32 * i, g.M - 1);  // clamp: rows >= M are never stored
    a_src[i] = A + (long)row * g.lda + c8 * 4;
  }
  const unsigned short* wp = g.Wp + HDR_BYTES / 2;
#pragma unroll
  for (int i = 0; i < W_LD; ++i) w_src[i] = wp + (long)(n0 + lr + 32 * i) * KT * 64 + c8 * 8;
  const int w_dst = ((c8 >> 2) ? W_LO : W_HI) + (c8 & 3) * 8;  // plane, column (halves)

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  f32x4 sa[A_LD];
  f16x8 sw[W_LD];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) sa[i] = *reinterpret_cast<const f32x4*>(a_src[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < W_LD; ++i) sw[i] = *reinterpret_cast<const f16x8*>(w_src[i] + kt * 64);
  };
  auto lstore = [&](int buf) {
    _Float16* st = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      f16x4 hi, lo;
      ctk_split4(sa[i], hi, lo);
      *reinterpret_cast<f16x4*>(st + A_HI + (lr + 32 * i) * PITCH + c8 * 4) = hi;
      *reinterpret_cast<f16x4*>(st + A_LO + (lr + 32 * i) * PITCH + c8 * 4) = lo;
    }
#pragma unroll
    for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f16x8*>(st + w_dst + (lr + 32 * i) * PITCH) = sw[i];
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const _Float16* st = lds + buf * STAGE;
    const _Float16* la = st + (wm * 32 * MR + r32) * PITCH + half * 8;
    const _Float16* lw = st + (wn * 32 * NR + r32) * PITCH + half * 8;
#pragma unroll
    for (int j = 0; j < BK / 16; ++j) {
      f16x8 ah[MR], al[MR], wh[NR], wl[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(la + A_HI + i * 32 * PITCH + j * 16);
        al[i] = *reinterpret_cast<const f16x8*>(la + A_LO + i * 32 * PITCH + j * 16);
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        wh[i] = *reinterpret_cast<const f16x8*>(lw + W_HI + i * 32 * PITCH + j * 16);
        wl[i] = *reinterpret_cast<const f16x8*>(lw + W_LO + i * 32 * PITCH + j * 16);
      }
      // operands swapped on purpose (see gemm.hip): D'[n][m], lane = output row m, register quad = 4 columns n.
      // Small terms first; the same accumulator is revisited every MR*NR MFMAs.
#pragma unroll
      for (int mi = 0; mi < MR; ++mi)
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ni], ah[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MR; ++mi)
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ni], al[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MR; ++mi)
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ni], ah[mi], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<MR, NR>(g, acc, m0 + wm * 32 * MR, n0 + wn * 32 * NR, r32, half, bz);
}

// ---- SH-operand kernel: A already split (SH format), both operands straight to LDS by DMA ------------
// Block = WM x WN waves, each wave MR x NR tiles of 32x32; K-tile = 32 columns = ONE 128-byte line per
// row per operand (hi 64 B | lo 64 B).  global_load_lds_dwordx4 writes 1 KiB = 8 rows per wave
// instruction, lane-linear, so the LDS image is unpadded [row][128 B]; bank conflicts are avoided by an
// XOR swizzle of the 16-byte chunk index applied on the SOURCE address (lane (row, pos) fetches chunk
// pos ^ ((row >> 1) & 7)) and undone in the fragment reads: the 16 lanes of a ds_read_b128 group then
// cover 16 distinct 16-byte bank groups.  No staging registers, no conversion, no ds_write.
// NS = LDS stages.  NS = 2: tile kt+2 is requested in the middle of tile kt and must have landed one tile later.
// NS = 3 (one workgroup per CU): tile kt+3 is requested there and has two tiles of MFMAs to land; the mid-tile
// barrier is then a raw s_barrier behind a COUNTED s_waitcnt vmcnt (one tile stays in flight across it) --
// __syncthreads() would drain the DMA queue (cdna_hip_programming.md, "Pipelining across barriers").
template <int WM, int WN, int MR, int NR, int NS, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(WM * WN * 64) void gemm_sh_kernel(CtkGemmP g) {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
  constexpr int GROUPS = (BM + BN) / 8;   // 8-row DMA groups per K-tile (A rows then W rows)
  constexpr int GPW = GROUPS / NW;        // groups per wave
  static_assert(GROUPS % NW == 0, "DMA groups must divide evenly over the waves");
  constexpr int STAGE = (BM + BN) * 128;  // bytes
  static_assert(NS == 2 || NS == 3, "2 or 3 LDS stages");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];

  const unsigned nblk = gridDim.x;
  unsigned tile = ctk_xcd_remap(blockIdx.x, nblk);
  const int nb = tile % g.nblocks;
  tile /= g.nblocks;
  const int mb = tile % g.mblocks;
  const int bz = tile / g.mblocks;
  const int m0 = mb * BM, n0 = nb * BN;
  const int KT = g.K / BK;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r32 = lane & 31, half = lane >> 5;

  // DMA source pointers: group gi = i * NW + wave covers tile rows [8 gi, 8 gi + 8)
  const _Float16* Ash = static_cast<const _Float16*>(g.A) + (long)bz * g.a_bs;
  const _Float16* Wsh = reinterpret_cast<const _Float16*>(g.Wp) + HDR_BYTES / 2;
  const _Float16* src[GPW];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int lrow = (i * NW + wave) * 8 + (lane >> 3);          // row inside the stage image
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);            // source chunk for this LDS position
    if (lrow < BM) {
      const int row = min(m0 + lrow, g.M - 1);                   // clamp: rows >= M are never stored
      src[i] = Ash + (long)row * g.lda + chunk * 8;
    } else {
      src[i] = Wsh + (long)(n0 + lrow - BM) * KT * 64 + chunk * 8;
    }
  }
  auto dma = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < GPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)kt * 64),
                                       (__attribute__((address_space(3))) void*)(lds + stage * STAGE + (i * NW + wave) * 1024),
                                       16, 0, 0);
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  // fragment addressing: row r, data chunk c = plane*4 + j*2 + half lives at r*128 + ((c ^ f(r)) << 4), f(r) = (r>>1)&7;
  // f depends on r32 only (the wave/tile row offsets are multiples of 32)
  const int fsw = (r32 >> 1) & 7;
  const int a_row = (wm * MR * 32 + r32) * 128;
  const int w_row = (BM + wn * NR * 32 + r32) * 128;
  int coff[2][2];  // [j][plane]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) coff[j][p] = ((p * 4 + j * 2 + half) ^ fsw) << 4;

  struct Frags {
    f16x8 ah[MR], al[MR], wh[NR], wl[NR];
  };
  auto load_frags = [&](int stage, int j, Frags& f) {
    const unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      f.ah[i] = *reinterpret_cast<const f16x8*>(st + a_row + i * 4096 + coff[j][0]);
      f.al[i] = *reinterpret_cast<const f16x8*>(st + a_row + i * 4096 + coff[j][1]);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      f.wh[i] = *reinterpret_cast<const f16x8*>(st + w_row + i * 4096 + coff[j][0]);
      f.wl[i] = *reinterpret_cast<const f16x8*>(st + w_row + i * 4096 + coff[j][1]);
    }
  };
  // operands swapped on purpose (see gemm.hip): D'[n][m], lane = output row m, register quad = 4 columns n.
  // Small terms first; the same accumulator is revisited every MR*NR MFMAs.
  auto mma_term = [&](const Frags& f, int term) {  // term 0: wl*ah, 1: wh*al, 2: wh*ah
#pragma unroll
    for (int mi = 0; mi < MR; ++mi)
#pragma unroll
      for (int ni = 0; ni < NR; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? f.wl[ni] : f.wh[ni], term == 1 ? f.al[mi] : f.ah[mi],
                                                             acc[mi][ni], 0, 0, 0);
  };

  // Software pipeline (2 LDS stages, 2 fragment register sets): the fragments of k-step 1 are read from LDS while
  // the MFMAs of k-step 0 run, and those of the NEXT tile's k-step 0 while this tile's k-step 1 runs, so an MFMA
  // phase never waits on LDS latency.  The barrier in the middle of the tile releases its stage: all waves have
  // finished reading it (lgkmcnt(0) is part of __syncthreads) and their DMA of tile kt+1 has landed (vmcnt(0));
  // the DMA of tile kt+2 then has a whole tile of MFMAs to land before it is waited for.
  // Wait until at most `tiles` of my DMA tiles are still in flight (and my LDS reads are done), then barrier.
  auto sync_tiles = [&](int tiles) {
    if (NS == 2) {
      __syncthreads();  // vmcnt(0): my DMA landed; lgkmcnt(0): my reads of the stage about to be overwritten are done
    } else {
      if (tiles >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(GPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  Frags fa, fb;
  dma(0, 0);
  if (KT > 1) dma(1, 1);
  if (NS == 3 && KT > 2) dma(2, 2);
  {
    const int later = min(NS - 1, KT - 1);  // tiles requested after tile 0
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GPW) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(0, 0, fa);
  // (the last tile is peeled so that no control-flow merge sits between a set's loads and the other set's MFMAs:
  //  at a merge hipcc falls back to lgkmcnt(0) and would expose the latency of the loads just issued)
  int st = 0;  // stage of tile kt
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int st_next = (st + 1 == NS) ? 0 : st + 1;
    // The LDS reads of the other register set are issued AFTER the first MFMA group of this one: hipcc waits
    // lgkmcnt(0) before a set's first use, so at that point only loads issued a half-tile ago may be in flight.
    mma_term(fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(st, 1, fb);
    __builtin_amdgcn_sched_barrier(0);
    mma_term(fa, 1);
    mma_term(fa, 2);
    __builtin_amdgcn_sched_barrier(0);
    // tile kt+1 must have landed; with 3 stages tile kt+2 (if any) stays in flight across the barrier
    sync_tiles((NS == 3 && kt + 2 < KT) ? 1 : 0);
    if (kt + NS < KT) dma(kt + NS, st);  // stage st is free: every wave has its fragments of tile kt in registers
    load_frags(st_next, 0, fa);
    __builtin_amdgcn_sched_barrier(0);
    mma_term(fb, 0);
    mma_term(fb, 1);
    mma_term(fb, 2);
    __builtin_amdgcn_sched_barrier(0);
    st = st_next;
  }
  {
    mma_term(fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(st, 1, fb);
    __builtin_amdgcn_sched_barrier(0);
    mma_term(fa, 1);
    mma_term(fa, 2);
    mma_term(fb, 0);
    mma_term(fb, 1);
    mma_term(fb, 2);
  }

  if (EPI == EPI_GENERIC) gemm_epilogue<MR, NR>(g, acc, m0 + wm * 32 * MR, n0 + wn * 32 * NR, r32, half, bz);
  else gemm_epilogue_c<MR, NR, EPI>(g, acc, m0 + wm * 32 * MR, n0 + wn * 32 * NR, r32, half, bz);
}

// ---- deep-pipeline 64 x 64 kernel for the small-M Linears (the 64 virtual tracks: M = 64 S rows) ------------------
// gemm_sh_kernel<2,2,1,1,2> spends a whole LDS-DMA latency per K-tile there when its operands are cold (6 MFMAs per wave
// and K-tile cannot hide an L2 miss behind two stages).  Same tile and fragment layout, but NS stages of 16 KiB: NS - 1
// K-tiles are always in flight behind a counted vmcnt, one raw barrier per K-tile.
template <int NS, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(256) void gemm_sh_deep64_kernel(CtkGemmP g) {
  constexpr int BM = 64, BN = 64, STAGE = (BM + BN) * 128, GPW = 4;  // 16 pieces of 8 rows per K-tile, 4 per wave
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];
  unsigned tile = ctk_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = tile % g.nblocks;
  tile /= g.nblocks;
  const int mb = tile % g.mblocks;
  const int bz = tile / g.mblocks;
  const int m0 = mb * BM, n0 = nb * BN;
  const int KT = g.K / BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  const _Float16* Ash = static_cast<const _Float16*>(g.A) + (long)bz * g.a_bs;
  const _Float16* Wsh = reinterpret_cast<const _Float16*>(g.Wp) + HDR_BYTES / 2;
  const _Float16* src[GPW];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int lrow = (i * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
    if (lrow < BM) src[i] = Ash + (long)min(m0 + lrow, g.M - 1) * g.lda + chunk * 8;
    else src[i] = Wsh + (long)(n0 + lrow - BM) * KT * 64 + chunk * 8;
  }
  auto dma = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < GPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)kt * 64),
                                       (__attribute__((address_space(3))) void*)(lds + stage * STAGE + (i * 4 + wave) * 1024), 16, 0, 0);
  };
  f32x16 acc[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.0f;
  const int fsw = (r32 >> 1) & 7;
  const int a_row = (wm * 32 + r32) * 128, w_row = (BM + wn * 32 + r32) * 128;
  int coff[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) coff[j][p] = ((p * 4 + j * 2 + half) ^ fsw) << 4;

  // prologue: NS - 1 K-tiles requested (short K: the missing ones are re-requests of the last tile, so the counts stay uniform)
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) dma(min(i, KT - 1), i);
  int st = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // K-tile kt has landed (mine: all but the NS - 2 younger requests are complete), then everyone's (barrier)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW * (NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the stage of K-tile kt - 1 is free now (every wave has consumed its fragments before reaching this barrier)
    dma(min(kt + NS - 1, KT - 1), st == 0 ? NS - 1 : st - 1);
    const unsigned char* sp = lds + st * STAGE;
    f16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ah[j] = *reinterpret_cast<const f16x8*>(sp + a_row + coff[j][0]);
      al[j] = *reinterpret_cast<const f16x8*>(sp + a_row + coff[j][1]);
      wh[j] = *reinterpret_cast<const f16x8*>(sp + w_row + coff[j][0]);
      wl[j] = *reinterpret_cast<const f16x8*>(sp + w_row + coff[j][1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // operands swapped, small terms first (as gemm_sh_kernel)
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], ah[j], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], al[j], acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], ah[j], acc[0][0], 0, 0, 0);
    }
    st = (st + 1 == NS) ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the duplicate tail requests
  // (compile-time epilogues for the six Linear flavours: the same arithmetic as gemm_sh_kernel's and the persistent kernels',
  // so that the tail rows of a split launch carry the bits the persistent kernel would have written -- EXCEPT for the "+ residual"
  // Linears (to_out, mlp.fc2), where gemm_pp192_kernel adds residual * s into the accumulators during K-tiles 1..7 and this
  // epilogue adds the residual after unscale and bias: identical up to ONE rounding (tests/test_gpu_gemm_pp.py allows 4e-6
  // relative), so a row's last bit there depends on which side of the cut it falls, i.e. on M and the CU count)
  if (EPI == EPI_GENERIC) gemm_epilogue<1, 1>(g, acc, m0 + wm * 32, n0 + wn * 32, r32, half, bz);
  else gemm_epilogue_c<1, 1, EPI>(g, acc, m0 + wm * 32, n0 + wn * 32, r32, half, bz);
}

// A persistent variant (resident workgroups walking over the tiles, the next tile's first two K-tiles prefetched by DMA
// across the epilogue) was built and measured in round 2 and REMOVED: identical times (q/out shape 151.2 vs 152.4 us,
// K >= 768 slightly slower; profiles/r02_gemm_persistent_ab.txt).  The per-tile cost that tools/bench_gemm_sweep.py
// exposes (t = a + rounds * (T0 + KT * tk): T0 ~ 7 us, tk ~ 1.45 us against 0.73 us of MFMA issue) is therefore not
// workgroup launch, prologue or a cold first DMA -- it is inside the tile.  SQ counters (profiles/r02_sq_counters.txt,
// r02_lds_counters.txt): waves are issuing 16 % of their lifetime and sit in s_waitcnt 61 % of it; the LDS array is only
// ~13 % busy with fragment reads (no conflicts), the MFMA pipe ~40 %: latency-bound at two waves per SIMD, not LDS-bound.

// ---- weight packing ------------------------------------------------------------------------
// hdr[0] = s = 2^(13 - floor(log2(max|W|))), hdr[1] = 1/s   (s = 1 for an all-zero matrix)
__global__ __launch_bounds__(1024) void weight_scale_kernel(const float* W, long ldw, int N, int K, float* hdr) {
  __shared__ float red[16];
  float m = 0.0f;
  const long total = (long)N * K;
  for (long i = threadIdx.x; i < total; i += 1024) m = fmaxf(m, fabsf(W[(i / K) * ldw + i % K]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    int e = 0;
    if (m > 0.0f && m < INFINITY) e = 13 - ilogbf(m);
    e = max(-100, min(100, e));
    hdr[0] = ldexpf(1.0f, e);
    hdr[1] = ldexpf(1.0f, -e);
  }
}

__global__ void weight_pack_kernel(const float* W, long ldw, int N, int K, const float* hdr, unsigned short* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of W
  const int K4 = K / 4;
  if (i >= (long)N * K4) return;
  const int n = i / K4, k = (i % K4) * 4;
  const float s = hdr[0];
  f32x4 v = *reinterpret_cast<const f32x4*>(W + (long)n * ldw + k);
  v *= s;
  f16x4 hi, lo;
  ctk_split4(v, hi, lo);
  _Float16* dst = reinterpret_cast<_Float16*>(out) + HDR_BYTES / 2 + ((long)n * (K / BK) + k / BK) * 64 + (k % BK);
  *reinterpret_cast<f16x4*>(dst) = hi;
  *reinterpret_cast<f16x4*>(dst + 32) = lo;
}

}  // namespace

namespace {
// Dev-build knob (read once; the release library always takes 0): CTK_GEMM_TILE = 0 auto (128x128, 2 blocks/CU) | 2 force 256x128 (8 waves, 1 block/CU,
// 2 LDS stages) | 3 force 256x128 with 3 LDS stages (counted vmcnt + raw barrier) | 4 use 128x384 for N = 384 | 5 64x128 tile (3 workgroups per CU) | 6 256x256 tile for every N % 256 == 0 launch |
// 1 128x128 everywhere (no 256x256).
int gemm_tile_pref() { return (int)CTK_DEV_KNOB("CTK_GEMM_TILE", 0); }

template <typename K>
int launch_with_lds(K kernel, unsigned blocks, unsigned threads, size_t lds_bytes, const CtkGemmP& g, hipStream_t s) {
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, s, g);
  (void)lds_bytes;
  return CTK_OK;
}
}  // namespace

// 64 x 64 tiles (SH operands): the 64 S-row virtual-track Linears, and the tail rows a persistent launch of gemm_pp.hip
// leaves when its tile count is a little more than a whole number of rounds (ctk_launch_gemm_pp)
int ctk_launch_gemm_sh64(CtkGemmP& g, double flops, double bytes, hipStream_t s) {
  g.mblocks = (g.M + 63) / 64; g.nblocks = g.N / 64;
  char pname[32];
  snprintf(pname, sizeof(pname), "gemm_sh_64_k%d_n%d", g.K, g.N);
  CtkProfScope ps(pname, flops, bytes, s);
  const int deep = (int)CTK_DEV_KNOB("CTK_GEMM_DEEP64", 4);  // dev builds: 0 = 2-stage kernel, 4 / 8 = stages
  const dim3 grid((unsigned)((long)g.mblocks * g.nblocks * g.batch));
  if (deep >= 8) hipLaunchKernelGGL((gemm_sh_deep64_kernel<8>), grid, dim3(256), 0, s, g);
  else if (deep >= 4) {
    const int code = epi_code(g.act, g.resid != nullptr, g.c_split != 0, g.bias_rows != nullptr, g.bias != nullptr);
#define CTK_SH64(E) hipLaunchKernelGGL((gemm_sh_deep64_kernel<4, E>), grid, dim3(256), 0, s, g)
    switch (code) {
      case epi_code(CTK_ACT_GELU_ERF, false, true, false, true): CTK_SH64(epi_code(CTK_ACT_GELU_ERF, false, true, false, true)); break;    // corr_mlp.fc1
      case epi_code(CTK_ACT_NONE, false, true, false, true): CTK_SH64(epi_code(CTK_ACT_NONE, false, true, false, true)); break;            // corr_mlp.fc2 -> x
      case epi_code(CTK_ACT_NONE, false, false, true, false): CTK_SH64(epi_code(CTK_ACT_NONE, false, false, true, false)); break;          // input_transform
      case epi_code(CTK_ACT_NONE, false, false, false, true): CTK_SH64(epi_code(CTK_ACT_NONE, false, false, false, true)); break;          // to_q / to_kv
      case epi_code(CTK_ACT_NONE, true, false, false, true): CTK_SH64(epi_code(CTK_ACT_NONE, true, false, false, true)); break;            // to_out / mlp.fc2 (+ residual)
      case epi_code(CTK_ACT_GELU_TANH, false, true, false, true): CTK_SH64(epi_code(CTK_ACT_GELU_TANH, false, true, false, true)); break;  // mlp.fc1
      default: CTK_SH64(EPI_GENERIC);
    }
#undef CTK_SH64
  }
  else hipLaunchKernelGGL((gemm_sh_kernel<2, 2, 1, 1, 2>), grid, dim3(256), 0, s, g);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

int ctk_launch_gemm_f16x3(CtkGemmP& g, double flops, double bytes, hipStream_t s) {
  // recorder rows are per (tile, K, N): the K = 384 Linears and corr_mlp.fc1 (K = 2432) sit at very different
  // fractions of the MFMA ceiling and must not be averaged under one name
  char pname[32];
  auto prof_name = [&](const char* tile) {
    snprintf(pname, sizeof(pname), "gemm_sh_%s_k%d_n%d", tile, g.K, g.N);
    return pname;
  };

  if (g.a_split) {  // round 3: persistent ping-pong kernels for the big N % 256 == 0 / N % 192 == 0 Linears
    const int rc = ctk_launch_gemm_pp(g, flops, bytes, s);
    if (rc >= 0) return rc;
  }
  const long blocks128 = (long)((g.M + 127) / 128) * (g.N / 128) * g.batch;
  // 128 x 128 tiles once they fill most of the 512 resident slots, 64 x 64 below (CTK_GEMM_BIG_MIN overrides the threshold)
  const long big_min = CTK_DEV_KNOB("CTK_GEMM_BIG_MIN", 384L);
  const bool big = (g.N % 128) == 0 && blocks128 >= big_min;
  if (g.a_split) {
    const int pref = gemm_tile_pref();
    // 128x384 tile (8 waves as 2 x 4, each 64 x 96): the block owns full rows of an N = 384 Linear, A is fetched once
    // instead of three times and a wave issues 1.6 instead of 3.2 non-MFMA instructions per MFMA.  Opt-in only
    // (CTK_GEMM_TILE=4): for corr_mlp.fc1 it is 6 % faster in tools/bench_gemm.py (2.62 -> 2.47 ms) but slower inside
    // the update iteration (2.49 ms per launch, +25 ms per C3 step) -- one 8-wave block per CU starts cold behind
    // the sampler where two 4-wave blocks overlap their prologues.  Round 2 re-measured it with compile-time epilogues
    // for every N = 384 Linear (profiles/r02_gemm_fullrow_ab.txt): q/out -3...-9 %, fc2 -3...-9 %, C3 step +3.5 % slower.
    const long rows128 = (g.M + 127) / 128;
    const int code256 = epi_code(g.act, g.resid != nullptr, g.c_split != 0, g.bias_rows != nullptr, g.bias != nullptr);
    const bool epi256 = code256 == epi_code(CTK_ACT_NONE, false, false, false, true) ||      // to_kv
                        code256 == epi_code(CTK_ACT_GELU_TANH, false, true, false, true) ||  // mlp.fc1
                        code256 == epi_code(CTK_ACT_NONE, false, true, false, true);         // corr_mlp.fc2
    const long blocks256 = (long)((g.M + 255) / 256) * (g.N / 256) * g.batch;  // one 8-wave block per CU: want >= 2 rounds
    if (big && (g.N % 256) == 0 && blocks256 >= 512 && pref != 1 && (pref == 6 || epi256)) {
      // 256x256 tile, 8 waves as 2 x 4, wave tile 128 x 64: 48 MFMAs per 24 fragment reads (128x128: 24 per 16).
      // Default for the N % 256 == 0 Linears that have a compile-time epilogue (to_kv, mlp.fc1, corr_mlp.fc2):
      // -5..-11 % per launch in tools/bench_gemm.py, -0.6 % per C3 step measured in situ.  CTK_GEMM_TILE=1 disables.
      g.mblocks = (g.M + 255) / 256; g.nblocks = g.N / 256;
      CtkProfScope ps(prof_name("256"), flops, bytes, s);
      const dim3 grid((unsigned)((long)g.mblocks * g.nblocks * g.batch)), blk(512);
      if (code256 == epi_code(CTK_ACT_NONE, false, false, false, true))
        hipLaunchKernelGGL((gemm_sh_kernel<2, 4, 4, 2, 2, epi_code(CTK_ACT_NONE, false, false, false, true)>), grid, blk, 0, s, g);
      else if (code256 == epi_code(CTK_ACT_GELU_TANH, false, true, false, true))
        hipLaunchKernelGGL((gemm_sh_kernel<2, 4, 4, 2, 2, epi_code(CTK_ACT_GELU_TANH, false, true, false, true)>), grid, blk, 0, s, g);
      else if (code256 == epi_code(CTK_ACT_NONE, false, true, false, true))
        hipLaunchKernelGGL((gemm_sh_kernel<2, 4, 4, 2, 2, epi_code(CTK_ACT_NONE, false, true, false, true)>), grid, blk, 0, s, g);
      else
        hipLaunchKernelGGL((gemm_sh_kernel<2, 4, 4, 2, 2, EPI_GENERIC>), grid, blk, 0, s, g);
    } else if (g.N == 384 && g.batch == 1 && pref == 4 && rows128 >= 256) {
      g.mblocks = (int)rows128; g.nblocks = 1;
      CtkProfScope ps(prof_name("128x384"), flops, bytes, s);
      const int code = epi_code(g.act, g.resid != nullptr, g.c_split != 0, g.bias_rows != nullptr, g.bias != nullptr);
      const dim3 grid((unsigned)rows128), blk(512);
#define CTK_SH384(E) hipLaunchKernelGGL((gemm_sh_kernel<2, 4, 2, 3, 2, E>), grid, blk, 0, s, g)
      switch (code) {
        case epi_code(CTK_ACT_GELU_ERF, false, true, false, true): CTK_SH384(epi_code(CTK_ACT_GELU_ERF, false, true, false, true)); break;  // corr_mlp.fc1
        case epi_code(CTK_ACT_NONE, false, false, true, false): CTK_SH384(epi_code(CTK_ACT_NONE, false, false, true, false)); break;        // input_transform
        case epi_code(CTK_ACT_NONE, false, false, false, true): CTK_SH384(epi_code(CTK_ACT_NONE, false, false, false, true)); break;        // to_q
        case epi_code(CTK_ACT_NONE, true, false, false, true): CTK_SH384(epi_code(CTK_ACT_NONE, true, false, false, true)); break;          // to_out / mlp.fc2
        default: CTK_SH384(EPI_GENERIC);
      }
#undef CTK_SH384
    } else if (g.N == 384 && g.batch == 1 && (pref == 8 || pref == 9) && rows128 >= 256) {
      // Round-2 experiments, ONE wave per SIMD (4 waves, one workgroup per CU, up to 512 registers per wave):
      //   8: 128 x 384 tile, waves 1 x 4, wave tile 128 x 96 (36 MFMAs per 14 fragment reads), 2 LDS stages of 64 KB
      //   9: 256 x 128 tile, waves 2 x 2, wave tile 128 x 64 (24 MFMAs per 12 reads), 3 LDS stages of 48 KB, counted vmcnt
      const int code = epi_code(g.act, g.resid != nullptr, g.c_split != 0, g.bias_rows != nullptr, g.bias != nullptr);
      const dim3 blk(256);
      if (pref == 8) {
        g.mblocks = (int)rows128; g.nblocks = 1;
        CtkProfScope ps(prof_name("w128x384"), flops, bytes, s);
        const dim3 grid((unsigned)rows128);
#define CTK_SHW(E) hipLaunchKernelGGL((gemm_sh_kernel<1, 4, 4, 3, 2, E>), grid, blk, 0, s, g)
        switch (code) {
          case epi_code(CTK_ACT_GELU_ERF, false, true, false, true): CTK_SHW(epi_code(CTK_ACT_GELU_ERF, false, true, false, true)); break;
          case epi_code(CTK_ACT_NONE, false, false, false, true): CTK_SHW(epi_code(CTK_ACT_NONE, false, false, false, true)); break;
          case epi_code(CTK_ACT_NONE, true, false, false, true): CTK_SHW(epi_code(CTK_ACT_NONE, true, false, false, true)); break;
          default: CTK_SHW(EPI_GENERIC);
        }
#undef CTK_SHW
      } else {
        g.mblocks = (g.M + 255) / 256; g.nblocks = g.N / 128;
        CtkProfScope ps(prof_name("w256x128x3"), flops, bytes, s);
        const dim3 grid((unsigned)((long)g.mblocks * g.nblocks));
#define CTK_SHW(E) hipLaunchKernelGGL((gemm_sh_kernel<2, 2, 4, 2, 3, E>), grid, blk, 0, s, g)
        switch (code) {
          case epi_code(CTK_ACT_GELU_ERF, false, true, false, true): CTK_SHW(epi_code(CTK_ACT_GELU_ERF, false, true, false, true)); break;
          case epi_code(CTK_ACT_NONE, false, false, false, true): CTK_SHW(epi_code(CTK_ACT_NONE, false, false, false, true)); break;
          case epi_code(CTK_ACT_NONE, true, false, false, true): CTK_SHW(epi_code(CTK_ACT_NONE, true, false, false, true)); break;
          default: CTK_SHW(EPI_GENERIC);
        }
#undef CTK_SHW
      }
    } else if (big && pref == 3) {
      g.mblocks = (g.M + 255) / 256; g.nblocks = g.N / 128;
      CtkProfScope ps("gemm_sh_256x128x3", flops, bytes, s);
      hipLaunchKernelGGL((gemm_sh_kernel<4, 2, 2, 2, 3>), dim3((unsigned)((long)g.mblocks * g.nblocks * g.batch)), dim3(512), 0, s, g);
    } else if (big && pref == 2) {
      g.mblocks = (g.M + 255) / 256; g.nblocks = g.N / 128;
      CtkProfScope ps("gemm_sh_256x128", flops, bytes, s);
      hipLaunchKernelGGL((gemm_sh_kernel<4, 2, 2, 2, 2>), dim3((unsigned)((long)g.mblocks * g.nblocks * g.batch)), dim3(512), 0, s, g);
    } else if (big) {
      g.mblocks = (g.M + 127) / 128; g.nblocks = g.N / 128;
      // the six flag combinations of the update path get compile-time epilogues; anything else the generic one
      const int code = epi_code(g.act, g.resid != nullptr, g.c_split != 0, g.bias_rows != nullptr, g.bias != nullptr);
      // CTK_GEMM_TILE=5: 64x128 tile (wave 32x64, 48 KB of LDS -> three workgroups per CU instead of two)
      const bool t64 = pref == 5;
      if (t64) { g.mblocks = (g.M + 63) / 64; }
      CtkProfScope ps(prof_name(t64 ? "64x128" : "128"), flops, bytes, s);
      const dim3 grid((unsigned)((long)g.mblocks * g.nblocks * g.batch)), blk(256);
#define CTK_SH128(E)                                                                     \
  do {                                                                                   \
    if (t64) hipLaunchKernelGGL((gemm_sh_kernel<2, 2, 1, 2, 2, E>), grid, blk, 0, s, g); \
    else hipLaunchKernelGGL((gemm_sh_kernel<2, 2, 2, 2, 2, E>), grid, blk, 0, s, g);     \
  } while (0)
      if (CTK_DEV_KNOB("CTK_GEMM_EPI", 1) == 0) CTK_SH128(EPI_GENERIC);  // dev builds: CTK_GEMM_EPI=0 forces the generic epilogue
      else switch (code) {
        case epi_code(CTK_ACT_GELU_ERF, false, true, false, true): CTK_SH128(epi_code(CTK_ACT_GELU_ERF, false, true, false, true)); break;    // corr_mlp.fc1
        case epi_code(CTK_ACT_NONE, false, true, false, true): CTK_SH128(epi_code(CTK_ACT_NONE, false, true, false, true)); break;            // corr_mlp.fc2 -> x
        case epi_code(CTK_ACT_NONE, false, false, true, false): CTK_SH128(epi_code(CTK_ACT_NONE, false, false, true, false)); break;          // input_transform (+ time bias rows)
        case epi_code(CTK_ACT_NONE, false, false, false, true): CTK_SH128(epi_code(CTK_ACT_NONE, false, false, false, true)); break;          // to_q / to_kv
        case epi_code(CTK_ACT_NONE, true, false, false, true): CTK_SH128(epi_code(CTK_ACT_NONE, true, false, false, true)); break;            // to_out / mlp.fc2 (+ residual)
        case epi_code(CTK_ACT_GELU_TANH, false, true, false, true): CTK_SH128(epi_code(CTK_ACT_GELU_TANH, false, true, false, true)); break;  // mlp.fc1
        default: CTK_SH128(EPI_GENERIC);
      }
#undef CTK_SH128
    } else {
      return ctk_launch_gemm_sh64(g, flops, bytes, s);
    }
  } else if (big) {
    g.mblocks = (g.M + 127) / 128; g.nblocks = g.N / 128;
    CtkProfScope ps("gemm_f16x3_128x128", flops, bytes, s);
    hipLaunchKernelGGL((gemm_f16x3_kernel<2, 2>), dim3((unsigned)blocks128), dim3(256), 0, s, g);
  } else {
    g.mblocks = (g.M + 63) / 64; g.nblocks = g.N / 64;
    const long blocks = (long)g.mblocks * g.nblocks * g.batch;
    CtkProfScope ps("gemm_f16x3_64x64", flops, bytes, s);
    hipLaunchKernelGGL((gemm_f16x3_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, s, g);
  }
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

// f32 [M][K] (row stride ld) -> SH [M][K/32][2][32]
namespace {
__global__ void split_rows_kernel(const float* x, long ld, long M, int K, _Float16* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4
  const int K4 = K / 4;
  if (i >= M * K4) return;
  const long m = i / K4;
  const int k = (int)(i % K4) * 4;
  f16x4 hi, lo;
  ctk_split4(*reinterpret_cast<const f32x4*>(x + m * ld + k), hi, lo);
  _Float16* dst = out + m * (long)(K / 32) * 64 + ctk_sh_col(k);
  *reinterpret_cast<f16x4*>(dst) = hi;
  *reinterpret_cast<f16x4*>(dst + 32) = lo;
}
}  // namespace

extern "C" int ctk_split_rows(const float* x, int64_t ld, int64_t M, int32_t K, void* out, void* stream) {
  if (!x || !out) return CTK_E_NULL;
  if (M <= 0 || K <= 0 || (K % 32)) return CTK_E_SHAPE;
  if ((ld % 4) || !ctk_aligned16(x) || !ctk_aligned16(out)) return CTK_E_ALIGN;
  const long total = M * (K / 4);
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     (long)ld, (long)M, K, static_cast<_Float16*>(out));
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_pack_weight_bytes(int32_t N, int32_t K, size_t* out_bytes) {
  if (!out_bytes) return CTK_E_NULL;
  if (N <= 0 || K <= 0 || (K % BK)) return CTK_E_SHAPE;
  *out_bytes = (size_t)HDR_BYTES + (size_t)N * K * 4;
  return CTK_OK;
}

extern "C" int ctk_pack_weight(const float* W, int64_t ldw, int32_t N, int32_t K, void* packed, void* stream) {
  if (!W || !packed) return CTK_E_NULL;
  if (N <= 0 || K <= 0 || (K % BK)) return CTK_E_SHAPE;
  if ((ldw % 4) || !ctk_aligned16(W) || !ctk_aligned16(packed)) return CTK_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1024), 0, s, W, (long)ldw, N, K, static_cast<float*>(packed));
  CTK_HIP_CHECK_LAUNCH();
  const long total = (long)N * (K / 4);
  hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, (long)ldw, N, K,
                     static_cast<const float*>(packed), static_cast<unsigned short*>(packed));
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
