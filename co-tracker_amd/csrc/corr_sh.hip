// Correlation sampler of the split-half pipeline: 49x49 correlation volumes in SH format, f16 MFMA x3.
//
// Replaces (cotracker3_online.py:192-204) get_correlation_feat -> bilinear_sampler -> einsum like corr.hip,
// but with the two linear steps exchanged: instead of blending 49 patch vectors of 128 channels and then
// correlating them (49 x 128 blends per frame), the 8x8 (at most 9x9) pixel FOOTPRINT that all 49 bilinear
// taps of a frame touch is correlated with the 49 support vectors first,
//     C[pixel][q] = sum_c f[pixel][c] * s[q][c]          (MFMA, 64..96 x 64 x 128 per frame)
// and the bilinear blend is applied to the 49x49 results,
//     D[p][q] = w00 C[pix00(p)][q] + w10 C[pix10(p)][q] + w01 C[pix01(p)][q] + w11 C[pix11(p)][q]
// with exactly the reference's tap indices and weights (ctk_tap: model_utils.py:242-251 + ATen
// grid_sampler_3d).  Same value up to f32 rounding (sum and blend commute), 40x less blend arithmetic, and
// the MFMA A operand is now raw pyramid data: the window's pyramid is converted once to SH format
// (scaled by 2^8, exact) so footprints need no per-iteration blend/split arithmetic at all.
// Products use the split-half scheme of gemm_f16x3.hip (3 x v_mfma_f32_32x32x16_f16, ~2^-21 relative).
//
// Workgroup = (point n, level l, <=16 frames), 4 waves, TWO workgroups per CU (LDS 77 KiB each) so that one
// workgroup's MFMA phase overlaps the other's blend / store phase:
//   prologue: the chunk's coordinates and all per-frame tap tables go to LDS; the support patch [49][128] is
//     split once into LDS (K-tile major, XOR-swizzled 16-byte chunks);
//   per frame: the footprint (<= 81 pixels x 512 B of SH data) is PREFETCHED INTO REGISTERS one frame ahead
//     (8 or 12 global_load_dwordx4 per thread, 32 lanes per pixel) and written to a single LDS buffer with the
//     swizzle; wave w = (row tile w>>1, column tile w&1) runs 8 k-steps x 3 MFMAs on two accumulators;
//     accumulators -> f32 C[pixel][q] in LDS (aliasing the consumed footprint); 245 threads blend 12 (or 1)
//     consecutive q of one tap p into an f32 staging row; all threads then split 4 consecutive outputs each and
//     store hi/lo halves into the SH volume row (n*S+t), column p*49+q == the reference's (h,w,i,j)
//     flattening (:205); columns 2401..2431 (K padding of corr_mlp.fc1) are written as zeros.
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"

namespace {

constexpr int TC = 16;                       // frames per workgroup
constexpr int FROWS = 96;                    // footprint rows held in LDS (>= 81)
constexpr int NKT = CTK_C / 32;              // 4 K-tiles
constexpr int SUP_KT = CTK_TAPS * 128;       // bytes per K-tile of the support image: 49 rows x 128 B (rows 49..63 of a
                                             // 64-row MFMA tile read the next K-tile / the footprint: finite, unused)
constexpr int SUP_BYTES = NKT * SUP_KT;      // 25088
constexpr int FP_BYTES = NKT * FROWS * 128;  // 48 KiB  [ktile][96 rows][128 B]; later C [96][68] f32 + staging [2432] f32
constexpr int TAB_BYTES = TC * 256 + 128;    // per-frame tap tables + the chunk's coordinates
constexpr int CPITCH = 68;                   // floats per C row (16-byte aligned rows for ds_read_b128)
constexpr int STG_OFF = FROWS * CPITCH * 4;  // 26112: f32 staging row of the blended outputs
constexpr float FSCALE = 256.0f;             // both operands are scaled by 2^8 before the f16 split
constexpr float UNSCALE = 1.0f / 65536.0f;
constexpr int QUADS = CTK_CORR_LD / 4;       // 608 output quads per (frame, level, point)
static_assert(STG_OFF + CTK_CORR_LD * 4 <= FP_BYTES, "C + staging must fit in the footprint buffer");
static_assert(2 * (SUP_BYTES + FP_BYTES + TAB_BYTES) <= 160 * 1024, "two workgroups per CU");

struct CorrShP {
  const _Float16* fm[CTK_LEVELS];  // SH pyramid of the window: [S][H][W][4][2][32] halves, scaled by 2^8
  const float* support[CTK_LEVELS];
  int H[CTK_LEVELS], W[CTK_LEVELS];
  float sx[CTK_LEVELS], sy[CTK_LEVELS];
  const float* coords;  // [S,N,2]
  const uint8_t* mask;  // [N] or null
  _Float16* out;        // [L][ncount*S][2*CTK_CORR_LD] halves
  long out_level_stride;
  int S, N, n0, ncount, tchunks;
};

struct FrameTab {  // per-frame tap table (LDS), 256 bytes
  int fx0[7], fx1[7], fy0[7], fy1[7];       // tap corner columns / rows relative to the footprint origin
  float wx0[7], wx1[7], wy0[7], wy1[7];
  int xb, yb, fw, fh;                        // footprint origin and size (<= 9 x 9)
  int pad[4];
};
static_assert(sizeof(FrameTab) == 256, "FrameTab layout");

__global__ __launch_bounds__(256, 2) void corr_volume_sh_kernel(CorrShP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[SUP_BYTES + FP_BYTES + TAB_BYTES];
  unsigned char* sup = lds;
  unsigned char* fp = lds + SUP_BYTES;
  FrameTab* tabs = reinterpret_cast<FrameTab*>(lds + SUP_BYTES + FP_BYTES);
  float* cxy = reinterpret_cast<float*>(lds + SUP_BYTES + FP_BYTES + TC * 256);  // [TC][2]

  unsigned bid = ctk_xcd_remap(blockIdx.x, gridDim.x);
  const int tc = bid % p.tchunks;
  bid /= p.tchunks;
  const int lvl = bid % CTK_LEVELS;
  const int nl = bid / CTK_LEVELS;  // local point index
  const int n = p.n0 + nl;
  const int t0 = tc * TC;
  const int nt = min(TC, p.S - t0);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r32 = lane & 31, half = lane >> 5;
  constexpr long ROW_H = 2 * CTK_CORR_LD;  // halves per SH volume row
  _Float16* out_base = p.out + (long)lvl * p.out_level_stride + ((long)nl * p.S + t0) * ROW_H;

  const bool live = p.mask ? (p.mask[n] != 0) : true;
  if (!live) {  // support features of not-yet-queried tracks are zeroed (cotracker3_online.py:493-496)
    const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long i = tid; i < (long)nt * ROW_H / 8; i += 256) reinterpret_cast<f16x8*>(out_base)[i] = z;
    return;
  }

  const int H = p.H[lvl], W = p.W[lvl];
  const float sx = p.sx[lvl], sy = p.sy[lvl];
  const float inv = 1.0f / (float)(1 << lvl);  // coords / 2**i : exact
  const _Float16* fm = p.fm[lvl];

  // ---- prologue: coordinates -> LDS; support patch -> split, scaled, swizzled image ------------------------
  if (tid < 2 * nt) cxy[tid] = p.coords[((long)(t0 + (tid >> 1)) * p.N + n) * 2 + (tid & 1)];
  {
    const float* sp = p.support[lvl] + (long)n * CTK_TAPS * CTK_C;
    f32x4 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {  // float4 i = tid + 256 j of the [49][32] float4 patch
      const int i = min(tid + 256 * j, CTK_TAPS * 32 - 1);
      v[j] = *reinterpret_cast<const f32x4*>(sp + i * 4);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
      if (i < CTK_TAPS * 32) {
        f16x4 hi, lo;
        ctk_split4(v[j] * FSCALE, hi, lo);
        const int kt = c4 >> 3, k8 = (c4 & 7) >> 1, sub = c4 & 1;  // K-tile, 16-byte chunk inside the plane, half of it
        const int fs = (row >> 1) & 7;
        unsigned char* base = sup + kt * SUP_KT + row * 128 + sub * 8;
        *reinterpret_cast<f16x4*>(base + ((k8 ^ fs) << 4)) = hi;
        *reinterpret_cast<f16x4*>(base + (((4 + k8) ^ fs) << 4)) = lo;
      }
    }
  }
  __syncthreads();
  // tap tables of all frames of the chunk: thread (frame tid/14, axis, tap k)
  if (tid < 14 * nt) {
    const int tl = tid / 14, j = tid - tl * 14, k = j % 7;
    FrameTab* tab = tabs + tl;
    if (j < 7) {
      const float cx = __fmul_rn(cxy[2 * tl], inv);
      const CtkTap a0 = ctk_tap(__fadd_rn(cx, -3.0f), W, sx);
      const CtkTap t = ctk_tap(__fadd_rn(cx, (float)(k - 3)), W, sx);
      tab->fx0[k] = t.i0 - a0.i0; tab->fx1[k] = t.i1 - a0.i0; tab->wx0[k] = t.w0; tab->wx1[k] = t.w1;
      if (k == 6) { tab->xb = a0.i0; tab->fw = t.i1 - a0.i0 + 1; }
    } else {
      const float cy = __fmul_rn(cxy[2 * tl + 1], inv);
      const CtkTap a0 = ctk_tap(__fadd_rn(cy, -3.0f), H, sy);
      const CtkTap t = ctk_tap(__fadd_rn(cy, (float)(k - 3)), H, sy);
      tab->fy0[k] = t.i0 - a0.i0; tab->fy1[k] = t.i1 - a0.i0; tab->wy0[k] = t.w0; tab->wy1[k] = t.w1;
      if (k == 6) { tab->yb = a0.i0; tab->fh = t.i1 - a0.i0 + 1; }
    }
  }
  __syncthreads();

  // ---- footprint prefetch into registers: load i of thread (wave w, lane l) = pixel row (4 i + w) * 2 + (l >> 5),
  //      16-byte chunk l & 31 of that pixel's 512 B (K-tile = chunk >> 3)
  f16x8 pre[12];
  auto prefetch = [&](int tl) {
    const FrameTab* tab = tabs + tl;
    const int fw = tab->fw, npx = fw * tab->fh;
    const _Float16* frame = fm + ((long)(t0 + tl) * H * W + (long)tab->yb * W + tab->xb) * (2 * CTK_C) + (lane & 31) * 8;
    const float rfw = 1.0f / (float)fw;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < 8 || npx > 64) {
        const int r = min((4 * i + wave) * 2 + (lane >> 5), npx - 1);  // rows past the footprint re-read its last pixel
        int fy = (int)((float)r * rfw);                               // r / fw (r < 96, fw <= 9), fixed up below
        fy -= (fy * fw > r);
        fy += ((fy + 1) * fw <= r);
        const int fx = r - fy * fw;
        pre[i] = *reinterpret_cast<const f16x8*>(frame + ((long)fy * W + fx) * (2 * CTK_C));
      }
    }
  };
  auto commit = [&](int npx) {  // registers -> LDS image [ktile][row][128 B], chunk position = chunk ^ ((row >> 1) & 7)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < 8 || npx > 64) {
        const int r = (4 * i + wave) * 2 + (lane >> 5);
        const int cc = lane & 31, kt = cc >> 3, c = cc & 7;
        *reinterpret_cast<f16x8*>(fp + kt * (FROWS * 128) + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = pre[i];
      }
    }
  };
  prefetch(0);

  // fragment addressing (as gemm_f16x3.hip): chunk c = plane*4 + s*2 + half of row r sits at r*128 + ((c ^ f(r)) << 4)
  const int fsw = (r32 >> 1) & 7;
  int coff[2][2];  // [k-step s][plane]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) coff[s][pl] = ((pl * 4 + s * 2 + half) ^ fsw) << 4;
  const int ctile = wave & 1, rtile = wave >> 1;
  const unsigned char* sup_frag = sup + (ctile * 32 + r32) * 128;

  // blend role of this thread: tap p = tid / 5, q chunk (tid % 5) * 12 (12 values; the last chunk holds q = 48 only)
  const int bp = tid / 5, bq0 = (tid - bp * 5) * 12;
  const int bcnt = (tid < 245) ? (bq0 < 48 ? 12 : 1) : 0;
  const int bhx = bp / 7, bwy = bp - bhx * 7;  // first 7-index = x offset, second = y (cotracker3_online.py:102-104)

  for (int tl = 0; tl < nt; ++tl) {
    const FrameTab* tab = tabs + tl;
    const int fw = tab->fw, npx = fw * tab->fh;

    // (1) footprint registers -> LDS (every reader of the previous frame's C / staging is past the barrier of (5)),
    //     then start fetching the next frame
    commit(npx);
    if (tl + 1 < nt) prefetch(tl + 1);
    __syncthreads();

    // (2) C[pixel][q] for my (row tile, column tile); a rare 9-wide footprint has a third row tile (waves 0,1).
    //     Two accumulators (even / odd K-tiles) halve the dependent-MFMA chain.
    auto mma_tile = [&](int row0, f32x16& acc) {
      const unsigned char* arow = fp + (row0 + r32) * 128;
      f32x16 acc_b;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[e] = 0.0f; acc_b[e] = 0.0f; }
      f16x8 ah[NKT][2], al[NKT][2], bh[NKT][2], bl[NKT][2];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          ah[kt][s] = *reinterpret_cast<const f16x8*>(arow + kt * (FROWS * 128) + coff[s][0]);
          al[kt][s] = *reinterpret_cast<const f16x8*>(arow + kt * (FROWS * 128) + coff[s][1]);
          bh[kt][s] = *reinterpret_cast<const f16x8*>(sup_frag + kt * SUP_KT + coff[s][0]);
          bl[kt][s] = *reinterpret_cast<const f16x8*>(sup_frag + kt * SUP_KT + coff[s][1]);
        }
      // A = pixels (rows i), B = support taps (columns j): D[i][j], lane holds column j = lane & 31
#pragma unroll
      for (int kt = 0; kt < NKT; kt += 2)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kt][s], bh[kt][s], acc, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kt + 1][s], bh[kt + 1][s], acc_b, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt][s], bl[kt][s], acc, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt + 1][s], bl[kt + 1][s], acc_b, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt][s], bh[kt][s], acc, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kt + 1][s], bh[kt + 1][s], acc_b, 0, 0, 0);
        }
      acc += acc_b;
    };
    const bool third = npx > 64 && wave < 2;
    f32x16 acc0, acc1;
    mma_tile(rtile * 32, acc0);
    if (third) mma_tile(64, acc1);

    // (3) everyone is done reading the footprint -> overwrite it with C (f32, [96][68])
    __syncthreads();
    float* C = reinterpret_cast<float*>(fp);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // row inside the 32x32 tile
      C[(rtile * 32 + i) * CPITCH + ctile * 32 + r32] = acc0[reg] * UNSCALE;
      if (third) C[(64 + i) * CPITCH + ctile * 32 + r32] = acc1[reg] * UNSCALE;
    }
    __syncthreads();

    // (4) bilinear blend of the 49x49 table: D[p][q] = sum over the 4 corners of w * C[corner pixel][q]
    //     (corner order and weight products of ATen grid_sampler_3d: (x0,y0),(x1,y0),(x0,y1),(x1,y1))
    float* stg = reinterpret_cast<float*>(fp + STG_OFF);
    if (bcnt > 0) {
      const int x0 = tab->fx0[bhx], x1 = tab->fx1[bhx], y0 = tab->fy0[bwy], y1 = tab->fy1[bwy];
      const float wx0 = tab->wx0[bhx], wx1 = tab->wx1[bhx], wy0 = tab->wy0[bwy], wy1 = tab->wy1[bwy];
      const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
      const float* c00 = C + (y0 * fw + x0) * CPITCH + bq0;
      const float* c10 = C + (y0 * fw + x1) * CPITCH + bq0;
      const float* c01 = C + (y1 * fw + x0) * CPITCH + bq0;
      const float* c11 = C + (y1 * fw + x1) * CPITCH + bq0;
      float* dst = stg + bp * CTK_TAPS + bq0;
      if (bcnt == 12) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const f32x4 a = reinterpret_cast<const f32x4*>(c00)[j], b = reinterpret_cast<const f32x4*>(c10)[j];
          const f32x4 c = reinterpret_cast<const f32x4*>(c01)[j], d = reinterpret_cast<const f32x4*>(c11)[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * j + e] = fmaf(d[e], w11, fmaf(c[e], w01, fmaf(b[e], w10, a[e] * w00)));
        }
      } else {
        dst[0] = fmaf(c11[0], w11, fmaf(c01[0], w01, fmaf(c10[0], w10, c00[0] * w00)));
      }
    }
    if (tid < CTK_CORR_LD - CTK_CORR_K) stg[CTK_CORR_K + tid] = 0.0f;  // K padding columns
    __syncthreads();

    // (5) staging row -> SH volume row: 4 consecutive outputs per thread, hi / lo halves
    _Float16* orow = out_base + (long)tl * ROW_H;
    for (int qd = tid; qd < QUADS; qd += 256) {
      f16x4 hi, lo;
      ctk_split4(reinterpret_cast<const f32x4*>(stg)[qd], hi, lo);
      _Float16* dst = orow + ctk_sh_col(qd * 4);
      *reinterpret_cast<f16x4*>(dst) = hi;
      *reinterpret_cast<f16x4*>(dst + 32) = lo;
    }
    __syncthreads();
  }
}

// f32 rows -> SH with a power-of-two scale (pyramid conversion)
__global__ void split_rows_scaled_kernel(const float* x, long n4, float scale, _Float16* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of a [*,128] matrix
  if (i >= n4) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
  v *= scale;
  f16x4 hi, lo;
  ctk_split4(v, hi, lo);
  const long k = i * 4;
  _Float16* dst = out + (k >> 5) * 64 + (k & 31);
  *reinterpret_cast<f16x4*>(dst) = hi;
  *reinterpret_cast<f16x4*>(dst + 32) = lo;
}

}  // namespace

#ifdef CTK_CORR_TIMING
// dev-only (libctk_hip_timing.so): summed s_memtime deltas of wave 0 per phase of corr_volume_sh_kernel
extern "C" int ctk_debug_read_corr_timing(unsigned long long* out16, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_corr_timing), 16 * sizeof(unsigned long long));
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_corr_timing), z, sizeof(z));
  }
  return (int)e;
}
#endif

// SH copy (scaled by 2^8) of one pyramid level of the window: f32 NHWC [S,H,W,128] -> halves [S*H*W][4][2][32]
int ctk_launch_pyramid_split(const float* fmap, long pixels, void* out, hipStream_t s) {
  const long n4 = pixels * (CTK_C / 4);
  CtkProfScope ps("pyramid_split", 0.0, 8.0 * 4.0 * n4, s);
  hipLaunchKernelGGL(split_rows_scaled_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, fmap, n4, FSCALE,
                     static_cast<_Float16*>(out));
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

// Correlation volumes of points [n0, n0+ncount) in SH format: out[l][(n-n0)*S + t][2*CTK_CORR_LD halves].
// fm_sh[l] = ctk_launch_pyramid_split of a->fmaps[l].
int ctk_launch_corr_volume_sh(const ctk_window_args* a, const void* const* fm_sh, int n0, int ncount, void* out,
                              long level_stride_halves, hipStream_t s) {
  if (!a || !out) return CTK_E_NULL;
  if (a->S <= 0 || a->N <= 0) return CTK_E_SHAPE;
  CorrShP p;
  for (int l = 0; l < CTK_LEVELS; ++l) {
    if (!fm_sh[l] || !a->support[l]) return CTK_E_NULL;
    if (a->H[l] <= 0 || a->W[l] <= 0) return CTK_E_SHAPE;
    if (!ctk_aligned16(fm_sh[l]) || !ctk_aligned16(a->support[l])) return CTK_E_ALIGN;
    p.fm[l] = static_cast<const _Float16*>(fm_sh[l]);
    p.support[l] = a->support[l];
    p.H[l] = a->H[l];
    p.W[l] = a->W[l];
    p.sx[l] = ctk_sampler_scale(a->W[l]);
    p.sy[l] = ctk_sampler_scale(a->H[l]);
  }
  if (!a->coords) return CTK_E_NULL;
  p.coords = a->coords;
  p.mask = a->point_mask;
  p.S = a->S;
  p.N = a->N;
  p.out = static_cast<_Float16*>(out);
  p.out_level_stride = level_stride_halves;
  p.n0 = n0;
  p.ncount = ncount;
  p.tchunks = (a->S + TC - 1) / TC;
  const long blocks = (long)ncount * CTK_LEVELS * p.tchunks;
  // algorithmic work per (t,n,level): 2*49*49*128 flop; (2r+2)^2*128*4 B footprint + support/S + volume out
  const double units = (double)ncount * a->S * CTK_LEVELS;
  CtkProfScope ps("corr_volume_sh", units * 2.0 * 49 * 49 * 128,
                  units * (64.0 * 128 * 4 + 49.0 * 128 * 4 / a->S + 2.0 + 2401.0 * 4), s);
  hipLaunchKernelGGL(corr_volume_sh_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
