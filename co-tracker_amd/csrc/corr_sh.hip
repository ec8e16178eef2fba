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
// Workgroup = (point n, level l, <=16 frames), 4 waves, TWO workgroups per CU (LDS 78 KiB each):
//   prologue: the chunk's coordinates and all per-frame tap tables go to LDS; the support patch [49][128] is
//     split once into LDS (K-tile major, XOR-swizzled 16-byte chunks), read into B-fragment registers, and its LDS is
//     reused for the C table and the staging row;
//   the footprint (<= 81 pixels x 512 B of SH data) is PREFETCHED INTO REGISTERS one frame ahead (8 or 12
//     global_load_dwordx4 per thread, 32 lanes per pixel) and committed to a single LDS buffer with the swizzle;
//   the frame loop is software-pipelined over TWO barriers per frame (round 3; it was five):
//     phase A: wave w = (row tile w>>1, column tile w&1) runs 8 k-steps x 3 MFMAs on two accumulators for frame t,
//              and 245 threads blend frame t-1: 12 (or 1) consecutive q of one tap p from the f32 C table into the f32
//              staging row;
//     phase B: accumulators of frame t -> C table; registers of frame t+1 -> footprint buffer, loads of frame t+2 issued;
//              all threads split 8 staged outputs of frame t-1 each and store whole 128-byte lines of the SH volume row
//              (n*S+t), column p*49+q == the reference's (h,w,i,j) flattening (:205); columns 2401..2431 (K padding of
//              corr_mlp.fc1) are zeros.
//     Footprint, C and staging are separate buffers, so nothing aliases inside the loop and a wave carries MFMA work and
//     VALU/LDS work of two different frames between the same pair of barriers.
#include "ctk_common.h"
#include "ctk_options.h"
#include "ctk_profile.h"
#include "gemm_params.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int TC = 16;                       // frames per workgroup
constexpr int FROWS = 88;                    // footprint rows held in LDS (>= 81; the third MFMA row tile reads 8 rows past it)
constexpr int NKT = CTK_C / 32;              // 4 K-tiles
constexpr int SUP_KT = CTK_TAPS * 128;       // bytes per K-tile of the support image: 49 rows x 128 B (rows 49..63 of a
                                             // 64-row MFMA tile read the next K-tile / what follows: unused columns)
constexpr int SUP_BYTES = NKT * SUP_KT;      // 25088
constexpr int FP_BYTES = NKT * FROWS * 128;  // 44 KiB  [ktile][88 rows][128 B]
constexpr int TAB_BYTES = TC * 256 + 128;    // per-frame tap tables + the chunk's coordinates
constexpr int CPITCH = 60;                   // floats per C row: 15 sixteen-byte slots, so the five q chunks (3 slots apart) of
                                             // three neighbouring pixel rows land in 15 different slots of a ds_read_b128 group
constexpr int C_BYTES = FROWS * CPITCH * 4;  // 21120: f32 C[pixel][q] (columns 0..48 are written)
constexpr int STG_BYTES1 = CTK_CORR_LD * 4;  // 9728: f32 staging row of the blended outputs
constexpr float FSCALE = 256.0f;             // both operands are scaled by 2^8 before the f16 split
constexpr float UNSCALE = 1.0f / 65536.0f;
constexpr int LDS1_BYTES = FP_BYTES + C_BYTES + STG_BYTES1 + TAB_BYTES;  // 80128
static_assert(SUP_BYTES + 15 * 128 <= C_BYTES + STG_BYTES1, "the support image (prologue only) aliases C + staging");
static_assert(2 * LDS1_BYTES <= 160 * 1024, "two workgroups per CU");

struct CorrShP {
  const _Float16* fm[CTK_LEVELS];  // SH pyramid of the window, scaled by 2^8: [S][H][W][4][2][32] halves (version 1) or eight planes [4][2][S][H][W][32] (version 3)
  long plane[CTK_LEVELS];          // halves per plane of the version-3 layout (S * H * W * 32)
  const float* support[CTK_LEVELS];
  int H[CTK_LEVELS], W[CTK_LEVELS];
  float sx[CTK_LEVELS], sy[CTK_LEVELS];
  const float* coords;  // [S,N,2]
  const uint8_t* mask;  // [N] or null
  _Float16* out;        // [L][ncount*S][2*CTK_CORR_LD] halves
  long out_level_stride;
  int S, N, n0, ncount, tchunks;
  int map;  // version 3: workgroup -> (point, level) dealing (CTK_OPT_CORR_MAP), see corr_volume_sh3_kernel
};

struct FrameTab {  // per-frame tap table (LDS), 256 bytes
  int fx0[7], fx1[7], fy0[7], fy1[7];       // tap corner columns / rows relative to the footprint origin
  float wx0[7], wx1[7], wy0[7], wy1[7];
  int xb, yb, fw, fh;                        // footprint origin and size (<= 9 x 9)
  int pad[4];
};
static_assert(sizeof(FrameTab) == 256, "FrameTab layout");

__global__ __launch_bounds__(256, 2) void corr_volume_sh_kernel(CorrShP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS1_BYTES];
  unsigned char* fp = lds;
  float* C = reinterpret_cast<float*>(lds + FP_BYTES);
  float* stg = reinterpret_cast<float*>(lds + FP_BYTES + C_BYTES);
  unsigned char* sup = lds + FP_BYTES;  // prologue only (aliases C + staging)
  FrameTab* tabs = reinterpret_cast<FrameTab*>(lds + FP_BYTES + C_BYTES + STG_BYTES1);
  float* cxy = reinterpret_cast<float*>(lds + FP_BYTES + C_BYTES + STG_BYTES1 + TC * 256);  // [TC][2]

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
#pragma unroll
  for (int i = 0; i < 12; ++i) pre[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};  // (rows a path does not load are committed as they are)
  // Round 3: the generic row -> (fy, fx) decode below costs ~19 VALU per load (float divide emulation, fix-ups, 64-bit
  // address arithmetic), 44 % of all the VALU instructions this kernel issued per frame.  A footprint is 8 pixels wide
  // unless a coordinate is an exact integer or the patch hangs over the image border, and then row r = 8 i + 2 wave + half
  // is simply pixel (fy, fx) = (i, 2 wave + half): one uniform base per load (SALU) plus a per-thread constant offset, i.e.
  // global_load_dwordx4 v, v_off, s[base] with no VALU at all.  Rows past the footprint (fh < 8) re-read its last ROW
  // instead of its last pixel; they are never blended.
  const unsigned lane_off = (unsigned)((2 * wave + (lane >> 5)) * (2 * CTK_C) + (lane & 31) * 8) * 2u;  // bytes
  auto prefetch = [&](int tl) {
    const FrameTab* tab = tabs + tl;
    const int fw = tab->fw, fh = tab->fh, npx = fw * fh;
    if (fw == 8) {
      const char* base = reinterpret_cast<const char*>(fm + ((long)(t0 + tl) * H * W + (long)tab->yb * W + tab->xb) * (2 * CTK_C));
      const long pitch = (long)W * (2 * CTK_C) * 2;  // bytes per pyramid row
#pragma unroll
      for (int i = 0; i < 12; ++i) {  // (the same loads are defined as on the generic path: no register shuffling at the join)
        if (i < 8 || npx > 64) {
          const char* rowp = base + (long)min(i, fh - 1) * pitch;  // uniform
          pre[i] = *reinterpret_cast<const f16x8*>(rowp + lane_off);
        }
      }
      return;
    }
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
        if (i < 11 || r < FROWS) *reinterpret_cast<f16x8*>(fp + kt * (FROWS * 128) + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = pre[i];
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
  // The support patch does not change over the chunk's frames: this wave's B fragments (its 32 tap columns, all four
  // K-tiles, hi and lo) are read from LDS ONCE and stay in 64 VGPRs instead of 16 of the 61 ds_read_b128 a wave issued
  // per frame (248 VGPRs, no spill; round 2: 2.78 -> 2.73 ms per launch at C3, same bits).
  f16x8 bh[NKT][2], bl[NKT][2];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bh[kt][s] = *reinterpret_cast<const f16x8*>(sup_frag + kt * SUP_KT + coff[s][0]);
      bl[kt][s] = *reinterpret_cast<const f16x8*>(sup_frag + kt * SUP_KT + coff[s][1]);
    }

  // blend role of this thread: tap p = tid / 5, q chunk (tid % 5) * 12 (12 values; the last chunk holds q = 48 only)
  const int bp = tid / 5, bq0 = (tid - bp * 5) * 12;
  const int bcnt = (tid < 245) ? (bq0 < 48 ? 12 : 1) : 0;
  // Threads are dealt to taps X-FASTEST (bp -> y = bp / 7, x = bp % 7): the corner pixels of neighbouring threads are then
  // neighbouring C rows (68 floats apart = 1 slot of the 16 four-bank slots a ds_read_b128 group can use) instead of rows
  // fw = 8 pixels apart (544 floats = 8 slots: taps k and k + 2 of a 16-lane group collided, a 2-way bank conflict on
  // all 12 reads).  The OUTPUT column of the tap is unchanged: p = x * 7 + y, first 7-index = x offset
  // (cotracker3_online.py:102-104).
  const int bwy = bp / 7, bhx = bp - bwy * 7;
  const int bpo = bhx * 7 + bwy;  // the tap's column block in the volume row

  // every wave has its B fragments: the support image's LDS becomes C + staging
  __syncthreads();
  if (tid < CTK_CORR_LD - CTK_CORR_K) stg[CTK_CORR_K + tid] = 0.0f;  // K padding columns (never touched by the blend)
  commit(tabs[0].fw * tabs[0].fh);
  if (1 < nt) prefetch(1);
  __syncthreads();

  // Bisection of the five-barrier loop this one replaces (profiles/r03_corr_bisect.txt; us per launch at the C3 window):
  // full 2822 | no volume stores 2301 | no footprint loads 2430 | no MFMA phase 2346 | no blend 2554 | no C write 2674 |
  // MFMA phase only 1367 | barriers + commit + split only 774.  The parts ADDED UP: every phase was a short latency chain
  // behind its own barrier with two workgroups per CU to overlap them; a pseudo-random start delay per workgroup changed
  // nothing and 40 % fewer VALU instructions bought 2 %.  Hence fewer, fatter phases.
  f32x16 acc0, acc1;
  for (int tl = 0; tl <= nt; ++tl) {
    const bool cur = tl < nt, prev = tl >= 1;
    const FrameTab* tab = tabs + min(tl, nt - 1);
    const bool third = cur && tab->fw * tab->fh > 64 && wave < 2;

    // ---- phase A ------------------------------------------------------------------------------------------
    // (A1) C[pixel][q] of frame tl for my (row tile, column tile); a rare 9-wide footprint has a third row tile
    //      (waves 0,1).  Two accumulators (even / odd K-tiles) halve the dependent-MFMA chain.
    auto mma_tile = [&](int row0, f32x16& acc) {
      const unsigned char* arow = fp + (row0 + r32) * 128;
      f32x16 acc_b;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[e] = 0.0f; acc_b[e] = 0.0f; }
      f16x8 ah[NKT][2], al[NKT][2];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          ah[kt][s] = *reinterpret_cast<const f16x8*>(arow + kt * (FROWS * 128) + coff[s][0]);
          al[kt][s] = *reinterpret_cast<const f16x8*>(arow + kt * (FROWS * 128) + coff[s][1]);
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
    if (cur) {
      mma_tile(rtile * 32, acc0);
      if (third) mma_tile(64, acc1);
    }

    // (A2) bilinear blend of frame tl-1's 49x49 table: D[p][q] = sum over the 4 corners of w * C[corner pixel][q]
    //      (corner order and weight products of ATen grid_sampler_3d: (x0,y0),(x1,y0),(x0,y1),(x1,y1))
    if (prev && bcnt > 0) {
      const FrameTab* tb = tabs + (tl - 1);
      const int fw = tb->fw;
      const int x0 = tb->fx0[bhx], x1 = tb->fx1[bhx], y0 = tb->fy0[bwy], y1 = tb->fy1[bwy];
      const float wx0 = tb->wx0[bhx], wx1 = tb->wx1[bhx], wy0 = tb->wy0[bwy], wy1 = tb->wy1[bwy];
      const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
      const float* c00 = C + (y0 * fw + x0) * CPITCH + bq0;
      const float* c10 = C + (y0 * fw + x1) * CPITCH + bq0;
      const float* c01 = C + (y1 * fw + x0) * CPITCH + bq0;
      const float* c11 = C + (y1 * fw + x1) * CPITCH + bq0;
      float* dst = stg + bpo * CTK_TAPS + bq0;
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
    __syncthreads();

    // ---- phase B ------------------------------------------------------------------------------------------
    // (B1) accumulators of frame tl -> C (f32, [88][60]; the 49 tap columns only)
    if (cur && ctile * 32 + r32 < CTK_TAPS) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // row inside the 32x32 tile
        C[(rtile * 32 + i) * CPITCH + ctile * 32 + r32] = acc0[reg] * UNSCALE;
        if (third && i < FROWS - 64) C[(64 + i) * CPITCH + ctile * 32 + r32] = acc1[reg] * UNSCALE;
      }
    }
    // (B2) footprint of frame tl+1: registers -> LDS (its loads are the youngest memory operations of this wave, so the
    //      wait does not cover the stores below); (B4) then fetch frame tl+2
    if (tl + 1 < nt) commit(tabs[tl + 1].fw * tabs[tl + 1].fh);
    // (B3) staging row of frame tl-1 -> SH volume row: FULL 128-byte lines per 8 lanes.  A CU retires a store instruction only
    //      every ~100 cycles whatever its width (tools/gemm_lab.cpp store experiments): lane (line, chunk c) writes 16 bytes --
    //      hi halves of 8 outputs for c < 4, their lo halves for c >= 4 (both lanes of an octet split the same 8 values) --
    //      76 lines x 8 lanes = 10 wave-instructions per frame, each a run of whole lines.
    if (prev) {
      _Float16* orow = out_base + (long)(tl - 1) * ROW_H;
      constexpr int NIDX = (CTK_CORR_LD / 32) * 8;  // 608 sixteen-byte pieces = 2 full rounds of the workgroup + 96 threads
      f32x4 va[3], vb[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // all staging reads first (one LDS round trip instead of three)
        const int idx = min(tid + 256 * k, NIDX - 1), oct = (idx >> 3) * 4 + (idx & 3);
        va[k] = reinterpret_cast<const f32x4*>(stg)[2 * oct];
        vb[k] = reinterpret_cast<const f32x4*>(stg)[2 * oct + 1];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = tid + 256 * k, line = idx >> 3, c = idx & 7;
        if (k == 2 && wave >= 2) break;  // the last 96 pieces belong to waves 0 and 1
        f16x8 hi, lo;
        ctk_split8(va[k], vb[k], hi, lo);
        if (idx < NIDX) *reinterpret_cast<f16x8*>(orow + line * 64 + c * 8) = (c < 4) ? hi : lo;
      }
    }
    if (tl + 2 < nt) prefetch(tl + 2);
    __syncthreads();
  }
}

// (Version 2 of the sampler -- one WAVE per frame, the blend as a second MFMA -- lived here in rounds 1-5: parity-green, 2.98 ms per
// launch against 2.52 / 2.30 ms for versions 1 / 3; its SQ-counter post-mortem is in profiles/r01_corr_v1_v2_pmc.txt and the code in
// git show 360f4f4:co-tracker_amd/csrc/corr_sh.hip.  Round 6 removed it from the library.)

// ---------------------------------------------------------------------------------------------------------
// Version 3 (round 5): the footprint never touches LDS and the frame loop has ONE barrier per frame.
//   * MFMA shape 16x16x32: wave w owns footprint rows 16 w .. 16 w + 15 (one pixel per lane & 15) against all 49 (64) tap
//     columns, so a footprint row is needed by exactly one wave and its A fragments come STRAIGHT from the SH pyramid:
//     lane (pixel i = lane & 15, k group g = lane >> 4) reads 16 bytes of plane pl of K-tile kt at
//     pixel * 512 + kt * 128 + pl * 64 + g * 16 -- 8 loads per lane and frame (what version 1 issued to fill its LDS copy),
//     no commit, no fragment ds_reads, no 44 KiB footprint buffer.  The support (B) fragments of all four 16-column tiles stay
//     in 128 VGPRs over the chunk's frames; both operands use the same (g, element) -> k assignment, which is all a
//     contraction needs.
//   * C and the staging row are double buffered (2 x 21 KiB + 2 x 9.5 KiB), so between two barriers a wave carries three
//     independent pieces of work of three different frames: store(t-2): staging[t & 1] -> split -> SH volume row;
//     blend(t-1): C[(t-1) & 1] -> staging[(t-1) & 1]; MFMA(t) -> C[t & 1]; then the loads of frame t+1 (a whole iteration
//     ahead of their use).  Version 1 needed two barriers per frame (C and staging single buffered) and parked its waves at
//     them for 37 % of their lifetime (profiles/r04_pmc_corr.txt).
//   * a footprint of more than 64 pixels (9 wide or tall: integer coordinates) has up to two more row tiles: waves 0 and 1
//     load them, exposed, behind their own tile (rare).
// Same values as version 1 up to f32 summation order (four accumulators per wave instead of even / odd K-tile pairs).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
constexpr int LDS3_BYTES = 2 * C_BYTES + 2 * STG_BYTES1 + TAB_BYTES;  // 65920
static_assert(SUP_BYTES + 15 * 128 <= 2 * C_BYTES, "the support image (prologue only) aliases the C tables");
static_assert(2 * LDS3_BYTES <= 160 * 1024, "two workgroups per CU");

// DBG (bisection bits; instantiated with DBG != 0 only in dev builds, make dev + CTK_CORR_DBG): 1 = no volume stores, 2 = every lane reads pixel 0 (no footprint traffic), 16 = no MFMAs,
// 32 = no blend, 64 = no store phase at all (no staging reads, no splits)
template <int DBG>
__global__ __launch_bounds__(256, 2) void corr_volume_sh3_kernel(CorrShP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS3_BYTES];
  float* Cb = reinterpret_cast<float*>(lds);                                  // [2][FROWS][CPITCH]
  float* stgb = reinterpret_cast<float*>(lds + 2 * C_BYTES);                  // [2][CTK_CORR_LD]
  unsigned char* sup = lds;                                                   // prologue only (aliases the C tables)
  FrameTab* tabs = reinterpret_cast<FrameTab*>(lds + 2 * C_BYTES + 2 * STG_BYTES1);
  float* cxy = reinterpret_cast<float*>(lds + 2 * C_BYTES + 2 * STG_BYTES1 + TC * 256);  // [TC][2]

  unsigned bid = ctk_xcd_remap(blockIdx.x, gridDim.x);
  int tc, lvl, nl;
  if (p.map == 3) {  // level-major (the default): consecutive ids = consecutive points of ONE level
    nl = bid % p.ncount;
    bid /= p.ncount;
    lvl = bid % CTK_LEVELS;
    tc = bid / CTK_LEVELS;
  } else if (p.map == 4) {  // point-major inside blocks of 16 points: (16 points x level) x 4 levels, then the next 16 points
    const unsigned blk = bid / (16 * CTK_LEVELS * p.tchunks), r = bid % (16 * CTK_LEVELS * p.tchunks);
    if ((blk + 1) * 16 <= (unsigned)p.ncount) {
      nl = blk * 16 + (r & 15);
      lvl = (r >> 4) % CTK_LEVELS;
      tc = (r >> 4) / CTK_LEVELS;
    } else {  // the ragged tail: plain order
      const unsigned t = bid - blk * 16 * CTK_LEVELS * p.tchunks;
      tc = t % p.tchunks;
      lvl = (t / p.tchunks) % CTK_LEVELS;
      nl = blk * 16 + t / (p.tchunks * CTK_LEVELS);
    }
  } else if (p.map != 0 && (p.ncount & 1) == 0) {
    // Neighbouring points (consecutive indices of a grid query) share most of their footprints; dealt as PAIRS at the same level to
    // workgroups that run on one CU at the same time, the second one's loads can hit the first one's lines in the 32 KiB L1.
    // map 1: the pair is (b, b + 32) inside a block of 64 consecutive workgroup ids; map 2: the pair is (b, b + 1).
    unsigned u = bid;
    if (p.map == 1 && (bid / 64 + 1) * 64 <= gridDim.x) u = (bid / 64) * 64 + (bid & 31) * 2 + ((bid >> 5) & 1);
    const unsigned member = u & 1;
    unsigned v = u >> 1;
    tc = v % p.tchunks;
    v /= p.tchunks;
    lvl = v % CTK_LEVELS;
    nl = (v / CTK_LEVELS) * 2 + member;
  } else {
    tc = bid % p.tchunks;
    bid /= p.tchunks;
    lvl = bid % CTK_LEVELS;
    nl = bid / CTK_LEVELS;  // local point index
  }
  const int n = p.n0 + nl;
  const int t0 = tc * TC;
  const int nt = min(TC, p.S - t0);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i16 = lane & 15, g4 = lane >> 4;
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
  const long plane = p.plane[lvl];
  // ---- prologue (as version 1): coordinates -> LDS; support patch -> split, scaled, swizzled image; tap tables ----
  if (tid < 2 * nt) cxy[tid] = p.coords[((long)(t0 + (tid >> 1)) * p.N + n) * 2 + (tid & 1)];
  {
    const float* sp = p.support[lvl] + (long)n * CTK_TAPS * CTK_C;
    f32x4 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int i = min(tid + 256 * j, CTK_TAPS * 32 - 1);
      v[j] = *reinterpret_cast<const f32x4*>(sp + i * 4);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int i = tid + 256 * j, row = i >> 5, c4 = i & 31;
      if (i < CTK_TAPS * 32) {
        f16x4 hi, lo;
        ctk_split4(v[j] * FSCALE, hi, lo);
        const int kt = c4 >> 3, k8 = (c4 & 7) >> 1, sub = c4 & 1;
        const int fs = (row >> 1) & 7;
        unsigned char* base = sup + kt * SUP_KT + row * 128 + sub * 8;
        *reinterpret_cast<f16x4*>(base + ((k8 ^ fs) << 4)) = hi;
        *reinterpret_cast<f16x4*>(base + (((4 + k8) ^ fs) << 4)) = lo;
      }
    }
  }
  __syncthreads();
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

  // B fragments: tap column ct * 16 + i16, K-tile kt, k group g4 (chunk pl * 4 + g4 of the row's 128-byte line).  Rows 49..63
  // of the last column tile read what follows the 49 rows of a K-tile: columns nobody stores.
  f16x8 bh[NKT][4], bl[NKT][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int row = ct * 16 + i16, fs = (row >> 1) & 7;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      bh[kt][ct] = *reinterpret_cast<const f16x8*>(sup + kt * SUP_KT + row * 128 + ((g4 ^ fs) << 4));
      bl[kt][ct] = *reinterpret_cast<const f16x8*>(sup + kt * SUP_KT + row * 128 + (((4 + g4) ^ fs) << 4));
    }
  }
  __syncthreads();  // tap tables visible; every wave has its B fragments: the image's LDS becomes the C tables
  if (tid < CTK_CORR_LD - CTK_CORR_K) {  // K padding columns of both staging rows (never touched by the blend)
    stgb[CTK_CORR_K + tid] = 0.0f;
    stgb[CTK_CORR_LD + CTK_CORR_K + tid] = 0.0f;
  }

  // ---- A fragments of footprint row tile `tile` of frame tl: one pixel per lane & 15, 8 x 16 bytes -------------
  f16x8 ah[NKT], al[NKT];
  auto load_a = [&](int tl, int tile) {
    const FrameTab* tab = tabs + tl;
    const int fw = tab->fw, npx = fw * tab->fh;
    const int r = min(tile * 16 + i16, npx - 1);  // rows past the footprint re-read its last pixel: never blended
    int fy = (int)((float)r * __builtin_amdgcn_rcpf((float)fw));  // r / fw (r < 96, fw <= 9; 1-ulp reciprocal), fixed up below
    fy -= (fy * fw > r);
    fy += ((fy + 1) * fw <= r);
    const int fx = r - fy * fw;
    const _Float16* px = fm + (((long)(t0 + tl) * H + tab->yb + fy) * W + tab->xb + fx) * 32 + g4 * 8;  // plane 0 (K-tile 0, hi) of my pixel
    if (DBG & 2) px = fm + g4 * 8;
    // (Compiler-managed loads.  Inline-asm loads with counted waits that leave the three younger volume stores in flight were
    // tried: not faster, and not safe -- nothing documents that a store cannot retire before an older load, so vmcnt(N) must
    // not be used to skip stores.)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      ah[kt] = *reinterpret_cast<const f16x8*>(px + (2 * kt) * plane);
      al[kt] = *reinterpret_cast<const f16x8*>(px + (2 * kt + 1) * plane);
    }
  };
  // C[pixel][tap] of one row tile -> table Cw (f32, [88][60]; the 49 tap columns only).  Between two MFMAs on the same
  // accumulator sit the three other column tiles.
  auto mma_tile = [&](int tile, float* Cw) {
    f32x4v acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (DBG & 16) {  // (keeps the loads alive)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct][0] += (float)(ah[kt][ct] + al[kt][ct]);
        continue;
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kt], bh[kt][ct], acc[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bl[kt][ct], acc[ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bh[kt][ct], acc[ct], 0, 0, 0);
    }
    // D: lane holds column i16 of the tile's rows 4 g4 + reg
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      if (ct * 16 + i16 < CTK_TAPS) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = tile * 16 + 4 * g4 + reg;
          if (tile < 4 || row < FROWS) Cw[row * CPITCH + ct * 16 + i16] = acc[ct][reg] * UNSCALE;
        }
      }
  };

  // blend role of this thread (as version 1): tap p = tid / 5 dealt x-fastest, q chunk (tid % 5) * 12.  Threads 245..255 repeat
  // tap 48's reads and write nothing.
  const int bp = min(tid / 5, CTK_TAPS - 1), bq0 = (tid - (tid / 5) * 5) * 12;
  const int bcnt = (tid < 245) ? (bq0 < 48 ? 12 : 1) : 0;
  const int bwy = bp / 7, bhx = bp - bwy * 7;
  const int bpo = bhx * 7 + bwy;  // the tap's column block in the volume row
  // store role: the 608 sixteen-byte pieces of a volume row are dealt 152 (19 whole lines) to a wave: two full instructions and
  // one of 24 lanes
  const int sp0 = 152 * wave + lane;  // pieces sp0, sp0 + 64 and (lane < 24) sp0 + 128

  // One iteration: MFMA(tl) -> C, the loads of frame tl+1 | store(tl-2) | blend(tl-1).  ST / BL / CUR are compile-time in the
  // steady loop, so its body is (apart from the lane-masked writes) one straight line.
  auto step = [&](int tl, bool ST, bool BL, bool CUR) {
    if (DBG & 32) BL = false;
    if (DBG & 64) ST = false;
    // -- LDS reads of the two older frames first
    f32x4 va[3], vb[3];
    if (ST) {
      const float* stg = stgb + (tl & 1) * CTK_CORR_LD;  // (tl - 2) & 1
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = min(sp0 + 64 * k, 152 * wave + 151), oct = (idx >> 3) * 4 + (idx & 3);
        va[k] = reinterpret_cast<const f32x4*>(stg)[2 * oct];
        vb[k] = reinterpret_cast<const f32x4*>(stg)[2 * oct + 1];
      }
    }
    // The four corner weights as (w, w) register pairs the compiler cannot look into: the packed FMAs below then take plain
    // operands.  Left alone, hipcc packs (w00, w10) / (w01, w11) into pairs and broadcasts with op_sel:[0,1,0] (low result lane
    // reading the HIGH half of a source); on gfx950 a ds_write2_b32 issued right behind such a v_pk_fma_f32 stored a stale first
    // data register in lanes 48..63 -- intermittently when the LDS queue was busy, every time when it was idle (round 5:
    // profiles/r05_sampler_v3_pk_hazard.txt).  The op_sel_hi-only and plain forms never did.
    f32x2v W00 = {0.0f, 0.0f}, W10 = W00, W01 = W00, W11 = W00;
    const float *c00 = Cb, *c10 = Cb, *c01 = Cb, *c11 = Cb;
    if (BL) {
      const FrameTab* tb = tabs + (tl - 1);
      const float* C = Cb + ((tl - 1) & 1) * (C_BYTES / 4);
      const int fw = tb->fw;
      const int x0 = tb->fx0[bhx], x1 = tb->fx1[bhx], y0 = tb->fy0[bwy], y1 = tb->fy1[bwy];
      const float wx0 = tb->wx0[bhx], wx1 = tb->wx1[bhx], wy0 = tb->wy0[bwy], wy1 = tb->wy1[bwy];
      const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
      W00 = f32x2v{w00, w00}; W10 = f32x2v{w10, w10}; W01 = f32x2v{w01, w01}; W11 = f32x2v{w11, w11};
      asm volatile("" : "+v"(W00), "+v"(W10), "+v"(W01), "+v"(W11));
      c00 = C + (y0 * fw + x0) * CPITCH + bq0;
      c10 = C + (y0 * fw + x1) * CPITCH + bq0;
      c01 = C + (y1 * fw + x0) * CPITCH + bq0;
      c11 = C + (y1 * fw + x1) * CPITCH + bq0;
    }
    // -- MFMAs of frame tl (its A fragments were requested a whole iteration ago), then the request for frame tl+1
    if (CUR) {
      float* Cw = Cb + (tl & 1) * (C_BYTES / 4);
      mma_tile(wave, Cw);
      const int npx = tabs[tl].fw * tabs[tl].fh;
      if (npx > 64 && wave < 2 && 64 + 16 * wave < npx) {  // rare: rows 64..80
        load_a(tl, 4 + wave);
        mma_tile(4 + wave, Cw);
      }
      if (tl + 1 < nt) load_a(tl + 1, wave);
    }
    // -- store(tl-2): staging row -> SH volume row, whole 128-byte lines per 8 lanes (see version 1)
    if (ST && !(DBG & 1)) {
      _Float16* orow = out_base + (long)(tl - 2) * ROW_H;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = sp0 + 64 * k, line = idx >> 3, c = idx & 7;
        f16x8 hi, lo;
        ctk_split8(va[k], vb[k], hi, lo);
        const f16x8 v = (c < 4) ? hi : lo;
        if (k < 2 || lane < 24) *reinterpret_cast<f16x8*>(orow + line * 64 + c * 8) = v;
      }
    }
    // -- blend(tl-1): D[p][q] = sum over the 4 corners of w * C[corner pixel][q] (corner order and weight products of ATen
    //    grid_sampler_3d: (x0,y0),(x1,y0),(x0,y1),(x1,y1))
    if (BL) {
      float* dst = stgb + ((tl - 1) & 1) * CTK_CORR_LD + bpo * CTK_TAPS + bq0;
#if defined(CTK_PK_NOP)
      // Round-6 hazard experiment (tools/pk_nop_experiment.sh; never defined in a library build): the control flow of round 5's
      // deterministic failing build (profiles/r05_sampler_v3_pk_hazard_variants.patch, CTK_CORR_DBG=8192: element-wise FMAs with
      // the (w, w) pairs indexed crosswise -> hipcc emits v_pk_fma_f32 ... op_sel:[0,1,0] in front of ds_write2_b32), with
      // the wait states themselves (s_nop) are inserted into the ISA by the tool's post-pass: any inline asm tied to the results makes
      // hipcc drop the packed form, and one that is not tied is scheduled elsewhere.
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const f32x4 a = reinterpret_cast<const f32x4*>(c00)[j], b = reinterpret_cast<const f32x4*>(c10)[j];
        const f32x4 c = reinterpret_cast<const f32x4*>(c01)[j], d = reinterpret_cast<const f32x4*>(c11)[j];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(d[e], W11[e & 1], fmaf(c[e], W01[e & 1], fmaf(b[e], W10[e & 1], a[e] * W00[e & 1])));
        if (bcnt == 12) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * j + e] = o[e];
        } else if (bcnt == 1 && j == 0) {
          dst[0] = o[0];
        }
      }
#else
      if (bcnt == 12) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const f32x4 a = reinterpret_cast<const f32x4*>(c00)[j], b = reinterpret_cast<const f32x4*>(c10)[j];
          const f32x4 c = reinterpret_cast<const f32x4*>(c01)[j], d = reinterpret_cast<const f32x4*>(c11)[j];
#pragma unroll
          for (int h = 0; h < 2; ++h) {  // outputs 4 j + 2 h, + 1 as one register pair (and one ds_write2_b32)
            const f32x2v a2 = {a[2 * h], a[2 * h + 1]}, b2 = {b[2 * h], b[2 * h + 1]};
            const f32x2v c2 = {c[2 * h], c[2 * h + 1]}, d2 = {d[2 * h], d[2 * h + 1]};
            const f32x2v o2 = __builtin_elementwise_fma(d2, W11, __builtin_elementwise_fma(c2, W01, __builtin_elementwise_fma(b2, W10, a2 * W00)));
            dst[4 * j + 2 * h] = o2[0];
            dst[4 * j + 2 * h + 1] = o2[1];
          }
        }
      } else if (bcnt == 1) {  // q = 48
        dst[0] = fmaf(c11[0], W11[0], fmaf(c01[0], W01[0], fmaf(c10[0], W10[0], c00[0] * W00[0])));
      }
#endif
    }
    __syncthreads();
  };

  load_a(0, wave);
  __syncthreads();  // staging pads written (iteration 0 itself only writes C[0])
  int tl = 0;
  for (; tl < 2; ++tl) step(tl, false, tl >= 1 && tl <= nt, tl < nt);
  for (; tl < nt; ++tl) step(tl, true, true, true);
  for (; tl <= nt + 1; ++tl) step(tl, true, tl <= nt, false);
}

// f32 rows -> SH with a power-of-two scale (pyramid conversion).  PLANES = false: the SH row format [pixel][4 K-tiles][hi | lo][32]
// (version 1 reads a pixel's 512 contiguous bytes); PLANES = true (version 3, round 6): eight planes [K-tile][hi | lo][pixel][32] of
// 64 bytes per pixel -- the 16 lanes x 4 k-groups of one of version 3's fragment loads then read the 64-byte pieces of 16
// CONSECUTIVE pixels (two footprint rows of 8: two contiguous 512-byte runs = 8-10 cache lines) instead of 64 bytes out of each of
// 16 different 512-byte pixel records (16 lines, half of each used).
template <bool PLANES>
__global__ void split_rows_scaled_kernel(const float* x, long n4, float scale, _Float16* out, long plane_halves) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of a [*,128] matrix
  if (i >= n4) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
  v *= scale;
  f16x4 hi, lo;
  ctk_split4(v, hi, lo);
  const long k = i * 4;
  if (PLANES) {
    const long pix = k >> 7;
    const int c = (int)(k & 127), kt = c >> 5;
    _Float16* dst = out + (long)(2 * kt) * plane_halves + pix * 32 + (c & 31);
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + plane_halves) = lo;
  } else {
    _Float16* dst = out + (k >> 5) * 64 + (k & 31);
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + 32) = lo;
  }
}

}  // namespace


// SH copy (scaled by 2^8) of one pyramid level of the window: f32 NHWC [S,H,W,128] -> halves [S*H*W][4][2][32] (sampler version 1)
// or [4][2][S*H*W][32] (version 3).  `version` must be the one later handed to ctk_launch_corr_volume_sh for this copy.
int ctk_launch_pyramid_split(const float* fmap, long pixels, void* out, int version, hipStream_t s) {
  const long n4 = pixels * (CTK_C / 4);
  CtkProfScope ps("pyramid_split", 0.0, 8.0 * 4.0 * n4, s);
  if (version == 3)
    hipLaunchKernelGGL(split_rows_scaled_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, fmap, n4, FSCALE,
                       static_cast<_Float16*>(out), pixels * 32);
  else
    hipLaunchKernelGGL(split_rows_scaled_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, fmap, n4, FSCALE,
                       static_cast<_Float16*>(out), 0L);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

// Correlation volumes of points [n0, n0+ncount) in SH format: out[l][(n-n0)*S + t][2*CTK_CORR_LD halves].
// fm_sh[l] = ctk_launch_pyramid_split of a->fmaps[l].
int ctk_launch_corr_volume_sh(const ctk_window_args* a, const void* const* fm_sh, int n0, int ncount, void* out,
                              long level_stride_halves, int version, hipStream_t s) {
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
    p.plane[l] = (long)a->S * a->H[l] * a->W[l] * 32;
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
  // CTK_OPT_CORR_VERSION (include/ctk.h): 3 = wave-owned footprint rows (default since round 5), 1 = the two-barrier LDS-footprint
  // kernel.  Measured on MI355X at the C3 window (tools/bench_corr.py, round 5): version 1 2.52-2.62 ms, version 3 2.25-2.48 ms.
  // CTK_OPT_CORR_MAP: workgroup -> (point, level) dealing of version 3.  Default 3 = level-major (consecutive workgroup ids, which
  // the XCD remap keeps on one XCD, are consecutive points of ONE level: a grid query's neighbours share 60-90 % of their footprints,
  // so the second one finds the lines in L1 / L2): 2.47 -> 2.25 ms per launch at the C3 window; 0 = point-major (levels innermost,
  // the order of version 1), 1 / 2 = point pairs, 4 = blocks of 16 points (profiles/r05_sampler_v3_bisect.txt).
  p.map = ctk_opt(CTK_OPT_CORR_MAP);
  const int v = version;  // (the caller read CTK_OPT_CORR_VERSION once, for the pyramid copy and for this launch)
  if (v == 3) {
#ifdef CTK_DEV
    // dev build only (make dev): CTK_CORR_DBG = bisection bits of version 3 (they change the RESULT: tools/bench_corr.py)
    static const int dbg = [] { const char* e = getenv("CTK_CORR_DBG"); return e ? atoi(e) : 0; }();
    switch (dbg) {
#define CTK_SH3(D) case D: hipLaunchKernelGGL(corr_volume_sh3_kernel<D>, dim3((unsigned)blocks), dim3(256), 0, s, p); break
      CTK_SH3(1); CTK_SH3(2); CTK_SH3(3); CTK_SH3(16); CTK_SH3(32); CTK_SH3(48); CTK_SH3(64); CTK_SH3(112);
#undef CTK_SH3
      default: hipLaunchKernelGGL(corr_volume_sh3_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    }
#else
    hipLaunchKernelGGL(corr_volume_sh3_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, p);
#endif
  } else {
    hipLaunchKernelGGL(corr_volume_sh_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  }
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
