// The encoder's convolutions (BasicEncoder, cotracker/models/core/cotracker/blocks.py:141-219; called at
// cotracker3_online.py:373-384) as IMPLICIT GEMMs on the split-half MFMA path -- no im2col matrix, no MIOpen.
//
//   out[(f, oy, ox)][n] = bias[n] + sum_{ky,kx,c} in[f][oy*s + ky - pad][ox*s + kx - pad][c] * W[n][ky][kx][c]
//
// Activations live NHWC in SH format, [F][H][W][C/32] lines of 128 bytes (32 hi | 32 lo halves): one K-tile of the GEMM
// = one (tap, 32-channel group) = ONE such line per output pixel, which is exactly the LDS-DMA granule of gemm_pp.hip.
// So the A operand of a 256-row tile is fetched by 32 global_load_lds pieces per K-tile whose per-lane source address is
// the tap-shifted pixel's line (or a 128-byte line of zeros for the padding ring); weights are repacked once to
// [N][ky][kx][c] and split by ctk_pack_weight; everything behind the LDS ring -- fragments, the 3 x f16 MFMA products,
// the ping-pong of the two wave groups, the LDS-transposed full-line stores -- is gemm_pp.hip's.
//
// One tile shape for all 22 convolutions: 256 output pixels x 128 output channels (waves 4 x 2, wave tile 64 x 64: two
// column phases of 12 MFMAs per K-tile).  Output widths 64 and 96 run on zero-padded weight rows (their columns are not
// stored); 256 outputs (conv2) are two column blocks.  Ring = 3 K-tiles of 48 KiB: both blocks of K-tile J+2 are
// requested while K-tile J is multiplied (2-3 phases ahead of their first read), K-tile J-1's slot is the one they land
// in (its last read is 2 phases old).  The DMA stream does NOT run across tiles here (a convolution tile has 18-117
// K-tiles; the per-lane pixel decode stays per tile), so the ring is idle at a tile's end and serves as the epilogue's
// transpose scratch.
#include "pp_common.h"
#include "ctk_options.h"
#include <cstdlib>

namespace {

struct CtkConvP {
  CtkGemmP g;              // A = SH input, M = F*Hout*Wout, N = padded width (multiple of 128), K = KH*KW*Cin, C/ldc, bias, Wp
  int F, Hin, Win, CL;     // input frames / size / 32-channel groups per pixel
  int Hout, Wout, KH, KW, stride, pad;
  int n_valid;             // stored output columns
  const unsigned char* zeros;  // >= 128 bytes of zeros (padding ring)
};

constexpr int CV_SLOT = 49152;          // one K-tile: A 256 rows (32 KiB) | B_0 64 rows | B_1 64 rows
constexpr int CV_RING = 3 * CV_SLOT;    // 144 KiB
constexpr int CV_BIAS = 147456;         // bias (<= 4 KiB) behind the ring

template <int EPI>
__global__ __launch_bounds__(512) void conv_pp128_kernel(CtkConvP p, int tiles_total, const bool conv_force_ph1) {
  constexpr bool DBG = false;
  constexpr int BM = 256, BN = 128;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS_ALL];
  static_assert(CV_BIAS + 4096 <= PP_LDS_ALL, "LDS budget");
  const CtkGemmP& g = p.g;
  const int dbg = 0;
  const int KT = g.K / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  unsigned jctr = 0;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const unsigned l3 = lane >> 3, l4 = lane >> 4, l7 = lane & 7;

  const float w_unscale = reinterpret_cast<const float*>(g.Wp)[1];
  const float* bias_lds = reinterpret_cast<const float*>(lds + CV_BIAS);
  if ((EPI & 32) != 0) {
    for (int i = tid; i < g.N / 4; i += 512) reinterpret_cast<f32x4*>(lds + CV_BIAS)[i] = reinterpret_cast<const f32x4*>(g.bias)[i];
    __syncthreads();
  }

  const unsigned char* in = static_cast<const unsigned char*>(g.A);
  const unsigned char* wsh = reinterpret_cast<const unsigned char*>(g.Wp) + PP_HDR_BYTES;
  const unsigned ldw_b = (unsigned)KT * 128;
  const int hw_out = p.Hout * p.Wout;
  const int taps_w = p.KW, cl = p.CL;

  // block-piece q (0..23) of the second block of a K-tile: A pieces 24..31 | B_0 pieces 0..7 | B_1 pieces 0..7
  auto cbyte = [&](const int piece) { return ((l7 ^ ((4 * piece + l4) & 7)) << 4); };

  // fragment addressing (as gemm_pp192_kernel): [j][plane], slot offset added per K-tile
  const int fsw = (r32 >> 1) & 7;
  unsigned a_rd0[2][2], b_rd0[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const unsigned co = (unsigned)(((pl * 4 + j * 2 + half) ^ fsw) << 4);
      a_rd0[j][pl] = (wm * 64 + r32) * 128 + co;
      b_rd0[j][pl] = 32768 + (wn * 32 + r32) * 128 + co;
    }

  f32x16 acc[2][2];
  f16x8 fa[2][2][2], fb[2][2];

  for (int q = 0;; ++q) {  // my tiles
    const int G = gridDim.x, first = q * G;
    if (first >= tiles_total) break;
    const int n_r = min(G, tiles_total - first);
    if ((int)blockIdx.x >= n_r) break;
    unsigned tile = first + ctk_xcd_remap(blockIdx.x, n_r);
    const int nb = tile % g.nblocks;
    const int mb = tile / g.nblocks;
    const int m0 = mb * BM, n0 = nb * BN;
    const bool ph1 = (p.n_valid - n0 > 64) || conv_force_ph1;  // workgroup-uniform: does the second 64-column phase carry stored columns?

    // ---- per-lane decode of my A rows: first block pieces 3w+e (e < 3), second block pieces 24 + 3w + e while 3w + e < 8
    int pix[6], iyx[6];  // pixel index of (f, 0, 0) in the input; packed (iy0 + 0x4000) << 16 | (ix0 + 0x4000)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int piece = k < 3 ? 3 * wave + k : 24 + 3 * wave + (k - 3);
      const int r = min(m0 + 8 * piece + (int)l3, g.M - 1);
      const int f = r / hw_out, rem = r - f * hw_out;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      pix[k] = f * p.Hin * p.Win;
      iyx[k] = ((oy * p.stride - p.pad + 0x4000) << 16) | (ox * p.stride - p.pad + 0x4000);
    }
    auto a_src = [&](const int k, const int piece, const int ky, const int kx, const int ct) -> const unsigned char* {
      const int iy = (iyx[k] >> 16) - 0x4000 + ky, ix = (iyx[k] & 0xffff) - 0x4000 + kx;
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      const long off = ((long)(pix[k] + iy * p.Win + ix) * cl + ct) * 128;
      return (ok ? in + off : p.zeros) + cbyte(piece);
    };
    // K-tile kt = (tap, channel group): tap-major, channel group fastest (the weight repack order)
    auto issue_i0 = [&](int kt, const int slot) {  // A pieces 3w .. 3w+2
      kt = min(kt, KT - 1);
      const int tap = kt / cl, ct = kt - tap * cl, ky = tap / taps_w, kx = tap - ky * taps_w;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int piece = 3 * wave + e;
        __builtin_amdgcn_global_load_lds((pp_gptr)a_src(e, piece, ky, kx, ct), (pp_lptr)(lds + slot * CV_SLOT + piece * 1024), 16, 0, 0);
      }
    };
    auto issue_i1 = [&](int kt, const int slot) {  // block pieces 3w .. 3w+2 of [A 24..31 | B_0 | B_1]
      kt = min(kt, KT - 1);
      const int tap = kt / cl, ct = kt - tap * cl, ky = tap / taps_w, kx = tap - ky * taps_w;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int bq = 3 * wave + e;
        if (bq < 8) {
          const int piece = 24 + bq;
          __builtin_amdgcn_global_load_lds((pp_gptr)a_src(3 + e, piece, ky, kx, ct), (pp_lptr)(lds + slot * CV_SLOT + piece * 1024), 16, 0, 0);
        } else {
          const int n = (bq - 8) >> 3, piece = (bq - 8) & 7;  // B_n piece: W rows n0 + n*64 + 8*piece + l3
          const unsigned char* src = wsh + (long)(n0 + n * 64 + 8 * piece + (int)l3) * ldw_b + (long)kt * 128 + cbyte(piece);
          __builtin_amdgcn_global_load_lds((pp_gptr)src, (pp_lptr)(lds + slot * CV_SLOT + 32768 + n * 8192 + piece * 1024), 16, 0, 0);
        }
      }
    };

    // ---- prologue of the tile: K-tiles 0 and 1 requested, K-tile 0 waited for
    issue_i0(0, 0);
    issue_i1(0, 0);
    issue_i0(1, 1);
    issue_i1(1, 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    PP_WAIT_VM(6);  // the 6 pieces of K-tile 1 may still be in flight
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();  // stagger in

    int slot = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const int slot2 = slot == 0 ? 2 : slot - 1;  // slot of K-tile kt + 2 (= of K-tile kt - 1)
      const unsigned so = slot * CV_SLOT;
      // ---- phase 0: read A, B_0; request the A rows 0..191 of K-tile kt+2
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa[mi][j][pl] = *reinterpret_cast<const f16x8*>(lds + so + a_rd0[j][pl] + mi * 4096);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(lds + so + b_rd0[j][pl]);
      issue_i0(kt + 2, slot2);
      PP_BARRIER();
      PP_WAIT_LGKM0();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][term == 0 ? 1 : 0], fa[mi][j][term == 1 ? 1 : 0], acc[mi][0], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      PP_BARRIER();
      // ---- phase 1: read B_1; request the rest of K-tile kt+2; K-tile kt+1 must have landed.
      //      Round 4: a tile whose columns 64..127 are all zero-padded weight rows (the 64-channel layers: the stem and the four
      //      convolutions of layer1, the largest-M launches of the encoder) skips this phase's fragment reads and its 12 MFMAs --
      //      their products were multiplied by zero weights and never stored.  The DMA requests, waits and barriers stay exactly
      //      as they are (every wave keeps issuing 6 pieces per K-tile: the counted vmcnt waits depend on it), so the schedule
      //      tools/check_pp_schedule.py proves is unchanged; same bits out.
      if (ph1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(lds + so + b_rd0[j][pl] + 8192);
      }
      issue_i1(kt + 2, slot2);
      PP_WAIT_VM(6);
      PP_BARRIER();
      if (ph1) {
        PP_WAIT_LGKM0();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][term == 0 ? 1 : 0], fa[mi][j][term == 1 ? 1 : 0], acc[mi][1], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      if (kt + 1 < KT) PP_BARRIER();
      slot = slot == 2 ? 0 : slot + 1;
    }
    // ---- tile end: drop the stagger, drain the duplicate tail requests, then the ring is the transpose scratch
    if (grp == 0) PP_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    {
      const int mrow = m0 + wm * 64, ncol = n0 + wn * 32;
      pp_epilogue<EPI, 2, 2>(g, acc, lane, 0, w_unscale, bias_lds, lds + wave * 4096, [&](int mi) { return mrow + mi * 32; },
                             [&](int ni) { return ncol + ni * 64; }, false, p.n_valid);
    }
    PP_BARRIER();  // scratch reads done before the next tile's prologue lands in the ring
  }
}

// ================================================================================================================
// Round 4: 3 x 3 / stride 1 / pad 1 convolutions with the input tile held in LDS ("halo" kernel).
//
// conv_pp128_kernel fetches a K-tile's A operand -- one 128-byte line per output pixel and (tap, channel group) -- by
// LDS-DMA for EVERY tap: nine fetches of (almost) the same input lines, 48 KiB of DMA per K-tile and workgroup.  Measured
// (profiles/r04_overlap_qkv_ab.txt and the per-launch rows of bench.py): 1.65 us per K-tile whatever the MFMA work (halving
// the MFMAs of the 64-channel layers bought 7 %) = 29 GB/s per CU, the rate one CU's LDS-DMA request stream sustains
// (MI355X_MICROARCH.md "ldsdma-fill": 25 GB/s per CU) -- the 3 x 3 layers are bound by that stream, not by the matrix pipe.
// Here a workgroup owns 8 x 32 output pixels (256 GEMM rows) x 128 output channels and fetches, per 32-channel group, the
// 10 x 34-pixel input halo ONCE (340 lines, 42.5 KiB); the nine taps read their A fragments from that halo at shifted
// rows.  DMA per K-tile: the weights' 16 KiB (8 KiB when only 64 output columns are stored) + 1/9 of a halo = 21 (13) KiB
// instead of 48.
//   LDS: halo[2] (double-buffered over channel groups, 48 KiB each: 384 rows of 128 B, XOR-swizzled by the PHYSICAL halo
//        row exactly like the GEMM's A rows, so a fragment read of 32 consecutive pixels of an output row is conflict-free
//        at any tap offset) | weight ring 3 x 16 KiB | bias; the epilogue's transpose scratch aliases halo[0].
//   K order: channel group outer, tap inner (the packed weights stay [n][ky][kx][c]: only the K-tile index is remapped).
//   Schedule (all 8 waves in the same phase, ONE barrier per K-tile, every wave issues the same number of pieces so the
//   counted vmcnt waits are uniform): iteration kt = (cg, tap):
//       wait  vmcnt(NB + (1 <= tap <= 6 ? 1 : 0)) -> my pieces of B(kt) have landed (and everything older: my halo pieces)
//       barrier                                     -> everyone's have; everyone is done reading B(kt-1) and, at tap 0, halo(cg-1)
//       read B(kt) fragments, then the A fragments of K-tile kt+1 (one K-tile ahead: the halo is resident); MFMAs of kt
//       issue [taps 0..5: ONE piece per wave of halo(cg+1) -> halo[(cg+1)&1]], B(kt+2) -> ring slot (kt+2) % 3, NB pieces per wave
//   (requests beyond the last K-tile / channel group are duplicates of the last one into free slots, drained at tile end).
//   First version (all DMA issued in front of the fragment reads, halo in one burst at tap 0, A fragments read in the K-tile that
//   uses them): 208 us on the 64-channel layer against 372 for conv_pp128_kernel; see profiles/r04_conv_halo_ab.txt.
// Same products, same f32 accumulation per output as conv_pp128_kernel up to the ORDER of the K-tiles (channel group outer
// instead of tap outer): results agree to f32 rounding, not bit for bit; per-frame determinism is unchanged (a tile never
// straddles frames).  Requires Hout % 8 == 0 and Wout % 32 == 0 (every layer of the 384 x 512 model resolution); other
// shapes and the stride-2 / 1 x 1 convolutions stay on conv_pp128_kernel.
constexpr int CH_PIECES = 43;                   // 340 halo rows = 42.5 pieces of 8 rows: waves 0..2 fetch 6 pieces, waves 3..7 five
constexpr int CH_HALO = CH_PIECES * 1024;       // one halo buffer (44 032 B)
constexpr int CH_B0 = 2 * CH_HALO;              // weight ring: 64 KiB = 8 units of 8 KiB (64 weight rows of one K-tile)
constexpr int CH_BRING = 65536;
constexpr int CH_BIAS = CH_B0 + CH_BRING;       // 153 600
static_assert(CH_BIAS + 4096 <= PP_LDS_ALL && (CH_B0 % 1024) == 0, "LDS budget");

// halo pieces a wave issued in the D - 1 iterations before K-tile (cg, tap): they are YOUNGER than its weight pieces B(kt)
// (issued D iterations ago), so the counted wait for B(kt) must let them stay in flight.  One piece per tap at taps < nh;
// `first`: channel group 0, where iterations before tap 0 do not exist.
constexpr int ch_halo_young(const int tap, const int D, const int nh, const bool first) {
  int c = 0;
  for (int i = 1; i < D; ++i) {
    int t = tap - i;
    if (first) {
      if (t < 0) continue;
    } else {
      t = ((t % 9) + 9) % 9;
    }
    if (t < nh) ++c;
  }
  return c;
}

template <bool DBG>
__device__ __forceinline__ void ch_wait_vm(const int n) {  // wave-uniform n in 0 .. 15 -> s_waitcnt vmcnt(n)
  switch (n) {
#define CH_CASE(N) case N: PP_WAIT_VM(N); break;
    CH_CASE(0) CH_CASE(1) CH_CASE(2) CH_CASE(3) CH_CASE(4) CH_CASE(5) CH_CASE(6) CH_CASE(7)
    CH_CASE(8) CH_CASE(9) CH_CASE(10) CH_CASE(11) CH_CASE(12) CH_CASE(13) CH_CASE(14) CH_CASE(15)
#undef CH_CASE
    default: PP_WAIT_VM(0); break;
  }
}

template <int EPI, bool N64>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(CtkConvP p, int tiles_total) {
  constexpr bool DBG = false;
  constexpr int NB = N64 ? 1 : 2;            // weight pieces per wave and K-tile (8 or 16 pieces of 8 rows)
  constexpr int BSLOT = N64 ? 8192 : 16384;  // bytes of one K-tile's weights
  constexpr int RING = CH_BRING / BSLOT;     // 8 or 4 K-tiles
  constexpr int D = RING - 1;                // weights are requested D K-tiles ahead
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS_ALL];
  const CtkGemmP& g = p.g;
  const int dbg = 0;
  unsigned jctr = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const unsigned l3 = lane >> 3, l4 = lane >> 4, l7 = lane & 7;
  const int cl = p.CL, KT = 9 * cl;
  const bool six = wave < 3;                               // my halo piece count: 6 (waves 0..2) or 5
  const int piece0 = six ? 6 * wave : 18 + 5 * (wave - 3);  // my first halo piece
  const float w_unscale = reinterpret_cast<const float*>(g.Wp)[1];
  const float* bias_lds = reinterpret_cast<const float*>(lds + CH_BIAS);
  if ((EPI & 32) != 0) {
    for (int i = tid; i < g.N / 4; i += 512) reinterpret_cast<f32x4*>(lds + CH_BIAS)[i] = reinterpret_cast<const f32x4*>(g.bias)[i];
    __syncthreads();
  }
  const unsigned char* in = static_cast<const unsigned char*>(g.A);
  const unsigned char* wsh = reinterpret_cast<const unsigned char*>(g.Wp) + PP_HDR_BYTES;
  const unsigned ldw_b = (unsigned)KT * 128;
  const int ty_tiles = p.Hout / 8, tx_tiles = p.Wout / 32;
  auto cbyte = [&](const int piece) { return ((l7 ^ ((4 * piece + l4) & 7)) << 4); };

  const int fswb = (r32 >> 1) & 7;
  unsigned b_rd0[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) b_rd0[j][pl] = (unsigned)((wn * 32 + r32) * 128 + (((pl * 4 + j * 2 + half) ^ fswb) << 4));

  f32x16 acc[2][2];
  f16x8 fb[2][2];

  for (int q = 0;; ++q) {  // my tiles
    const int G = gridDim.x, first = q * G;
    if (first >= tiles_total) break;
    const int n_r = min(G, tiles_total - first);
    if ((int)blockIdx.x >= n_r) break;
    unsigned tile = first + ctk_xcd_remap(blockIdx.x, n_r);
    const int nb = tile % g.nblocks;
    tile /= g.nblocks;
    const int tx = tile % tx_tiles;
    tile /= tx_tiles;
    const int ty = tile % ty_tiles;
    const int f = tile / ty_tiles;
    const int n0 = nb * 128, y0 = ty * 8, x0 = tx * 32;

    // ---- per-lane halo sources of my pieces (halo row r = 8 piece + l3 = hy*34 + hx)
    long hsrc[6];  // byte offset of the pixel's first line in `in`, or -1: zero line (padding ring / rows >= 340)
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int r = 8 * (piece0 + e) + (int)l3;
      const int hy = r / 34, hx = r - hy * 34;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = r < 340 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      hsrc[e] = ok ? ((long)(f * p.Hin + iy) * p.Win + ix) * cl * 128 : -1;
    }
    auto issue_halo_piece = [&](int cg, const int buf, const int e) {  // my e-th piece (e: compile-time; e = 5 only for waves 0..2)
      cg = min(cg, cl - 1);
      const int piece = piece0 + e;
      const unsigned char* src = (hsrc[e] >= 0 ? in + hsrc[e] + (long)cg * 128 : p.zeros) + cbyte(piece);
      __builtin_amdgcn_global_load_lds((pp_gptr)src, (pp_lptr)(lds + buf * CH_HALO + piece * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int kt, const int slot) {  // K-tile kt = cg * 9 + tap  ->  packed K-tile tap * cl + cg; NB pieces per wave
      kt = min(kt, KT - 1);
      const int cg = kt / 9, tap = kt - cg * 9;
      const long koff = (long)(tap * cl + cg) * 128;
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        const int piece = NB * wave + e;  // W rows n0 + 8 piece + l3
        const unsigned char* src = wsh + (long)(n0 + 8 * piece + (int)l3) * ldw_b + koff + cbyte(piece);
        __builtin_amdgcn_global_load_lds((pp_gptr)src, (pp_lptr)(lds + CH_B0 + slot * BSLOT + piece * 1024), 16, 0, 0);
      }
    };

    // ---- prologue: the first channel group's halo, the weights of K-tiles 0 .. D-1
#pragma unroll
    for (int e = 0; e < 5; ++e) issue_halo_piece(0, 0, e);
    if (six) issue_halo_piece(0, 0, 5);
#pragma unroll
    for (int i = 0; i < D; ++i) issue_b(i, i);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // A fragments of K-tile (cg, tap) -> fa2[tap & 1]: the halo is resident, so they are read one K-tile AHEAD, behind the weight
    // fragments of the current K-tile and in front of its MFMAs (the LDS serves a wave in order: the MFMAs wait for the weight
    // fragments only).  Nine taps per channel group is odd, so the fragments prefetched at tap 8 land in buffer 1 and move to
    // buffer 0 once per channel group.
    f16x8 fa2[2][2][2][2];  // [buffer][mi][j][plane]
    auto read_a = [&](const int buf, const unsigned hbase, const int ky, const int kx) {
      unsigned xl = (unsigned)r32;
      asm volatile("" : "+v"(xl));  // opaque per call: otherwise hipcc hoists the swizzled addresses of all nine taps out of the
                                    // channel-group loop (72 VGPRs of loop invariants -> 256 registers and scratch spills)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const unsigned R = (unsigned)((wm * 2 + mi + ky) * 34 + kx) + xl;
        const unsigned base = hbase + R * 128, sw = (R >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa2[buf][mi][j][pl] = *reinterpret_cast<const f16x8*>(lds + base + ((((unsigned)(pl * 4 + j * 2 + half)) ^ sw) << 4));
      }
    };

    int slot = 0, kt = 0;
    for (int cg = 0; cg < cl; ++cg) {
      const unsigned hb = (cg & 1) * CH_HALO;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap, ++kt) {
        // my pieces of B(kt) have landed when at most the requests YOUNGER than them are in flight: the weights of K-tiles
        // kt+1 .. kt+D-1 and the halo pieces of the last D-1 iterations
        {
          constexpr int base = (D - 1) * NB;
          const int young = cg == 0 ? (six ? ch_halo_young(tap, D, 6, true) : ch_halo_young(tap, D, 5, true))
                                    : (six ? ch_halo_young(tap, D, 6, false) : ch_halo_young(tap, D, 5, false));
          int allow = base + young;
          // tap 8 also reads the NEXT channel group's halo (the A prefetch below): all of it must have landed, i.e. only the
          // weight pieces issued at or after the iteration of my last halo piece (tap 5 / tap 4) may still be in flight
          if (tap == 8) allow = min(allow, NB * (six ? 3 : 4));
          ch_wait_vm<DBG>(allow);
        }
        PP_BARRIER();
        const unsigned so = CH_B0 + slot * BSLOT;
        if (kt == 0) read_a(0, hb, 0, 0);  // first K-tile of the tile: nothing was prefetched
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(lds + so + b_rd0[j][pl]);
        // next K-tile's A fragments: same channel group, or (tap 8) the next group's halo -- complete: the stricter wait of tap 8
        // covered every wave's own halo pieces and the barrier made that global
        if (tap < 8) read_a((tap + 1) & 1, hb, (tap + 1) / 3, (tap + 1) % 3);
        else if (cg + 1 < cl) read_a(1, hb ^ CH_HALO, 0, 0);
#pragma unroll
        for (int n = 0; n < (N64 ? 1 : 2); ++n) {
          if (n == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const f16x8*>(lds + so + b_rd0[j][pl] + 8192);
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
                acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][term == 0 ? 1 : 0], fa2[tap & 1][mi][j][term == 1 ? 1 : 0], acc[mi][n], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
        // DMA requests behind the MFMAs: one piece of the next channel group's halo per tap (taps 0..4, and 5 for waves 0..2), then
        // the weights D K-tiles ahead into the slot consumed in the previous iteration
        if (tap < 5 || (tap == 5 && six)) issue_halo_piece(cg + 1, (cg + 1) & 1, tap < 5 ? tap : 5);
        issue_b(kt + D, slot == 0 ? RING - 1 : slot - 1);
        slot = slot == RING - 1 ? 0 : slot + 1;
      }
      if (cg + 1 < cl) {  // tap 8 prefetched into buffer 1; tap 0 of the next group reads buffer 0
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa2[0][mi][j][pl] = fa2[1][mi][j][pl];
      }
    }
    // ---- tile end: drain the duplicate tail requests; halo[0] becomes the transpose scratch
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    {
      const int prow = (f * p.Hout + y0 + wm * 2) * p.Wout + x0;  // linear output pixel of (f, y0 + wm*2 + mi, x0): 32 consecutive pixels per mi
      const int ncol = n0 + wn * 32;
      pp_epilogue<EPI, 2, 2>(g, acc, lane, 0, w_unscale, bias_lds, lds + wave * 4096, [&](int mi) { return prow + mi * p.Wout; },
                             [&](int ni) { return ncol + ni * 64; }, false, p.n_valid);
    }
    PP_BARRIER();  // scratch reads done before the next tile's halo lands
  }
}

}  // namespace

// in_sh: SH activations NHWC [F][Hin][Win][Cin/32] lines; wp: ctk_pack_weight of the [Npad][KH*KW*Cin] matrix ([n][ky][kx][c] order,
// rows >= n_out zero); bias: Npad floats; out: f32 [F*Hout*Wout][n_out]; zeros: >= 128 zero bytes on the device.
extern "C" int ctk_conv2d_sh(const void* in_sh, int32_t F, int32_t Hin, int32_t Win, int32_t Cin, const void* wp, const float* bias,
                             int32_t n_out, int32_t n_pad, int32_t KH, int32_t KW, int32_t stride, int32_t pad, float* out,
                             const void* zeros, void* stream) {
  if (!in_sh || !wp || !bias || !out || !zeros) return CTK_E_NULL;
  if (F <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || (Cin % 32) || n_out <= 0 || n_pad < n_out || (n_pad % 128) || (n_out % 32) ||
      n_pad > 1024 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || Hin > 16000 || Win > 16000)
    return CTK_E_SHAPE;
  if (!ctk_aligned16(in_sh) || !ctk_aligned16(wp) || !ctk_aligned16(bias) || !ctk_aligned16(out) || !ctk_aligned16(zeros)) return CTK_E_ALIGN;
  const int Hout = (Hin + 2 * pad - KH) / stride + 1, Wout = (Win + 2 * pad - KW) / stride + 1;
  if (Hout <= 0 || Wout <= 0) return CTK_E_SHAPE;
  const long M = (long)F * Hout * Wout;
  if (M > 0x7fffffffL || (long)F * Hin * Win > 0x7fffffffL) return CTK_E_SHAPE;
  CtkConvP p;
  CtkGemmP& g = p.g;
  g = CtkGemmP{};
  g.A = in_sh; g.lda = 0; g.M = (int)M;
  g.N = n_pad; g.K = KH * KW * Cin;
  g.Wp = static_cast<const unsigned short*>(wp);
  g.C = out; g.ldc = n_out;
  g.bias = bias;
  g.act = CTK_ACT_NONE;
  g.batch = 1;
  g.mblocks = (int)((M + 255) / 256); g.nblocks = n_pad / 128;
  p.F = F; p.Hin = Hin; p.Win = Win; p.CL = Cin / 32;
  p.Hout = Hout; p.Wout = Wout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.n_valid = n_out;
  p.zeros = static_cast<const unsigned char*>(zeros);
  const long tiles = (long)g.mblocks * g.nblocks;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus <= 0) cus = 256;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char pname[48];
  snprintf(pname, sizeof(pname), "conv_pp128_%dx%d_s%d_c%d_n%d", KH, KW, stride, Cin, n_out);
  const double flops = 2.0 * M * (double)n_out * g.K;
  // 3 x 3 / stride 1 / pad 1 on tile-aligned maps: the halo kernel (input tile in LDS, nine taps read it there).  CTK_CONV_HALO=0
  // (dev builds only) keeps every convolution on conv_pp128_kernel.
  const bool halo_on = CTK_DEV_KNOB("CTK_CONV_HALO", 1) != 0;
  if (halo_on && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Hout % 8 == 0 && Wout % 32 == 0 && Hout == Hin && Wout == Win) {
    const long htiles = (long)F * (Hout / 8) * (Wout / 32) * g.nblocks;
    char hname[48];
    snprintf(hname, sizeof(hname), "conv_halo_3x3_c%d_n%d", Cin, n_out);
    CtkProfScope hps(hname, flops, 4.0 * ((double)F * Hin * Win * Cin + (double)M * n_out), s);
    const dim3 hgrid((unsigned)(htiles < cus ? htiles : cus));
    if (n_out <= 64) hipLaunchKernelGGL((conv3x3_halo_kernel<32, true>), hgrid, dim3(512), 0, s, p, (int)htiles);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<32, false>), hgrid, dim3(512), 0, s, p, (int)htiles);
    CTK_HIP_CHECK_LAUNCH();
    return CTK_OK;
  }
  CtkProfScope ps(pname, flops, 4.0 * ((double)F * Hin * Win * Cin + (double)M * n_out), s);
  const bool force_ph1 = CTK_DEV_KNOB("CTK_CONV_PH1", 0) == 1;  // dev builds: 1 = round-3 behaviour
  hipLaunchKernelGGL((conv_pp128_kernel<32>), dim3((unsigned)(tiles < cus ? tiles : cus)), dim3(512), 0, s, p, (int)tiles, force_ph1);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
