// Split-half GEMM, round-3 kernel: ONE persistent 8-wave workgroup per CU, the two waves of every SIMD in
// anti-phase ("ping-pong").
//
// Why (profiles/r02_sq_counters.txt read with MI355X_MICROARCH.md's definitions): in gemm_f16x3.hip's kernels a wave
// is issue-stalled 61 % of its life while the matrix pipe is idle 55 % of the time -- the co-resident waves of a
// SIMD run IN PHASE (both in their MFMA burst, then both issuing 8 LDS-DMA pieces at 100-185 cycles each, both
// reading fragments, both parked at the vmcnt(0) barrier), and a 128x128 tile needs 26 TB/s of L2->LDS traffic at
// the MFMA ceiling.  This kernel changes the structure, not the arithmetic:
//   * 256-row tiles, 8 waves = 2 per SIMD in one workgroup; waves 0-3 (group 0) and 4-7 (group 1) run the same
//     phase sequence one s_barrier apart: while one group issues its 12 MFMAs of a phase (384 matrix-pipe cycles)
//     the other reads the next fragments from LDS and issues its 2-3 LDS-DMA pieces (cdna_hip_programming.md,
//     "The 256^2 8-phase template");
//   * the K-tile (32 f32 columns = one 128-byte SH line per row) is cut into blocks that are each read in exactly
//     one phase, so the LDS ring (2 K-tiles) is recycled block by block: a block is requested 4-6 phases before it
//     is read, waited for with a COUNTED s_waitcnt vmcnt (never 0) one phase before, behind raw s_barriers;
//   * persistent: a workgroup walks over its tiles and the DMA stream simply continues into the next tile, so
//     only the first tile of a workgroup pays a cold prologue; the two groups run their epilogues one MFMA phase
//     apart without dropping the stagger;
//   * two tile shapes with the same per-wave structure: 256 x 256 (waves 2 x 4, wave tile 128 x 64) for
//     N % 256 == 0 and 256 x 192 (waves 4 x 2, wave tile 64 x 96) for N % 192 == 0 (N = 384: A is fetched
//     twice instead of three times).
// Same numerics as gemm_f16x3.hip (3 x v_mfma_f32_32x32x16_f16 per product, small terms first within a k-step).
#include "pp_common.h"
#include "ctk_options.h"
#include <type_traits>

namespace {

#ifdef CTK_DEV
// ---- wave timeline (DBG kernels only, ctk_gemm_pp_mode bit 6): the first PP_TRACE_WGS workgroups stamp s_memtime at the start of
// every MFMA phase (right behind the lgkmcnt(0) wait that is there anyway), before and after every epilogue, into the 8 KiB of LDS
// no kernel uses, and dump them at exit; ctk_debug_pp_trace() copies them out (tools/gemm_lab trace).  The stamp itself waits
// for its SMEM result (~100 cycles per phase): periods are inflated uniformly, their RATIOS are what is read.
constexpr int PP_TRACE_WGS = 4, PP_TRACE_STAMPS = 128, PP_TRACE_OFF = 155648;
__device__ unsigned long long g_pp_trace[PP_TRACE_WGS * 8 * PP_TRACE_STAMPS];
// (s_memtime, s_memrealtime) pair: the last four slots of a wave's trace hold it at kernel start and end -- s_memrealtime is the
// constant 100 MHz counter, so the pair gives the rate s_memtime ticked at over the kernel's life
__device__ __forceinline__ void pp_trace_clocks(unsigned char* dst, const int lane) {
  unsigned long long a, b;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : : "memory");
  if (lane == 0) {
    reinterpret_cast<unsigned long long*>(dst)[0] = a;
    reinterpret_cast<unsigned long long*>(dst)[1] = b;
  }
}
// Every launch of a DEV build (make dev): wave 0 of workgroup 0 leaves (s_memtime, s_memrealtime) at its start and end in g_pp_clock --
// two scalar loads and two 16-byte stores per launch.  s_memrealtime ticks at 100 MHz, so the pair gives the shader clock the
// launch actually ran at (ctk_debug_pp_clock; tools/gemm_lab clock): the GEMMs are power-capped, and by how much is a number.
__device__ unsigned long long g_pp_clock[4];
__device__ __forceinline__ void pp_clock_probe(const int which, const int tid) {
  if (blockIdx.x == 0 && tid < 64) {
    unsigned long long a, b;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : : "memory");
    if (tid == 0) {
      g_pp_clock[2 * which] = a;
      g_pp_clock[2 * which + 1] = b;
    }
  }
}
#define PP_STAMP()                                                                                                      \
  do {                                                                                                                  \
    if (DBG && tr_on) {                                                                                                 \
      unsigned long long t_;                                                                                            \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");                                     \
      if (lane == 0 && tr_n < PP_TRACE_STAMPS - 4) *reinterpret_cast<unsigned long long*>(lds + PP_TRACE_OFF + wave * 1024 + tr_n * 8) = t_; \
      ++tr_n;                                                                                                           \
    }                                                                                                                   \
  } while (0)
#define PP_TRACE_INIT()                                                                                                 \
  const bool tr_on = DBG && (dbg & 64) != 0 && blockIdx.x < PP_TRACE_WGS;                                               \
  int tr_n = 0;                                                                                                         \
  if (DBG && tr_on) {                                                                                                   \
    for (int i = lane; i < PP_TRACE_STAMPS; i += 64) *reinterpret_cast<unsigned long long*>(lds + PP_TRACE_OFF + wave * 1024 + i * 8) = 0ull; \
    pp_trace_clocks(lds + PP_TRACE_OFF + wave * 1024 + (PP_TRACE_STAMPS - 4) * 8, lane);                                  \
  }
#define PP_TRACE_DUMP()                                                                                                 \
  do {                                                                                                                  \
    if (DBG && tr_on) {                                                                                                 \
      pp_trace_clocks(lds + PP_TRACE_OFF + wave * 1024 + (PP_TRACE_STAMPS - 2) * 8, lane);                              \
      for (int i = lane; i < PP_TRACE_STAMPS; i += 64)                                                                  \
        g_pp_trace[((long)blockIdx.x * 8 + wave) * PP_TRACE_STAMPS + i] = *reinterpret_cast<unsigned long long*>(lds + PP_TRACE_OFF + wave * 1024 + i * 8); \
    }                                                                                                                   \
  } while (0)
#else  // release build: no instruments, no device-side globals written by production launches
#define PP_STAMP() do {} while (0)
#define PP_TRACE_INIT() do {} while (0)
#define PP_TRACE_DUMP() do {} while (0)
__device__ __forceinline__ void pp_clock_probe(const int, const int) {}
#endif

// ---- tile walk -----------------------------------------------------------------------------------
// The tiles are dealt to the G persistent workgroups in rounds: in round r the G (or fewer) tiles [r*G, r*G + n_r) are dealt so
// that XCD x (workgroup b sits on XCD b % 8) gets a contiguous run of logical tiles (neighbours share A rows / W in its L2).
// 800 tiles on 256 CUs (every N = 384 Linear of the model at C3) would be FOUR rounds with the last one 1/8 full; the launcher's
// tail split (ctk_launch_gemm_pp) hands such a last round to the 64 x 64-tile kernel instead.  A stream-K walk (the tiles' K ranges
// dealt as one stream, partial tiles exchanged through cache-bypassing stores and per-wave flags) lived here in rounds 3-5, off by
// default: it won 5 % on mlp.fc2 only (profiles/r03_gemm_lab_streamk.txt) and brought a spin-wait into the hottest kernels;
// round 6 removed it (git show 360f4f4:co-tracker_amd/csrc/gemm_pp.hip has the code).
struct PPTile {
  const unsigned char* a;  // A rows of the tile, K-tile 0 (bytes)
  const unsigned char* w;  // W rows of the tile, K-tile 0
  int m0, n0, bz;
  unsigned lim;            // last valid row inside the tile (rows beyond M are clamped, never stored)
  int kb, ke;              // K-tiles [kb, ke) = [0, KT)
};

struct PPWalk {
  int tiles_total, KT;
};

template <int BM, int BN>
__device__ __forceinline__ void pp_tile_at(const CtkGemmP& g, unsigned tile, const int KT, PPTile& t) {
  const int nb = tile % g.nblocks;
  tile /= g.nblocks;
  const int mb = tile % g.mblocks;
  t.bz = tile / g.mblocks;
  t.m0 = mb * BM;
  t.n0 = nb * BN;
  t.a = reinterpret_cast<const unsigned char*>(g.A) + ((long)t.bz * g.a_bs + (long)t.m0 * g.lda) * 2;
  t.w = reinterpret_cast<const unsigned char*>(g.Wp) + PP_HDR_BYTES + (long)t.n0 * KT * 128;
  t.lim = (unsigned)min(BM - 1, g.M - 1 - t.m0);
}

__device__ __forceinline__ PPWalk pp_walk_init(const int tiles_total, const int KT) { return PPWalk{tiles_total, KT}; }

// q-th tile of this workgroup's walk
template <int BM, int BN>
__device__ __forceinline__ bool pp_tile(const CtkGemmP& g, const PPWalk& w, int q, PPTile& t) {
  const int G = gridDim.x, b = blockIdx.x;
  const int first = q * G;
  if (first >= w.tiles_total) return false;
  const int n_r = min(G, w.tiles_total - first);
  if (b >= n_r) return false;
  pp_tile_at<BM, BN>(g, first + ctk_xcd_remap(b, n_r), w.KT, t);
  t.kb = 0;
  t.ke = w.KT;
  return true;
}

// DMA cursor: one K-tile of the workgroup's stream (tile q, K-tile kt); saturates at the end of the stream (the ring
// then receives harmless duplicate blocks, which keeps every vmcnt count of the steady state valid in the tail).
struct PPCursor {
  const unsigned char* a;
  const unsigned char* w;
  unsigned lim;
  int q, kt, ke;
};

__device__ __forceinline__ PPCursor pp_cursor_at(const PPTile& t) {
  return PPCursor{t.a + t.kb * 128, t.w + t.kb * 128, t.lim, 0, t.kb, t.ke};
}

template <int BM, int BN>
__device__ __forceinline__ void pp_cursor_next(const CtkGemmP& g, const PPWalk& w, PPCursor& c) {
  if (c.kt + 1 < c.ke) {
    c.kt += 1;
    c.a += 128;
    c.w += 128;
  } else {
    PPTile t;
    if (pp_tile<BM, BN>(g, w, c.q + 1, t)) {  // (only the first tile of a walk starts past K-tile 0)
      c.q += 1;
      c.kt = 0;
      c.ke = t.ke;
      c.a = t.a;
      c.w = t.w;
      c.lim = t.lim;
    }
  }
}

// ================================================================================================
// 256 x 256 tile.  waves 2 (M) x 4 (N); wave (wm, wn) owns rows {a*128 + wm*64 + mi*32 + [0,32)} and columns
// {b*128 + wn*32 + [0,32)} for a, mi, b in {0,1}: each 128-row half of the A tile and each 128-row half of the W tile is
// one BLOCK (16 KiB, 16 LDS-DMA pieces, 2 per wave), and phase (a, b) of a K-tile -- 2 x 1 accumulators x 2 k-steps x 3
// terms = 12 MFMAs -- reads block A_a and block B_b only.  Phase order (0,0) (0,1) (1,1) (1,0): the A fragments are read
// in phases 0 and 2, B_0 in phase 0 (kept for phase 3), B_1 in phase 1.
// Stream of blocks: i = 4 J + {0: A_0, 1: B_0, 2: B_1, 3: A_1} (J = K-tile of the workgroup's stream); block i is read in
// phase <= i, issued in the load segment of phase i - 6 (slot = that of block i - 8, last read >= 2 phases earlier) and
// every wave has waited for its pieces of blocks <= g + 2 at the end of the load segment of phase g (vmcnt(8): the 4
// younger blocks stay in flight), one s_barrier before any wave reads them.
// LDS: A blocks at (J&1)*32K + a*16K, B blocks at 64K + (J&1)*32K + b*16K.
// (Round 5 measured the phase's LDS-DMA pieces issued INSIDE its MFMA burst instead of in the load segment: +-0, bit-identical --
// profiles/r05_dma_in_mfma_ab.txt; the variant left the tree in round 6.)
template <int EPI, bool DBG, int TAG = 0>  // TAG: no code difference, only a distinct kernel NAME per Linear for rocprofv3 (tools/pmc_traffic.py)
__global__ __launch_bounds__(512) void gemm_pp256_kernel(CtkGemmP g, int tiles_total, int dbg_arg) {
  const int dbg = DBG ? dbg_arg : 0;  // the experiment knobs exist only in the DBG instantiations
  constexpr int BM = 256, BN = 256;
  constexpr int RING = 131072;
  // ONE LDS object (a second one makes hipcc drain vmcnt before ds_reads), and ALL 160 KiB of the CU: see PP_LDS_ALL
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS_ALL];
  static_assert(RING + PP_BIAS_BYTES + 4 * 4096 <= PP_LDS_ALL, "LDS budget");

  const int KT = g.K / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // ping-pong group (waves w and w + 4 share a SIMD)
  unsigned jctr = blockIdx.x * 977u + 1u;
  const int wm = wave >> 2, wn = wave & 3;
  const int r32 = lane & 31, half = lane >> 5;

  const PPWalk walk = pp_walk_init(tiles_total, KT);
  PPTile tile;
  if (!pp_tile<BM, BN>(g, walk, 0, tile)) return;
  const float* bias_lds = reinterpret_cast<const float*>(lds + RING);
  const float w_scale = reinterpret_cast<const float*>(g.Wp)[0], w_unscale = reinterpret_cast<const float*>(g.Wp)[1];
  pp_stage_bias<EPI>(g, lds + RING, tid);
  PP_TRACE_INIT();
  pp_clock_probe(0, tid);
  if (dbg >> 8) {  // experiment: de-phase the workgroups (class = (blockIdx / 8) % 4 sleeps class * (dbg >> 8) * 8128 cycles)
    const int n = ((blockIdx.x >> 3) & 3) * (dbg >> 8);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- DMA addressing.  Piece p = 2*wave + e of a block covers block rows 8p + (lane >> 3); lane position (lane & 7) of a row
  // receives source chunk (lane & 7) ^ f(row), f(r) = (r >> 1) & 7 (undone by the fragment reads).
  const unsigned prow = 16 * wave + (lane >> 3);                       // e = 0 row inside the block
  const unsigned cb0 = (((lane & 7) ^ ((lane >> 4) & 7)) << 4);       // f(row) for e = 0: ((16w + (lane>>3)) >> 1) & 7 = lane >> 4
  const unsigned cb1 = cb0 ^ 64;                                       // e = 1: rows + 8 -> f ^ 4
  const unsigned lda_b = (unsigned)g.lda * 2;                          // bytes per A row
  const unsigned ldw_b = (unsigned)KT * 128;                           // bytes per packed W row
  const unsigned wv0 = prow * ldw_b + cb0, wv1 = (prow + 8) * ldw_b + cb1;

  auto dma_a = [&](const PPCursor& c, const int a, const int par) {
    unsigned char* dst = lds + par * 32768 + a * 16384 + wave * 2048;
    const unsigned r0 = min(prow + a * 128, c.lim), r1 = min(prow + a * 128 + 8, c.lim);
    pp_dma16(c.a, r0 * lda_b + cb0, dst);
    pp_dma16(c.a, r1 * lda_b + cb1, dst + 1024);
  };
  auto dma_b = [&](const PPCursor& c, const int b, const int par) {
    unsigned char* dst = lds + 65536 + par * 32768 + b * 16384 + wave * 2048;
    const unsigned char* src = c.w + (long)b * 128 * ldw_b;
    pp_dma16(src, wv0, dst);
    pp_dma16(src, wv1, dst + 1024);
  };

  // ---- fragment addressing: row r of a block, data chunk c = plane*4 + j*2 + half at r*128 + ((c ^ f(r)) << 4)
  const int fsw = (r32 >> 1) & 7;
  unsigned a_rd[2][2], b_rd[2][2];  // [j][plane], including the K-tile parity bit (32 KiB)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned co = (unsigned)(((p * 4 + j * 2 + half) ^ fsw) << 4);
      a_rd[j][p] = (wm * 64 + r32) * 128 + co;
      b_rd[j][p] = 65536 + (wn * 32 + r32) * 128 + co;
    }

  f16x8 fa[2][2][2];   // [mi][j][plane]
  f16x8 fb[2][2][2];   // [b][j][plane]
  auto read_a = [&](const int a) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) fa[mi][j][p] = *reinterpret_cast<const f16x8*>(lds + a_rd[j][p] + a * 16384 + mi * 4096);
  };
  auto read_b = [&](const int b) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) fb[b][j][p] = *reinterpret_cast<const f16x8*>(lds + b_rd[j][p] + b * 16384);
  };

  f32x16 acc[4][2];  // [a*2 + mi][b]
  int par = 0;  // parity of the K-tile being computed
  auto row_of = [&](int i) { return tile.m0 + wm * 64 + (i >> 1) * 128 + (i & 1) * 32; };
  auto col_of = [&](int b) { return tile.n0 + wn * 32 + b * 128; };
  // wave-private 4 KiB for the store / residual transposes.  Group 0: dedicated space behind the bias.  It must NOT borrow a
  // ring slot: its waves fall straight from the epilogue into phase 0 of the next K-tile, whose LDS-DMA (B_1 of K-tile J+2) a
  // fast wave issues while a slow one is still transposing (found by the timing-jitter run of tools/gemm_lab.cpp).  Group 1
  // borrows the idle slot A_1 of the K-tile just finished (parity par ^ 1 once par has advanced): the next block landing
  // there (A_1 of K-tile J+2) is issued in phase 1, which no wave enters before a barrier that every group-1 wave reaches
  // only after its epilogue and accumulator set-up.
  auto scratch = [&]() { return grp == 0 ? lds + RING + PP_BIAS_BYTES + wave * 4096 : lds + (par ^ 1) * 32768 + 16384 + (wave & 3) * 4096; };
  PPResid<4, 2> resid;
  auto resid_issue = [&]() { pp_resid_issue<EPI, 4, 2>(g, resid, lane, tile.bz, row_of, col_of); };
  auto init_acc = [&](const float sc) { pp_init_acc<EPI, 4, 2>(acc, resid, lane, sc, scratch()); };
  // operands swapped on purpose (D'[n][m]: lane = output row, register quad = 4 consecutive columns); small terms first
  auto mma = [&](const int a, const int b) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[a * 2 + mi][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[b][j][term == 0 ? 1 : 0], fa[mi][j][term == 1 ? 1 : 0],
                                                                      acc[a * 2 + mi][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  constexpr int WV = 8;  // in flight behind a phase's wait: 4 blocks

  // ---- stream set-up: blocks 0..5 = K-tile 0 (all four) + K-tile 1 (A_0, B_0)
  PPCursor c1, c2;  // K-tiles J+1 and J+2 of the stream
  {
    PPCursor c0 = pp_cursor_at(tile);
    dma_a(c0, 0, 0);
    dma_b(c0, 0, 0);
    dma_b(c0, 1, 0);
    dma_a(c0, 1, 0);
    c1 = c0;
    pp_cursor_next<BM, BN>(g, walk, c1);
    dma_a(c1, 0, 1);
    dma_b(c1, 0, 1);
    c2 = c1;
    pp_cursor_next<BM, BN>(g, walk, c2);
  }
  resid_issue();
  init_acc(w_scale);
  PP_WAIT_VM(8);   // blocks 0, 1 (K-tile 0: A_0, B_0) have landed
  PP_BARRIER();
  if (grp == 1) PP_BARRIER();  // stagger: group 1 runs one barrier behind group 0

  for (int q = 0;; ++q) {  // my tiles
    for (int kt = tile.kb; kt < tile.ke; ++kt) {
      // ---- phase 0 (a=0, b=0): read A_0, B_0; issue B_1 of K-tile J+1
      read_a(0);
      read_b(0);
      dma_b(c1, 1, par ^ 1);
      PP_WAIT_VM(WV);
      PP_BARRIER();
      PP_WAIT_LGKM0();
      PP_STAMP();
      mma(0, 0);
      PP_BARRIER();
      // ---- phase 1 (a=0, b=1): read B_1; issue A_1 of K-tile J+1
      read_b(1);
      dma_a(c1, 1, par ^ 1);
      PP_WAIT_VM(WV);
      PP_BARRIER();
      PP_WAIT_LGKM0();
      PP_STAMP();
      mma(0, 1);
      PP_BARRIER();
      // ---- phase 2 (a=1, b=1): read A_1; issue A_0 of K-tile J+2 (into the slot A_0 of this K-tile left in phase 0)
      read_a(1);
      dma_a(c2, 0, par);
      PP_WAIT_VM(WV);
      PP_BARRIER();
      PP_WAIT_LGKM0();
      PP_STAMP();
      mma(1, 1);
      PP_BARRIER();
      // ---- phase 3 (a=1, b=0): nothing to read (B_0 is still in registers); issue B_0 of K-tile J+2
      dma_b(c2, 0, par);
      PP_WAIT_VM(WV);
      const bool last = kt + 1 == tile.ke;
      PP_BARRIER();
      PP_STAMP();
      mma(1, 0);
      if (!last) PP_BARRIER();
      // advance the stream
      c1 = c2;
      pp_cursor_next<BM, BN>(g, walk, c2);
      par ^= 1;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a_rd[j][p] ^= 32768;
          b_rd[j][p] ^= 32768;
        }
    }
    // ---- epilogue: group 0 after the phase's closing barrier, group 1 before it -- both write their tile at the same
    // time (one MFMA phase apart) and the stagger survives into the next tile
    if (grp == 0) PP_BARRIER();
    PP_STAMP();  // epilogue begins
    // the next tile's residual is requested BEFORE this tile's epilogue (it lands behind the epilogue's arithmetic and stores);
    // the epilogue therefore works on a copy of the tile descriptor
    const PPTile done = tile;
    const bool more = pp_tile<BM, BN>(g, walk, q + 1, tile);
    resid_issue();  // unconditional (on the last tile it re-reads valid addresses; a conditional re-init doubles the live accumulators)
    pp_epilogue<EPI, 4, 2>(g, acc, lane, done.bz, w_unscale, bias_lds, scratch(),
                           [&](int i) { return done.m0 + wm * 64 + (i >> 1) * 128 + (i & 1) * 32; }, [&](int b) { return done.n0 + wn * 32 + b * 128; },
                           (dbg & 2) != 0);
    init_acc(w_scale);
    PP_STAMP();  // epilogue done
    if (grp == 1) PP_BARRIER();
    if (!more) break;
  }
  if (grp == 0) PP_BARRIER();  // balance group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // duplicate tail blocks may still be landing
  PP_TRACE_DUMP();
  pp_clock_probe(1, tid);
}

// ================================================================================================
// 256 x 192 tile (N = 384 as two column halves).  waves 4 (M) x 2 (N); wave (wm, wn) owns rows wm*64 + mi*32 + [0,32)
// (mi < 2) and columns ni*64 + wn*32 + [0,32) (ni < 3).  Phase n of a K-tile = accumulators (mi, ni = n), 12 MFMAs:
// phase 0 reads the wave's A fragments (kept for the K-tile) and B_0, phase 1 B_1, phase 2 B_2 (B_n = the 64 W rows of
// column block n, 8 KiB).  Per K-tile 56 pieces in order of need -- A (32), B_0 (8), B_1 (8), B_2 (8) -- issued as
// three blocks I0 = A pieces 0..23 (3 per wave), I1 = A pieces 24..31 + B_0 (2 per wave), I2 = B_1 + B_2 (2 per wave);
// block i = 3 J + k is issued in phase i - 4 into the K-tile slot (J & 1) (56 KiB each), where block i - 6 was last read
// >= 2 phases earlier; waits: end of phase 3J+2 -> I0, I1 of K-tile J+1 (vmcnt(5): I2(J+1) and I0(J+2) stay in flight),
// end of phase 3J -> I2 of K-tile J (vmcnt(5)), end of phase 3J+1 -> nothing new.
template <int EPI, bool DBG, int TAG = 0>
__global__ __launch_bounds__(512) void gemm_pp192_kernel(CtkGemmP g, int tiles_total, int dbg_arg) {
  const int dbg = DBG ? dbg_arg : 0;
  constexpr int BM = 256, BN = 192;
  constexpr int SLOT = 57344;  // 32 KiB A + 3 x 8 KiB B
  constexpr int RING = 2 * SLOT;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS_ALL];  // ring | bias | store-transpose scratch (| unused: see PP_LDS_ALL)
  static_assert(RING + PP_BIAS_BYTES + 8 * 4096 <= PP_LDS_ALL, "LDS budget");

  const int KT = g.K / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  unsigned jctr = blockIdx.x * 977u + 1u;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  const PPWalk walk = pp_walk_init(tiles_total, KT);
  PPTile tile;
  if (!pp_tile<BM, BN>(g, walk, 0, tile)) return;
  const float* bias_lds = reinterpret_cast<const float*>(lds + RING);
  const float w_scale = reinterpret_cast<const float*>(g.Wp)[0], w_unscale = reinterpret_cast<const float*>(g.Wp)[1];
  pp_stage_bias<EPI>(g, lds + RING, tid);
  PP_TRACE_INIT();
  pp_clock_probe(0, tid);
  // power experiments (DBG only; tools/gemm_lab clock <mode>, profiles/r05_gemm_power_experiments.txt): bit 2 = issue no MFMA
  // (everything else unchanged: what the operand feed alone costs), stagger field 255 = no fragment reads
  const bool no_mfma = DBG && (dbg & 4) != 0, no_frag = DBG && (dbg >> 8) == 255;
  if ((dbg >> 8) && !no_frag) {  // experiment: de-phase the workgroups (class = (blockIdx / 8) % 4 sleeps class * (dbg >> 8) * 8128 cycles)
    const int n = ((blockIdx.x >> 3) & 3) * (dbg >> 8);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- DMA addressing.  A piece p covers tile rows 8p + (lane>>3) -> LDS p*1024; f(row) = (row >> 1) & 7 = (4p + (lane >> 4)) & 7.
  const unsigned l3 = lane >> 3, l4 = lane >> 4, l7 = lane & 7;
  const unsigned lda_b = (unsigned)g.lda * 2;
  const unsigned ldw_b = (unsigned)KT * 128;
  auto cbyte = [&](const int piece) { return ((l7 ^ ((4 * piece + l4) & 7)) << 4); };
  auto dma_a_piece = [&](const PPCursor& c, const int piece, const int par) {  // piece: uniform
    const unsigned r = min((unsigned)(8 * piece) + l3, c.lim);
    pp_dma16(c.a, r * lda_b + cbyte(piece), lds + par * SLOT + piece * 1024);
  };
  // B_n piece p (0..7) covers W rows n*64 + 8p + (lane>>3) -> LDS 32768 + n*8192 + p*1024; swizzle by the row inside B_n
  auto dma_b_piece = [&](const PPCursor& c, const int n, const int piece, const int par) {
    const unsigned r = (unsigned)(n * 64 + 8 * piece) + l3;
    pp_dma16(c.w, r * ldw_b + cbyte(piece), lds + par * SLOT + 32768 + n * 8192 + piece * 1024);
  };
  auto issue_i0 = [&](const PPCursor& c, const int par) {
#pragma unroll
    for (int e = 0; e < 3; ++e) dma_a_piece(c, 3 * wave + e, par);
  };
  auto issue_i1 = [&](const PPCursor& c, const int par) {
    if (wave < 4) {
#pragma unroll
      for (int e = 0; e < 2; ++e) dma_a_piece(c, 24 + 2 * wave + e, par);
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) dma_b_piece(c, 0, 2 * (wave - 4) + e, par);
    }
  };
  auto issue_i2 = [&](const PPCursor& c, const int par) {
    const int n = wave < 4 ? 1 : 2, w4 = wave & 3;
#pragma unroll
    for (int e = 0; e < 2; ++e) dma_b_piece(c, n, 2 * w4 + e, par);
  };

  // ---- fragment addressing
  const int fsw = (r32 >> 1) & 7;
  unsigned a_rd[2][2], b_rd[2][2];
  auto set_rd = [&](const int par) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned co = (unsigned)(((p * 4 + j * 2 + half) ^ fsw) << 4);
        a_rd[j][p] = par * SLOT + (wm * 64 + r32) * 128 + co;
        b_rd[j][p] = par * SLOT + 32768 + (wn * 32 + r32) * 128 + co;
      }
  };
  f16x8 fa[2][2][2];  // [mi][j][plane]
  f16x8 fb[2][2];     // [j][plane] of the current column block
  bool kt_seen = false;  // (no_frag experiment: the first K-tile's fragments are read, later ones are not)
  auto read_a = [&]() {
    if (no_frag && kt_seen) return;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) fa[mi][j][p] = *reinterpret_cast<const f16x8*>(lds + a_rd[j][p] + mi * 4096);
  };
  auto read_b = [&](const int n) {
    if (no_frag && kt_seen) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) fb[j][p] = *reinterpret_cast<const f16x8*>(lds + b_rd[j][p] + n * 8192);
  };

  f32x16 acc[2][3];
  auto row_of = [&](int mi) { return tile.m0 + wm * 64 + mi * 32; };
  auto col_of = [&](int ni) { return tile.n0 + wn * 32 + ni * 64; };
  // ---- residual (EPI bit 2), round 5.  Until round 4 the NEXT tile's residual (196 KiB per workgroup) was requested in one burst
  // in front of the epilogue and transposed into the accumulators right behind it -- a fully exposed burst with the matrix pipe
  // idle (tools/gemm_lab: to_out 148 us against 111 us for the same Linear without residual).  Holding it in registers across
  // the main loop instead does not fit (96 + 48 + 96 registers: hipcc spills, and a spilled load is a vmcnt(0) in the hot loop).
  // Now the CURRENT tile's residual rides on the tile's first eight K-tiles, one 32 x 32 sub-tile k = (mi, ni) at a time:
  //   K-tile k,     phase 1: its four 1-KiB line pieces are requested (asm loads: hipcc must neither count nor wait for them);
  //   K-tile k + 1, phase 1: (landed: the counted wait of phase 0 retired them) scaled by s, written to the wave's LDS image;
  //   K-tile k + 1, phase 2 (ni = 2: K-tile k + 2, phase 0): read back in accumulator layout and ADDED to acc[mi][ni] behind the
  //                          phase's MFMAs, which work on another column block.
  // 16 + 16 registers for three phases, four loads per wave and K-tile beside 7 DMA pieces; the wait of the phase behind the
  // loads leaves 4 more operations in flight (vmcnt 9: the loads are younger than the two DMA pieces that may stay in flight
  // with them), every other wait is unchanged.  tools/check_pp_schedule.py replays this program.
  constexpr bool RP = (EPI & 4) != 0;
  f32x4 rp[4], rv[4];
  unsigned char* const scr = lds + RING + PP_BIAS_BYTES + wave * 4096;
  const int rrow = lane >> 3, rchunk = lane & 7, rsw = r32 & 7;
  auto res_load = [&](auto k_tag) {
    constexpr int k = decltype(k_tag)::value, mi = k / 3, ni = k % 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowc = min(row_of(mi) + rrow + 8 * i, g.M - 1);
      const float* ptr = g.resid + (long)tile.bz * g.c_bs + (long)rowc * g.ldr + rchunk * 4 + col_of(ni);
      f32x4 t;  // (an asm operand inside a generic lambda cannot name a captured variable)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(ptr) : "memory");
      rp[i] = t;
    }
  };
  auto res_write = [&]() {  // row r = rrow + 8 i, chunk c at position c ^ (r & 7) (as pp_init_acc)
    asm volatile("" : "+v"(rp[0]), "+v"(rp[1]), "+v"(rp[2]), "+v"(rp[3]));  // no consumer above this point
    unsigned char* wr = scr + rrow * 128 + ((rchunk ^ rrow) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(wr + i * 1024) = rp[i] * w_scale;
  };
  auto res_read = [&]() {
    const unsigned char* rd = scr + r32 * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const f32x4*>(rd + (((2 * q + half) ^ rsw) << 4));
  };
  auto res_add = [&](auto k_tag) {
    constexpr int k = decltype(k_tag)::value, mi = k / 3, ni = k % 3;
    PP_SCHED_FENCE();  // behind the MFMAs of the phase, not in front of them
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[mi][ni][q * 4 + e] += rv[q][e];
  };
  auto init_acc = [&]() {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
  };
  auto mma = [&](const int n) {
    if (no_mfma) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][term == 0 ? 1 : 0], fa[mi][j][term == 1 ? 1 : 0], acc[mi][n], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- stream set-up: blocks 0..3 = K-tile 0 (I0, I1, I2) + I0 of K-tile 1
  PPCursor c1, c2;
  {
    PPCursor c0 = pp_cursor_at(tile);
    issue_i0(c0, 0);
    issue_i1(c0, 0);
    issue_i2(c0, 0);
    c1 = c0;
    pp_cursor_next<BM, BN>(g, walk, c1);
    issue_i0(c1, 1);
    c2 = c1;
    pp_cursor_next<BM, BN>(g, walk, c2);
  }
  init_acc();
  set_rd(0);
  PP_WAIT_VM(5);  // I0, I1 of K-tile 0 landed (I2: 2 pieces and I0 of K-tile 1: 3 pieces may be in flight)
  PP_BARRIER();
  if (grp == 1) PP_BARRIER();

  int par = 0;
  int kt = 0;
  // One K-tile; I = its index inside the tile when it carries residual work (0..7), -1 otherwise.
  auto ktile = [&](auto i_tag) {
    constexpr int I = decltype(i_tag)::value;
    constexpr bool LD = RP && I >= 0 && I <= 5;                     // request sub-tile I
    constexpr bool WR = RP && I >= 1 && I <= 6;                     // sub-tile I - 1 -> LDS image
    constexpr bool RD2 = RP && (I == 1 || I == 2 || I == 4 || I == 5);  // sub-tile I - 1 (ni != 2): read back + add in phase 2
    constexpr bool RD0 = RP && (I == 4 || I == 7);                      // sub-tile I - 2 (ni == 2): read back + add in phase 0
    // ---- phase 0: read A, B_0; issue I1 of K-tile J+1; wait for I2 of this K-tile
    read_a();
    read_b(0);
    if constexpr (RD0) res_read();
    issue_i1(c1, par ^ 1);
    PP_WAIT_VM(5);
    PP_BARRIER();
    PP_WAIT_LGKM0();
    PP_STAMP();
    mma(0);
    if constexpr (RD0) res_add(std::integral_constant<int, (RD0 ? I - 2 : 0)>{});
    PP_BARRIER();
    // ---- phase 1: read B_1; issue I2 of K-tile J+1
    read_b(1);
    issue_i2(c1, par ^ 1);
    if constexpr (WR) res_write();
    if constexpr (LD) res_load(std::integral_constant<int, (LD ? I : 0)>{});
    PP_BARRIER();
    PP_WAIT_LGKM0();
    PP_STAMP();
    mma(1);
    PP_BARRIER();
    // ---- phase 2: read B_2; issue I0 of K-tile J+2 (the A rows of this K-tile were read in phase 0); wait for I0, I1 of J+1
    read_b(2);
    if constexpr (RD2) res_read();
    issue_i0(c2, par);
    // behind I0 / I1 of K-tile J+1: I2 of J+1 (2 pieces), where this K-tile carries them the 4 residual pieces of phase 1 (issued
    // behind I2), and I0 of K-tile J+2
    PP_WAIT_VM(5 + (LD ? 4 : 0));
    const bool last = kt + 1 == tile.ke;
    PP_BARRIER();
    PP_WAIT_LGKM0();
    PP_STAMP();
    mma(2);
    if constexpr (RD2) res_add(std::integral_constant<int, (RD2 ? I - 1 : 0)>{});
    if (!last) PP_BARRIER();
    c1 = c2;
    pp_cursor_next<BM, BN>(g, walk, c2);
    par ^= 1;
    set_rd(par);
    ++kt;
    kt_seen = true;
  };
  for (int q = 0;; ++q) {
    kt = tile.kb;
    if constexpr (RP) {  // (a "+ residual" tile has >= 8 K-tiles: ctk_launch_gemm_pp)
      ktile(std::integral_constant<int, 0>{});
      ktile(std::integral_constant<int, 1>{});
      ktile(std::integral_constant<int, 2>{});
      ktile(std::integral_constant<int, 3>{});
      ktile(std::integral_constant<int, 4>{});
      ktile(std::integral_constant<int, 5>{});
      ktile(std::integral_constant<int, 6>{});
      ktile(std::integral_constant<int, 7>{});
    }
    while (kt < tile.ke) ktile(std::integral_constant<int, -1>{});
    if (grp == 0) PP_BARRIER();
    PP_STAMP();  // epilogue begins
    const PPTile done = tile;
    const bool more = pp_tile<BM, BN>(g, walk, q + 1, tile);
    pp_epilogue<EPI, 2, 3>(g, acc, lane, done.bz, w_unscale, bias_lds, scr,
                           [&](int mi) { return done.m0 + wm * 64 + mi * 32; }, [&](int ni) { return done.n0 + wn * 32 + ni * 64; }, (dbg & 2) != 0);
    init_acc();
    PP_STAMP();  // epilogue done
    if (grp == 1) PP_BARRIER();
    if (!more) break;
  }
  if (grp == 0) PP_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_TRACE_DUMP();
  pp_clock_probe(1, tid);
}

int pp_num_cus() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    return cu;
  }();
  return n;
}

}  // namespace

// include/ctk.h: the historical name of ctk_set_option(CTK_OPT_GEMM_PP, mode)
extern "C" void ctk_gemm_pp_mode(int mode) { ctk_set_option(CTK_OPT_GEMM_PP, mode); }

#ifdef CTK_DEV
// dev tools (make dev; tools/gemm_lab trace / clock; not part of include/ctk.h).
// ctk_debug_pp_clock: {s_memtime, s_memrealtime (100 MHz)} at the start and at the end of workgroup 0 of the LAST persistent launch;
// ctk_debug_pp_trace: the wave timeline the last DBG launch with mode bit 6 recorded, [PP_TRACE_WGS][8 waves][PP_TRACE_STAMPS].
extern "C" int ctk_debug_pp_clock(unsigned long long* host_out4) {
  if (!host_out4) return CTK_E_NULL;
  const hipError_t e = hipMemcpyFromSymbol(host_out4, HIP_SYMBOL(g_pp_clock), 32);
  return e == hipSuccess ? CTK_OK : (int)e;
}

extern "C" int ctk_debug_pp_trace(unsigned long long* host_out, int n) {
  if (!host_out || n <= 0 || n > PP_TRACE_WGS * 8 * PP_TRACE_STAMPS) return CTK_E_SHAPE;
  const hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_trace), (size_t)n * 8);
  return e == hipSuccess ? CTK_OK : (int)e;
}
#endif

// the compile-time epilogues the persistent kernels are instantiated for (the PP_CASE list below)
static bool pp_epi_supported(int code) {
  return code == pp_epi(CTK_ACT_GELU_ERF, false, true, false, true) || code == pp_epi(CTK_ACT_NONE, false, true, false, true) ||
         code == pp_epi(CTK_ACT_NONE, false, false, true, false) || code == pp_epi(CTK_ACT_NONE, false, false, false, true) ||
         code == pp_epi(CTK_ACT_NONE, true, false, false, true) || code == pp_epi(CTK_ACT_GELU_TANH, false, true, false, true);
}

// Tail split (CTK_OPT_GEMM_PP bit 5, default ON).  A persistent launch deals whole 256-row tiles in rounds of #CUs: the N = 384
// Linears of a C3 window are 800 / 808 tiles = 3 full rounds + a round that is 1/8 full, and the launch lasts four tile times
// (profiles/r03_gemm_lab_streamk.txt: mlp.fc2 350 us for 768 tiles, 414 us for 800).  When the last round would be at most
// CTK_OPT_GEMM_TAIL_PCT % full (default 25), the persistent kernel gets the whole rounds only and the remaining row blocks go to the
// 64 x 64-tile kernel of gemm_f16x3.hip (same split-half arithmetic, same K order inside a row), which spreads them over
// every CU.  Rows are independent, so the result does not depend on where the cut is -- up to ONE rounding for the "+ residual"
// Linears (to_out, mlp.fc2): the persistent kernel adds residual * s into the accumulators during its first K-tiles, the 64 x 64
// kernel adds the residual after unscale and bias (tests/test_gpu_gemm_pp.py bounds the difference by 4e-6 relative).
//
// Returns CTK_OK after launching, or -1 when the shape is not one of the persistent kernels' (caller falls back).
int ctk_launch_gemm_pp(CtkGemmP& g, double flops, double bytes, hipStream_t s) {
  const int mode = ctk_opt(CTK_OPT_GEMM_PP);
  if ((mode & 1) == 0 || !g.a_split || !g.Wp) return -1;
  const bool t256 = (g.N % 256) == 0, t192 = !t256 && (g.N % 192) == 0;
  if ((!t256 && !t192) || g.N * 4 > PP_BIAS_BYTES) return -1;
  if (g.lda * 2 > 0xffffff) return -1;  // row offsets are formed with 32-bit arithmetic inside a tile
  const int BN = t256 ? 256 : 192;
  int mblocks = (g.M + 255) / 256;
  const int nblocks = g.N / BN;
  long tiles = (long)mblocks * nblocks * g.batch;
  const int cus = pp_num_cus();
  if (tiles < cus) return -1;  // less than one tile per CU (virtual-track GEMMs, short streaming windows): the 64x64 / 128x128 kernels fill the chip better (tools/gemm_lab.cpp)
  const int wgs = cus;
  const int code = pp_epi(g.act, g.resid != nullptr, g.c_split != 0, g.bias_rows != nullptr, g.bias != nullptr);
  if (g.resid && g.K < 8 * 32) return -1;  // the residual rides on a tile's first eight K-tiles (gemm_pp192_kernel)
  if (t256 && g.resid) return -1;  // residual preload of a 128-register accumulator tile spills; no Linear of the path has this shape
  if (!pp_epi_supported(code)) return -1;
  // ---- tail split: whole rounds here, the row blocks of a nearly empty last round as 64 x 64 tiles
  CtkGemmP tail = g;
  int tail_rows = 0;
  const long rem = tiles % wgs;
  if ((mode & 32) != 0 && g.batch == 1 && tiles > wgs && rem > 0 && rem * 100 <= (long)wgs * ctk_opt(CTK_OPT_GEMM_TAIL_PCT)) {
    const int mb_full = (int)((tiles - rem) / nblocks);  // row blocks of the whole rounds (rounded down to whole row blocks)
    const long rows_full = (long)mb_full * 256;
    if (mb_full > 0 && (!g.bias_rows || rows_full % g.bias_period == 0)) {
      tail_rows = (int)(g.M - rows_full);
      tail.M = tail_rows;
      tail.A = static_cast<const unsigned char*>(g.A) + rows_full * g.lda * 2;  // (a_split: lda counts halves)
      tail.C = static_cast<unsigned char*>(g.C) + rows_full * g.ldc * (g.c_split ? 2 : 4);
      if (g.resid) tail.resid = g.resid + rows_full * g.ldr;
      g.M = (int)rows_full;
      mblocks = mb_full;
      tiles = (long)mblocks * nblocks;
    }
  }
  const double frac = tail_rows ? (double)g.M / (double)(g.M + tail_rows) : 1.0;
  {
  const dim3 grid((unsigned)(tiles < wgs ? tiles : wgs)), blk(512);
  // every eligibility check is done: only now touch g and open the profile row (a fallback to gemm_f16x3.hip must not leave
  // a phantom gemm_sh_pp* row or a changed tile grid behind)
  g.mblocks = mblocks;
  g.nblocks = nblocks;
  char pname[40];
  snprintf(pname, sizeof(pname), "gemm_sh_pp%d_k%d_n%d", BN, g.K, g.N);
  CtkProfScope ps(pname, flops * frac, bytes * frac, s);
#ifdef CTK_DEV
  const bool dbgk = (mode & ~(1 | 32)) != 0;  // experiments (DBG instantiations): bit 1 = no stores, 2 = no MFMAs, 3 = timing jitter, 6 = wave timeline, 8.. = start stagger
#define PP_DBG_CASE(K, E) if (dbgk) hipLaunchKernelGGL((K<E, true>), grid, blk, 0, s, g, (int)tiles, mode); else
#else
#define PP_DBG_CASE(K, E)
#endif
#define PP_CASE(E)                                                                                             \
  case E:                                                                                                      \
    if (t256) { PP_DBG_CASE(gemm_pp256_kernel, E) hipLaunchKernelGGL((gemm_pp256_kernel<E, false>), grid, blk, 0, s, g, (int)tiles, 0); } \
    else { PP_DBG_CASE(gemm_pp192_kernel, E) {                                                                  \
      if (g.K > 768) hipLaunchKernelGGL((gemm_pp192_kernel<E, false, 1>), grid, blk, 0, s, g, (int)tiles, 0);  \
      else hipLaunchKernelGGL((gemm_pp192_kernel<E, false, 0>), grid, blk, 0, s, g, (int)tiles, 0); } }        \
    break
  switch (code) {
    PP_CASE(pp_epi(CTK_ACT_GELU_ERF, false, true, false, true));    // corr_mlp.fc1
    PP_CASE(pp_epi(CTK_ACT_NONE, false, true, false, true));        // corr_mlp.fc2 -> x (SH)
    PP_CASE(pp_epi(CTK_ACT_NONE, false, false, true, false));       // input_transform (+ per-frame bias rows)
    PP_CASE(pp_epi(CTK_ACT_NONE, false, false, false, true));       // to_q / to_kv
    PP_CASE(pp_epi(CTK_ACT_NONE, true, false, false, true));        // to_out / mlp.fc2 (+ residual)
    PP_CASE(pp_epi(CTK_ACT_GELU_TANH, false, true, false, true));   // mlp.fc1
    default:
      return -1;
  }
#undef PP_CASE
#undef PP_DBG_CASE
  CTK_HIP_CHECK_LAUNCH();
  }
  if (tail_rows) return ctk_launch_gemm_sh64(tail, flops * (1.0 - frac), bytes * (1.0 - frac), s);
  return CTK_OK;
}
