// feed_lab: which path feeds a 256 x 192 split-half GEMM tile fastest?  (round 6, VERDICT r5 item 1: "spend one gpurun call
// first on a MFMA-less feed micro-kernel (DMA only / VGPR loads only / both) in the GEMM's exact access pattern")
//
// A stand-alone replay of gemm_pp192_kernel's operand traffic (co-tracker_amd/csrc/gemm_pp.hip): one persistent 8-wave workgroup
// per CU walks over 256-row x 192-column tiles of an SH matrix A[M][K] and packed weights W[384][K]; per K-tile (one 128-byte
// line per row) it brings 32 KiB of A and 24 KiB of W on chip.  What differs between the variants is the PATH:
//   A: 0 none | 1 LDS-DMA (global_load_lds_dwordx4: today's kernel) | 2 coalesced global_load_dwordx4 -> VGPR -> ds_write_b128
//      (same swizzled LDS image) | 3 global_load_dwordx4 straight into MFMA fragment order (row per lane, no LDS at all)
//   W: 0 none | 1 LDS-DMA | 2 coalesced -> VGPR -> ds_write_b128
//   FRAG: the wave's 20 ds_read_b128 fragment reads per K-tile;  MF: the 36 v_mfma_f32_32x32x16_f16 per wave and K-tile.
// Not the product kernel: all waves in phase, two plain barriers per K-tile, prefetch distance one K-tile, no epilogue.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/feed_lab.cpp -o tools/feed_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include <algorithm>

#define HIP_OK(x)                                                                                    \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) {                                                                          \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));            \
      exit(2);                                                                                       \
    }                                                                                                \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ unsigned xcd_remap(unsigned pid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u, xcd = pid & 7u, idx = pid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
__device__ __forceinline__ void dma16(const unsigned char* base, unsigned voff, unsigned char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gptr_t)(base + voff), (lptr_t)lds_dst, 16, 0, 0);
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define WAIT_VM(N)                                           \
  do {                                                       \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); \
    FENCE();                                                 \
  } while (0)
#define WAIT_LGKM0()                                     \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    FENCE();                                             \
  } while (0)
#define BARRIER()                      \
  do {                                 \
    FENCE();                           \
    __builtin_amdgcn_s_barrier();      \
    FENCE();                           \
  } while (0)

struct P {
  const unsigned char* A;
  const unsigned char* W;
  int M, KT, tiles;
  float* sink;
  unsigned long long* clk;
  int store;
};

constexpr int SLOT = 57344;

template <int AM, int WM, int FRAG, int MF>
__global__ __launch_bounds__(512) void feed_kernel(P g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[163840];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const unsigned l3 = lane >> 3, l4 = lane >> 4, l7 = lane & 7;
  const int KT = g.KT;
  const unsigned lda_b = (unsigned)KT * 128, ldw_b = (unsigned)KT * 128;
  if (blockIdx.x == 0 && tid < 64) {
    unsigned long long a, b;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : : "memory");
    if (tid == 0) {
      g.clk[0] = a;
      g.clk[1] = b;
    }
  }
  constexpr int NA = AM == 1 ? 4 : AM == 2 ? 4 : AM == 3 ? 8 : 0;
  constexpr int NW = WM ? 3 : 0;
  constexpr int NV = NA + NW;

  f32x4 sa[2][AM == 3 ? 8 : 4];  // staged A (AM 2: coalesced pieces, AM 3: the fragments themselves)
  f32x4 sw[2][3];
  f16x8 fa[2][2][2], fb[2][2];
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  unsigned xacc = 0;
  {  // pseudo-random fragments for the variants that never read any
    unsigned s = (unsigned)tid * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = s * 1664525u + 1013904223u;
        t[e] = (s & 0x8fff8fffu) | 0x30003000u;
      }
      fa[i >> 2][(i >> 1) & 1][i & 1] = __builtin_bit_cast(f16x8, t);
      if (i < 4) fb[i >> 1][i & 1] = __builtin_bit_cast(f16x8, t);
    }
  }

  const int fsw = (r32 >> 1) & 7;
  const int G = gridDim.x;
  for (int q = 0;; ++q) {
    const int first = q * G;
    if (first >= g.tiles) break;
    const int n_r = min(G, g.tiles - first);
    if ((int)blockIdx.x >= n_r) break;
    const unsigned tile = first + xcd_remap(blockIdx.x, n_r);
    const int nb = tile & 1, mb = tile >> 1;
    const unsigned lim = (unsigned)min(255, g.M - 1 - mb * 256);
    const unsigned char* a0 = g.A + (long)mb * 256 * lda_b;
    const unsigned char* w0 = g.W + (long)nb * 192 * ldw_b;

    auto issue = [&](const int kt, const int set) {
      const unsigned char* a = a0 + kt * 128;
      const unsigned char* w = w0 + kt * 128;
      unsigned char* slot = lds + set * SLOT;
      if (AM == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int piece = 4 * wave + e;
          const unsigned r = min((unsigned)(8 * piece) + l3, lim);
          dma16(a, r * lda_b + ((l7 ^ ((4 * piece + l4) & 7)) << 4), slot + piece * 1024);
        }
      } else if (AM == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned r = min((unsigned)(32 * wave + 8 * e) + l3, lim);
          const unsigned char* ptr = a + r * lda_b + (l7 << 4);
          f32x4 t;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(ptr) : "memory");
          sa[set][e] = t;
        }
      } else if (AM == 3) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int c = 0; c < 4; ++c) {  // chunk 2c + half = (p*4 + j*2 + half), c = p*2 + j
            const unsigned r = min((unsigned)(wm * 64 + mi * 32 + r32), lim);
            const unsigned char* ptr = a + r * lda_b + ((2 * c + half) << 4);
            f32x4 t;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(ptr) : "memory");
            sa[set][mi * 4 + c] = t;
          }
      }
      if (WM == 1) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const int piece = 3 * wave + e;
          const unsigned r = (unsigned)(8 * piece) + l3;
          dma16(w, r * ldw_b + ((l7 ^ ((4 * piece + l4) & 7)) << 4), slot + 32768 + piece * 1024);
        }
      } else if (WM == 2) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const unsigned r = (unsigned)(24 * wave + 8 * e) + l3;
          const unsigned char* ptr = w + r * ldw_b + (l7 << 4);
          f32x4 t;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(ptr) : "memory");
          sw[set][e] = t;
        }
      }
    };
    auto commit = [&](const int set) {  // staged registers -> the swizzled LDS image
      unsigned char* slot = lds + set * SLOT;
      if (AM == 2) {
        asm volatile("" : "+v"(sa[set][0]), "+v"(sa[set][1]), "+v"(sa[set][2]), "+v"(sa[set][3]));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned r = (unsigned)(32 * wave + 8 * e) + l3;
          *reinterpret_cast<f32x4*>(slot + r * 128 + ((l7 ^ ((r >> 1) & 7)) << 4)) = sa[set][e];
        }
      }
      if (AM == 3) {
        asm volatile("" : "+v"(sa[set][0]), "+v"(sa[set][1]), "+v"(sa[set][2]), "+v"(sa[set][3]), "+v"(sa[set][4]), "+v"(sa[set][5]), "+v"(sa[set][6]),
                     "+v"(sa[set][7]));
      }
      if (WM == 2) {
        asm volatile("" : "+v"(sw[set][0]), "+v"(sw[set][1]), "+v"(sw[set][2]));
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const unsigned r = (unsigned)(24 * wave + 8 * e) + l3;
          *reinterpret_cast<f32x4*>(slot + 32768 + r * 128 + ((l7 ^ ((r >> 1) & 7)) << 4)) = sw[set][e];
        }
      }
    };

    issue(0, 0);
    auto ktile = [&](const int kt, auto set_tag) {
      constexpr int set = decltype(set_tag)::value;
      // (the last K-tile re-requests K-tile KT-1: harmless duplicate, keeps the counts uniform)
      issue(min(kt + 1, KT - 1), set ^ 1);
      WAIT_VM(NV);
      commit(set);
      WAIT_LGKM0();
      BARRIER();
      const unsigned char* slot = lds + set * SLOT;
      if (FRAG) {
        if (AM != 3) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int p = 0; p < 2; ++p)
                fa[mi][j][p] = *reinterpret_cast<const f16x8*>(slot + (wm * 64 + mi * 32 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
        }
      }
      if (AM == 3) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) fa[mi][j][p] = __builtin_bit_cast(f16x8, sa[set][mi * 4 + p * 2 + j]);
      }
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        if (FRAG) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
              fb[j][p] = *reinterpret_cast<const f16x8*>(slot + 32768 + n * 8192 + (wn * 32 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
          WAIT_LGKM0();
        }
        if (MF) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
                acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][term == 0 ? 1 : 0], fa[mi][j][term == 1 ? 1 : 0], acc[mi][n], 0, 0, 0);
        } else if (FRAG || AM == 3) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              xacc ^= __builtin_bit_cast(u32x4, fb[j][p]).x;
#pragma unroll
              for (int mi = 0; mi < 2; ++mi) xacc ^= __builtin_bit_cast(u32x4, fa[mi][j][p]).y;
            }
        }
      }
      BARRIER();
    };
    for (int kt = 0; kt < KT; kt += 2) {
      ktile(kt, std::integral_constant<int, 0>{});
      if (kt + 1 < KT) ktile(kt + 1, std::integral_constant<int, 1>{});
    }
    WAIT_VM(0);
  }
  if (g.store) {  // never: keeps the accumulators alive
    float s = (float)xacc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    g.sink[blockIdx.x * 512 + tid] = s;
  }
  if (blockIdx.x == 0 && tid < 64) {
    unsigned long long a, b;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : : "memory");
    if (tid == 0) {
      g.clk[2] = a;
      g.clk[3] = b;
    }
  }
}


// ---- second kernel: LDS-DMA only, ONE stream over all of the workgroup's K-tiles (no drain at tile boundaries), A requested
// PD K-tiles ahead (ring of PD + 1 A slots), W one K-tile ahead (2 slots); stream order = order of need (W(g+1), A(g+PD)).
template <int PD, int AON, int WON, int FRAG, int MF, int AUX, int NBLK = 3, int BLK = 0>
__global__ __launch_bounds__(512) void feed2_kernel(P g) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[163840];
  constexpr int WBASE = (PD + 1) * 32768;
  static_assert(WBASE + (WON ? 2 * NBLK * 8192 : 0) <= 163840, "LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;
  const unsigned l3 = lane >> 3, l4 = lane >> 4, l7 = lane & 7;
  const int KT = g.KT;
  const unsigned lda_b = (unsigned)KT * 128, ldw_b = (unsigned)KT * 128;
  if (blockIdx.x == 0 && tid < 64) {
    unsigned long long a, b;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : : "memory");
    if (tid == 0) {
      g.clk[0] = a;
      g.clk[1] = b;
    }
  }
  constexpr int NA = AON ? 4 : 0, NW = WON ? NBLK : 0;
  f16x8 fa[2][2][2], fb[2][2];
  f32x16 acc[2][NBLK];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  unsigned xacc = 0;
  {
    unsigned s = (unsigned)tid * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      u32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s = s * 1664525u + 1013904223u;
        t[e] = (s & 0x8fff8fffu) | 0x30003000u;
      }
      fa[i >> 2][(i >> 1) & 1][i & 1] = __builtin_bit_cast(f16x8, t);
      if (i < 4) fb[i >> 1][i & 1] = __builtin_bit_cast(f16x8, t);
    }
  }
  const int fsw = (r32 >> 1) & 7;
  const int G = gridDim.x;
  const int my_tiles = g.tiles / G;  // whole rounds (host)
  const int GT = my_tiles * KT;
  struct Cur {
    int q, kt;
    const unsigned char* a;
    const unsigned char* w;
    unsigned lim;
  };
  auto cur_at = [&](int q) {
    const unsigned tile = q * G + xcd_remap(blockIdx.x, G);
    const int nb = NBLK == 6 ? 0 : (tile & 1), mb = NBLK == 6 ? tile : (tile >> 1);
    Cur c;
    c.q = q;
    c.kt = 0;
    c.lim = (unsigned)min(255, g.M - 1 - mb * 256);
    c.a = g.A + (long)mb * 256 * lda_b;  // (BLK: the row block's KT x 32 KiB are contiguous too: same base)
    c.w = g.W + (long)nb * 192 * ldw_b;
    return c;
  };
  auto cur_next = [&](Cur& c) {
    if (c.kt + 1 < KT) {
      c.kt += 1;
      c.a += BLK ? 32768 : 128;
      c.w += 128;
    } else if (c.q + 1 < my_tiles) {
      c = cur_at(c.q + 1);
    }
  };
  auto issue_a = [&](const Cur& c, const int slot) {
    if (!AON) return;
    unsigned char* dst = lds + slot * 32768;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int piece = 4 * wave + e;
      const unsigned r = min((unsigned)(8 * piece) + l3, c.lim);
      __builtin_amdgcn_global_load_lds((gptr_t)(c.a + r * (BLK ? 128u : lda_b) + ((l7 ^ ((4 * piece + l4) & 7)) << 4)), (lptr_t)(dst + piece * 1024), 16, 0, AUX);
    }
  };
  auto issue_w = [&](const Cur& c, const int slot) {
    if (!WON) return;
    unsigned char* dst = lds + WBASE + slot * (NBLK * 8192);
#pragma unroll
    for (int e = 0; e < NBLK; ++e) {
      const int piece = NBLK * wave + e;
      const unsigned r = (unsigned)(8 * piece) + l3;
      dma16(c.w, r * ldw_b + ((l7 ^ ((4 * piece + l4) & 7)) << 4), dst + piece * 1024);
    }
  };
  Cur ca = cur_at(0), cw = ca;
  issue_w(cw, 0);
  cur_next(cw);
  int aslot_issue = 0;
#pragma unroll
  for (int i = 0; i < PD; ++i) {
    issue_a(ca, aslot_issue);
    cur_next(ca);
    aslot_issue = aslot_issue == PD ? 0 : aslot_issue + 1;
  }
  int aslot = 0;
  for (int gk = 0; gk < GT; ++gk) {
    issue_w(cw, (gk + 1) & 1);
    cur_next(cw);
    issue_a(ca, aslot_issue);
    cur_next(ca);
    aslot_issue = aslot_issue == PD ? 0 : aslot_issue + 1;
    if (PD == 1) WAIT_VM(NA + NW);
    else WAIT_VM(2 * NA + NW);
    BARRIER();
    const unsigned char* sa_ = lds + aslot * 32768;
    const unsigned char* sw_ = lds + WBASE + (gk & 1) * (NBLK * 8192);
    if (NBLK == 6 && FRAG && MF) {
      // register-lean order for the 256 x 384 tile (192 accumulators): per k-step j the A fragments of j only (16 registers) against
      // all six column blocks, B fragments per (n, j) (8 registers); per accumulator the MFMA order is unchanged (j outer, term inner)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x8 a0[2], a1[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a0[p] = *reinterpret_cast<const f16x8*>(sa_ + (wm * 64 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
          a1[p] = *reinterpret_cast<const f16x8*>(sa_ + (wm * 64 + 32 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
        }
#pragma unroll
        for (int n = 0; n < NBLK; ++n) {
          f16x8 b[2];
#pragma unroll
          for (int p = 0; p < 2; ++p)
            b[p] = *reinterpret_cast<const f16x8*>(sw_ + n * 8192 + (wn * 32 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
#pragma unroll
          for (int term = 0; term < 3; ++term) {
            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[term == 0 ? 1 : 0], a0[term == 1 ? 1 : 0], acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[term == 0 ? 1 : 0], a1[term == 1 ? 1 : 0], acc[1][n], 0, 0, 0);
          }
        }
      }
    } else {
    if (FRAG) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            fa[mi][j][p] = *reinterpret_cast<const f16x8*>(sa_ + (wm * 64 + mi * 32 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
    }
#pragma unroll
    for (int n = 0; n < NBLK; ++n) {
      if (FRAG) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            fb[j][p] = *reinterpret_cast<const f16x8*>(sw_ + n * 8192 + (wn * 32 + r32) * 128 + (((p * 4 + j * 2 + half) ^ fsw) << 4));
        WAIT_LGKM0();
      }
      if (MF) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j][term == 0 ? 1 : 0], fa[mi][j][term == 1 ? 1 : 0], acc[mi][n], 0, 0, 0);
      } else if (FRAG) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            xacc ^= __builtin_bit_cast(u32x4, fb[j][p]).x;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xacc ^= __builtin_bit_cast(u32x4, fa[mi][j][p]).y;
          }
      }
    }
    }
    BARRIER();
    aslot = aslot == PD ? 0 : aslot + 1;
  }
  WAIT_VM(0);
  if (g.store) {
    float s = (float)xacc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    g.sink[blockIdx.x * 512 + tid] = s;
  }
  if (blockIdx.x == 0 && tid < 64) {
    unsigned long long a, b;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a), "=s"(b) : : "memory");
    if (tid == 0) {
      g.clk[2] = a;
      g.clk[3] = b;
    }
  }
}

typedef void (*kern_t)(P);
struct Variant {
  const char* name;
  kern_t k;
  int am, wm, frag, mf;
};
#define V(AM, WM, FR, MF) \
  { "A" #AM " W" #WM " frag" #FR " mfma" #MF, feed_kernel<AM, WM, FR, MF>, AM, WM, FR, MF }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 103424;
  const int K = argc > 2 ? atoi(argv[2]) : 1536;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int KT = K / 32;
  HIP_OK(hipSetDevice(0));
  int cus = 0;
  HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const size_t a_bytes = (size_t)M * K * 4, w_bytes = (size_t)384 * K * 4;
  unsigned char *dA, *dW;
  float* dsink;
  unsigned long long* dclk;
  HIP_OK(hipMalloc(&dA, a_bytes + 4096));
  HIP_OK(hipMalloc(&dW, w_bytes + 4096));
  HIP_OK(hipMalloc(&dsink, (size_t)cus * 512 * 4));
  HIP_OK(hipMalloc(&dclk, 64));
  {  // random halves in [-2, 2): realistic toggling for the MFMA variants
    std::vector<unsigned short> h(a_bytes / 2);
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      v = (unsigned short)(((s >> 16) & 0x8fff) | 0x3000);
    }
    if (getenv("FEED_ZERO")) std::fill(h.begin(), h.end(), (unsigned short)0);
    if (getenv("FEED_CONST")) std::fill(h.begin(), h.end(), (unsigned short)0x3c00);
    HIP_OK(hipMemcpy(dA, h.data(), a_bytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dW, h.data(), w_bytes, hipMemcpyHostToDevice));
  }
  const int mblocks = (M + 255) / 256;
  int tiles = mblocks * 2;
  if (getenv("FEED_WHOLE_ROUNDS")) tiles = tiles / cus * cus;  // what the tail split hands the persistent kernel
  P p{dA, dW, M, KT, tiles, dsink, dclk, 0};
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  const Variant vs[] = {
      V(1, 1, 0, 0), V(2, 1, 0, 0), V(3, 1, 0, 0), V(2, 2, 0, 0), V(1, 0, 0, 0), V(0, 1, 0, 0), V(2, 0, 0, 0), V(3, 0, 0, 0), V(0, 2, 0, 0),
      V(1, 1, 1, 0), V(2, 1, 1, 0), V(3, 1, 1, 0), V(2, 2, 1, 0),
      V(1, 1, 1, 1), V(2, 1, 1, 1), V(3, 1, 1, 1), V(2, 2, 1, 1), V(0, 0, 0, 1),
  };
  const Variant v2[] = {
#define V2(PD, A, W, FR, MF, AUX) { "feed2 PD" #PD " A" #A " W" #W " frag" #FR " mfma" #MF " aux" #AUX, feed2_kernel<PD, A, W, FR, MF, AUX>, A, W, FR, MF }
      V2(1, 1, 0, 0, 0, 0), V2(2, 1, 0, 0, 0, 0), V2(3, 1, 0, 0, 0, 0), V2(2, 1, 0, 0, 0, 2), V2(3, 1, 0, 0, 0, 2),
      V2(1, 1, 1, 0, 0, 0), V2(2, 1, 1, 0, 0, 0), V2(2, 1, 1, 0, 0, 2),
      V2(1, 1, 1, 1, 0, 0), V2(2, 1, 1, 1, 0, 0),
      V2(1, 1, 1, 1, 1, 0), V2(2, 1, 1, 1, 1, 0), V2(2, 1, 1, 1, 1, 2), V2(1, 0, 0, 0, 1, 0),
#define V2B(PD, A, W, FR, MF) { "feed2 BLOCKED-A PD" #PD " A" #A " W" #W " frag" #FR " mfma" #MF, feed2_kernel<PD, A, W, FR, MF, 0, 3, 1>, A, W, FR, MF }
      V2B(1, 1, 0, 0, 0), V2B(2, 1, 0, 0, 0), V2B(1, 1, 1, 0, 0), V2B(1, 1, 1, 1, 0), V2B(1, 1, 1, 1, 1), V2B(2, 1, 1, 1, 1),
      V2(1, 0, 0, 1, 1, 0), V2(1, 1, 1, 0, 1, 0), V2(1, 1, 0, 0, 1, 0), V2(1, 0, 1, 0, 1, 0), V2(1, 0, 1, 1, 1, 0), V2(1, 1, 0, 1, 1, 0),
  };
  printf("feed_lab M=%d K=%d tiles=%d (256x192) on %d CUs, %d reps\n", M, K, tiles, cus, reps);
  printf("%-26s %10s %9s %9s %9s %10s\n", "variant", "us/launch", "TB/s", "GHz", "B/clk/CU", "TF/s(x3)");
  for (const Variant& v : vs) {
    if (getenv("FEED_SKIP1")) break;
    if (getenv("FEED_ONLY") && !strstr(getenv("FEED_ONLY"), v.name)) continue;
    const double bytes = (double)tiles * KT * ((v.am ? 32768.0 : 0.0) + (v.wm ? 24576.0 : 0.0));
    const dim3 grid(tiles < cus ? tiles : cus), blk(512);
    int nrep = reps;
    if (reps <= 0) {  // duration mode: -reps milliseconds of warm-up, then as many timed
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipStreamSynchronize(st));
      float ms4 = 0;
      HIP_OK(hipEventElapsedTime(&ms4, e0, e1));
      nrep = (int)(-reps / (ms4 / 4)) + 1;
      for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p);
    } else {
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p);
      HIP_OK(hipStreamSynchronize(st));
    }
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p);
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / nrep;
    unsigned long long c[4];
    HIP_OK(hipMemcpy(c, dclk, 32, hipMemcpyDeviceToHost));
    const double ghz = (double)(c[2] - c[0]) / ((double)(c[3] - c[1]) * 10.0);  // s_memrealtime: 100 MHz
    const double tbs = bytes / us * 1e-6;
    const double bpc = bytes / (us * 1e-6) / (ghz * 1e9) / cus;
    const double tf = v.mf ? (double)tiles * KT * 8 * 36 * 32768.0 / us * 1e-6 : 0.0;  // issued f16 MFMA flops
    printf("%-26s %10.1f %9.2f %9.3f %9.1f %10.0f\n", v.name, us, tbs, ghz, bpc, tf);
    fflush(stdout);
  }
  const int grid2 = getenv("FEED_GRID") ? atoi(getenv("FEED_GRID")) : cus;
  if (getenv("FEED_W384")) {  // 256 x 384 whole-row tile (wave tile 64 x 192), grid = #CUs, FEED_ROUNDS rounds
    const int rounds = getenv("FEED_ROUNDS") ? atoi(getenv("FEED_ROUNDS")) : 1;
    P p3 = p;
    p3.tiles = cus * rounds;
    const Variant v3[] = {
#define V3(PD, A, W, FR, MF) { "feed384 PD" #PD " A" #A " W" #W " frag" #FR " mfma" #MF, feed2_kernel<PD, A, W, FR, MF, 0, 6>, A, W, FR, MF }
        V3(1, 1, 1, 0, 0), V3(1, 1, 1, 1, 0), V3(1, 1, 1, 1, 1), V3(1, 0, 0, 0, 1), V3(1, 0, 0, 1, 1), V3(1, 1, 1, 0, 1),
    };
    printf("feed384: %d tiles of 256 x 384 on %d workgroups (M must be >= %d)\n", p3.tiles, cus, p3.tiles * 256);
    for (const Variant& v : v3) {
      const double bytes = (double)p3.tiles * KT * ((v.am ? 32768.0 : 0.0) + (v.wm ? 49152.0 : 0.0));
      const dim3 grid(cus), blk(512);
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p3);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipStreamSynchronize(st));
      float ms4 = 0;
      HIP_OK(hipEventElapsedTime(&ms4, e0, e1));
      const int nrep = (int)(300 / (ms4 / 4)) + 1;
      for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p3);
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p3);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipStreamSynchronize(st));
      HIP_OK(hipGetLastError());
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / nrep;
      unsigned long long c[4];
      HIP_OK(hipMemcpy(c, dclk, 32, hipMemcpyDeviceToHost));
      const double ghz = (double)(c[2] - c[0]) / ((double)(c[3] - c[1]) * 10.0);
      const double tf = v.mf ? (double)p3.tiles * KT * 8 * 72 * 32768.0 / us * 1e-6 : 0.0;
      printf("%-40s %10.1f %9.2f %9.3f %9.1f %10.0f\n", v.name, us, bytes / us * 1e-6, ghz, bytes / (us * 1e-6) / (ghz * 1e9) / cus, tf);
      fflush(stdout);
    }
    return 0;
  }
  const int tiles2 = getenv("FEED_ROUNDS") ? grid2 * atoi(getenv("FEED_ROUNDS")) : tiles / grid2 * grid2;
  P p2 = p;
  p2.tiles = tiles2;
  printf("feed2: one stream per workgroup, %d tiles (whole rounds) on %d workgroups\n", tiles2, grid2);
  for (const Variant& v : v2) {
    if (getenv("FEED_ONLY2") && !strstr(v.name, getenv("FEED_ONLY2"))) continue;
    const double bytes = (double)tiles2 * KT * ((v.am ? 32768.0 : 0.0) + (v.wm ? 24576.0 : 0.0));
    const dim3 grid(grid2), blk(512);
    int nrep = reps;
    if (reps <= 0) {  // duration mode: -reps milliseconds of warm-up, then as many timed
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p2);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipStreamSynchronize(st));
      float ms4 = 0;
      HIP_OK(hipEventElapsedTime(&ms4, e0, e1));
      nrep = (int)(-reps / (ms4 / 4)) + 1;
      for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p2);
    } else {
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p2);
      HIP_OK(hipStreamSynchronize(st));
    }
    HIP_OK(hipEventRecord(e0, st));
    for (int i = 0; i < nrep; ++i) hipLaunchKernelGGL(v.k, grid, blk, 0, st, p2);
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / nrep;
    unsigned long long c[4];
    HIP_OK(hipMemcpy(c, dclk, 32, hipMemcpyDeviceToHost));
    const double ghz = (double)(c[2] - c[0]) / ((double)(c[3] - c[1]) * 10.0);
    const double tbs = bytes / us * 1e-6;
    const double bpc = bytes / (us * 1e-6) / (ghz * 1e9) / cus;
    const double tf = v.mf ? (double)tiles2 * KT * 8 * 36 * 32768.0 / us * 1e-6 : 0.0;
    printf("%-40s %10.1f %9.2f %9.3f %9.1f %10.0f\n", v.name, us, tbs, ghz, bpc, tf);
    fflush(stdout);
  }
  return 0;
}
