// Multi-head attention for the factored time / virtual-track transformer
// (Attention.forward, blocks.py:379-398: softmax(q k^T * 48^-0.5) v, 8 heads x 48).
//
// Shapes on this path are small and irregular: time attention has S = 16..120 keys per
// track, point->virtual has 64 keys, virtual->point has 64 queries over N keys.  All of it is
// < 2 % of the update's FLOPs, so the kernel favours generality and exact fp32
// math over MFMA: one wavefront per workgroup, one query per lane, the K/V rows of the
// lane's batch streamed through LDS in chunks of KC keys and read back as broadcast
// ds_read_b128, online (running max / sum) softmax per lane, and an optional split of the key
// range over several workgroups whose (m, l, acc) partials are merged by a second kernel
// (needed for virtual->point, where 64 queries x N keys would otherwise fill only 8*S waves).
// Row addressing is strided (row(b,i) = b*bs + i*is) so that the time axis and the track axis
// of the [(N+64), S, 384] token tensor are both reached without the reference's
// permute+contiguous copies (cotracker.py:494,504,520).
#include "ctk_common.h"
#include "ctk_options.h"
#include "ctk_profile.h"
#include "gemm_params.h"
#include <cstdlib>

namespace {

constexpr int HD = CTK_HEAD_DIM;  // 48
constexpr int KC = 16;            // keys per LDS chunk
constexpr int MAXB = 8;           // max batches packed into one wave (n1 >= 8)
constexpr int BPAD = 4;           // floats between batches in LDS (bank spread)
constexpr float NEG_MAX = -3.402823466e+38f;  // -torch.finfo(float32).max: the reference's mask bias (cotracker.py:571)

struct AttnP {
  const float* q; long q_ld, q_bs, q_is;
  const float* k; const float* v; long kv_ld, kv_bs, kv_is;
  float* out; long o_ld, o_bs, o_is;
  int nbatch, n1, n2;
  int splits, keys_per_split;
  float* partial;
  int o_split; // out is SH halves (o_ld = halves per row)
  int bpw;     // batches per wave (n1 < 64) or 1
  int qtiles;  // ceil(n1/64) when n1 >= 64
  float scale;   // 48^-0.5
  float scale2;  // 48^-0.5 * log2(e): the MFMA kernels keep scores in the log2 domain (raw v_exp_f32, no multiply)
  int log2m;     // partial maxima are in the log2 domain (attention_q64_kernel) -> the merge uses exp2
  const uint8_t* kmask;  // [n2] or null: masked keys get logit -FLT_MAX (CoTracker2)
  const uint8_t* qmask;  // [n1] or null: a masked query attends uniformly
};

__global__ __launch_bounds__(64) void attention_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][bpw][KC*HD+BPAD]
  float* lk = smem;
  float* lv = smem + p.bpw * (KC * HD + BPAD);
  const int lane = threadIdx.x;
  const int head = blockIdx.y;
  const int split = blockIdx.z;

  int b0, bl, qi;
  if (p.n1 >= 64) {
    b0 = blockIdx.x / p.qtiles;
    bl = 0;
    qi = (blockIdx.x % p.qtiles) * 64 + lane;
  } else {
    b0 = blockIdx.x * p.bpw;
    bl = lane / p.n1;
    qi = lane % p.n1;
  }
  const int nb_here = min(p.bpw, p.nbatch - b0);
  const bool active = (bl < nb_here) && (qi < p.n1);
  const int myb = b0 + min(bl, nb_here - 1);
  const int myq = min(qi, p.n1 - 1);

  // query row -> registers, pre-scaled
  float qr[HD];
  {
    const float* qp = p.q + (myb * p.q_bs + myq * p.q_is) * p.q_ld + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qp + d);
      qr[d] = t[0] * p.scale; qr[d + 1] = t[1] * p.scale; qr[d + 2] = t[2] * p.scale; qr[d + 3] = t[3] * p.scale;
    }
  }
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.0f;
  float m = -INFINITY, l = 0.0f;
  const bool qmasked = p.qmask && !p.qmask[myq];

  const int kbeg = split * p.keys_per_split;
  const int kend = min(p.n2, kbeg + p.keys_per_split);
  const int bstride = KC * HD + BPAD;
  const float* myk = lk + min(bl, nb_here - 1) * bstride;
  const float* myv = lv + min(bl, nb_here - 1) * bstride;

  for (int k0 = kbeg; k0 < kend; k0 += KC) {
    const int kn = min(KC, kend - k0);
    __syncthreads();
    // cooperative stage: nb_here batches x kn keys x 12 float4 for K and V
    const int items = nb_here * kn * (HD / 4);
    for (int i = lane; i < items; i += 64) {
      const int d4 = i % (HD / 4);
      const int kk = (i / (HD / 4)) % kn;
      const int bb = i / ((HD / 4) * kn);
      const long row = ((long)(b0 + bb) * p.kv_bs + (long)(k0 + kk) * p.kv_is) * p.kv_ld + head * HD + d4 * 4;
      *reinterpret_cast<f32x4*>(&lk[bb * bstride + kk * HD + d4 * 4]) = *reinterpret_cast<const f32x4*>(p.k + row);
      *reinterpret_cast<f32x4*>(&lv[bb * bstride + kk * HD + d4 * 4]) = *reinterpret_cast<const f32x4*>(p.v + row);
    }
    __syncthreads();

    float s[KC];
    float cmax = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      float dot = 0.0f;
      if (kk < kn) {
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(myk + kk * HD + d);
          dot += qr[d] * t[0] + qr[d + 1] * t[1] + qr[d + 2] * t[2] + qr[d + 3] * t[3];
        }
        if (p.kmask && !p.kmask[k0 + kk]) dot = NEG_MAX;
        if (qmasked) dot = NEG_MAX;
        cmax = fmaxf(cmax, dot);
      } else {
        dot = -INFINITY;
      }
      s[kk] = dot;
    }
    const float mnew = fmaxf(m, cmax);
    const float alpha = expf(m - mnew);  // m = -inf on the first chunk -> 0
    l *= alpha;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= alpha;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      if (kk < kn) {
        const float pr = expf(s[kk] - mnew);
        l += pr;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(myv + kk * HD + d);
          acc[d] += pr * t[0]; acc[d + 1] += pr * t[1]; acc[d + 2] += pr * t[2]; acc[d + 3] += pr * t[3];
        }
      }
    }
    m = mnew;
  }

  if (!active) return;
  if (p.splits == 1) {
    const float inv = 1.0f / l;
    float* op = p.out + (myb * p.o_bs + myq * p.o_is) * p.o_ld + head * HD;
    _Float16* oh = reinterpret_cast<_Float16*>(p.out) + (myb * p.o_bs + myq * p.o_is) * p.o_ld;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      f32x4 t = {acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv};
      if (p.o_split) {
        f16x4 hi, lo;
        ctk_split4(t, hi, lo);
        _Float16* dst = oh + ctk_sh_col(head * HD + d);
        *reinterpret_cast<f16x4*>(dst) = hi;
        *reinterpret_cast<f16x4*>(dst + 32) = lo;
      } else {
        *reinterpret_cast<f32x4*>(op + d) = t;
      }
    }
  } else {
    float* pp = p.partial + ((((long)split * p.nbatch + myb) * CTK_HEADS + head) * p.n1 + myq) * (HD + 2);
    pp[0] = m;
    pp[1] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) pp[2 + d] = acc[d];
  }
}

// ---------------------------------------------------------------------------------------------------------
// MFMA kernels for the two space-attention shapes that carry ~90 % of the attention flops (4*S*64*N*384 each per
// layer): points <- virtual (N queries x 64 keys) and virtual <- points (64 queries x N keys).  Products use the
// split-half scheme of gemm_f16x3.hip (x = hi + lo IEEE halves, 3 x v_mfma_f32_32x32x16_f16, f32 accumulate,
// ~2^-21 relative per product), so the result stays fp32-class while the contraction leaves the VALU.
//
// Both kernels compute the score tile TRANSPOSED, S'[key][query] = K . Q^T (K = first MFMA operand), so that a
// lane owns ONE query (column lane & 31) and 16 of the tile's 32 keys in registers (row 8*(reg>>2) + 4*(lane>>5) +
// (reg&3)); the softmax row reductions are then per-lane plus one exchange with lane ^ 32.  The probabilities
// feed the second MFMA  O^T[dim][query] = V^T . P^T  straight from those registers as its B operand: B wants, for
// k-step s and lane half h, 8 consecutive k values -- we simply DEFINE the k order inside each block of 16 keys as
// k = 8h + e  <->  key = 16s + 8(e>>2) + 4h + (e&3)  (what the accumulator layout hands us) and build the V^T
// operand with the same permutation.  No LDS round trip, no shuffles for P.
// P is scaled by 2^12 before the split so that its lo half stays a normal f16 number.
// ---------------------------------------------------------------------------------------------------------
constexpr int KP = 56;              // K image row pitch in halves: 7 x 16-byte slots -> conflict-free ds_read_b128
constexpr int VP = 72;              // V^T image row pitch in halves: 9 slots
constexpr float PSCALE = 4096.0f;
constexpr int QT_PER_WG = 4;        // 128-query tiles per workgroup of attention_kv64_kernel

// ---- SH output of one head of a 32-query tile: whole lines instead of 16-byte crumbs ---------------------------
// The accumulator layout hands a lane (query r32, half) 4 consecutive output columns per register quad; written from
// there a store instruction touches 32 rows x 16 bytes, and a CU retires such an instruction only every ~100 cycles
// (tools/gemm_lab.cpp store experiments): 12 of them per 32-query job were half of attention_p2v's and a third of
// attention_time's launch time.  A head's 48 columns are one whole 128-byte SH line (32 hi | 32 lo halves) plus half of
// a line it shares with its neighbour head (16 hi halves = 32 bytes, 16 lo halves = 32 bytes).  The job's output goes
// through a wave-private LDS image -- [32 rows][full line 128 B | hi part 32 B | lo part 32 B], pitch 208 B -- and
// leaves as 6 instructions whose lanes (row, 16-byte chunk) cover a row's 192 bytes in order.  Same values, same bits.
constexpr int VT_PITCH = 52;                       // floats per key / token slot of the V transpose images (attention_q64_kernel, attention_time16_kernel)
constexpr int VT_BYTES = 32 * VT_PITCH * 4;        // 6656 B per wave
constexpr int OIMG_PITCH = 208;
constexpr int OIMG_BYTES = 32 * OIMG_PITCH;  // 6656 per wave

// d0 = first of 4 consecutive head dims (multiple of 4) -> byte position of their hi halves in the row image (lo: +64 in
// the full line, +32 in the part)
__device__ __forceinline__ int oimg_pos(const bool odd_head, const int d0) {
  if (!odd_head) return d0 < 32 ? d0 * 2 : 128 + (d0 - 32) * 2;
  return d0 < 16 ? 128 + d0 * 2 : (d0 - 16) * 2;
}

// RowPtr: int row (0..31) -> _Float16* start of that query's SH output row, or nullptr when the slot is padding
template <class RowPtr>
__device__ __forceinline__ void attn_store_sh_head(const f32x16 (&oacc)[2], const float inv, unsigned char* img /* this wave's */,
                                                   const int lane, const int head, RowPtr row_ptr) {
  const int r32 = lane & 31, half = lane >> 5;
  const bool odd = head & 1;
  unsigned char* wr = img + r32 * OIMG_PITCH;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (dt == 1 && q >= 2) continue;  // dims 48..63 do not exist
      const int d0 = dt * 32 + q * 8 + half * 4;
      const f32x4 t = {oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv, oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv};
      f16x4 hi, lo;
      ctk_split4(t, hi, lo);
      const int pos = oimg_pos(odd, d0);
      *reinterpret_cast<f16x4*>(wr + pos) = hi;
      *reinterpret_cast<f16x4*>(wr + pos + (pos < 128 ? 64 : 32)) = lo;
    }
  // global byte offsets inside the row: the head's whole line, and the shared line's half
  const int line_full = odd ? (3 * head + 1) / 2 : (3 * head) / 2;       // (48 head + 16) / 32  |  48 head / 32
  const int line_part = odd ? (3 * head) / 2 : (3 * head) / 2 + 1;
  const int part_off = odd ? 32 : 0;                                       // odd heads own columns 16..31 of the shared line
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int idx = k * 64 + lane, row = idx / 12, c = idx - row * 12;
    const f16x8 v = *reinterpret_cast<const f16x8*>(img + row * OIMG_PITCH + c * 16);
    _Float16* base = row_ptr(row);
    const int off = c < 8 ? line_full * 128 + c * 16 : line_part * 128 + part_off + (c < 10 ? (c - 8) * 16 : 64 + (c - 10) * 16);
    if (base) *reinterpret_cast<f16x8*>(reinterpret_cast<unsigned char*>(base) + off) = v;
  }
}

// ---- n2 == 64 keys (points <- virtual, virtual self): workgroup = (frame b, head, QT_PER_WG x 128 queries) ----
// K [64][48] and V^T [48][64 keys, permuted] of this (b, head) are split once into LDS; every wave then takes 32
// queries at a time: Q fragment from global (scaled, split in registers), 18 MFMAs for S', in-register softmax,
// 24 MFMAs for O^T, float4 / SH stores (a lane owns 4 consecutive output columns per register quad).
__global__ __launch_bounds__(256) void attention_kv64_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * 64 * KP + 2 * 64 * VP + 4 * OIMG_BYTES / 2];
  _Float16* kimg = lds;                 // [hi|lo][64 keys][KP]
  _Float16* vimg = lds + 2 * 64 * KP;   // [hi|lo][64 dims (48 used)][VP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r32 = lane & 31, half = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  unsigned char* oimg = reinterpret_cast<unsigned char*>(lds + 2 * 64 * KP + 2 * 64 * VP) + wave * OIMG_BYTES;  // SH output image of this wave

  for (int i = tid; i < 64 * (HD / 4); i += 256) {
    const int key = i / (HD / 4), d4 = i - key * (HD / 4);
    const long row = ((long)b * p.kv_bs + (long)key * p.kv_is) * p.kv_ld + head * HD + d4 * 4;
    const f32x4 kk = *reinterpret_cast<const f32x4*>(p.k + row);
    const f32x4 vv = *reinterpret_cast<const f32x4*>(p.v + row);
    f16x4 hi, lo;
    ctk_split4(kk, hi, lo);
    *reinterpret_cast<f16x4*>(kimg + key * KP + d4 * 4) = hi;
    *reinterpret_cast<f16x4*>(kimg + 64 * KP + key * KP + d4 * 4) = lo;
    ctk_split4(vv, hi, lo);
    const int ko = key & 15;
    const int pos = (key & ~15) + 8 * ((ko >> 2) & 1) + 4 * (ko >> 3) + (ko & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      vimg[(d4 * 4 + e) * VP + pos] = hi[e];
      vimg[64 * VP + (d4 * 4 + e) * VP + pos] = lo[e];
    }
  }

  // raw Q rows of a 32-query tile (6 float4 per lane); the next tile's are requested before this tile's MFMAs
  f32x4 qraw[6];
  auto load_q = [&](int q0) {
    const int qi = min(q0 + r32, p.n1 - 1);
    const float* qp = p.q + ((long)b * p.q_bs + (long)qi * p.q_is) * p.q_ld + head * HD + half * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      qraw[2 * j] = *reinterpret_cast<const f32x4*>(qp + 16 * j);
      qraw[2 * j + 1] = *reinterpret_cast<const f32x4*>(qp + 16 * j + 4);
    }
  };
  const int q_first = blockIdx.x * QT_PER_WG * 128 + wave * 32;
  if (q_first < p.n1) load_q(q_first);
  __syncthreads();

  for (int qt = 0; qt < QT_PER_WG; ++qt) {
    const int q0 = q_first + qt * 128;
    if (q0 >= p.n1) break;  // wave-uniform
    const int qi = min(q0 + r32, p.n1 - 1);
    f16x8 qh[3], ql[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) ctk_split8(qraw[2 * j] * p.scale2, qraw[2 * j + 1] * p.scale2, qh[j], ql[j]);
    if (qt + 1 < QT_PER_WG && q0 + 128 < p.n1) load_q(q0 + 128);
    // S'[key][query]
    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const _Float16* ka = kimg + (kt * 32 + r32) * KP + j * 16 + half * 8;
        const f16x8 kh = *reinterpret_cast<const f16x8*>(ka);
        const f16x8 kl = *reinterpret_cast<const f16x8*>(ka + 64 * KP);
        sacc[kt] = ctk_mma3(kh, kl, qh[j], ql[j], sacc[kt]);
      }
    }
    if (p.kmask || p.qmask) {  // CoTracker2 masks: masked keys / every key of a masked query -> the same huge negative logit
      const bool qm = p.qmask && !p.qmask[qi];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = 32 * kt + 8 * (e >> 2) + 4 * half + (e & 3);
          if (qm || (p.kmask && !p.kmask[key])) sacc[kt][e] = NEG_MAX;
        }
    }
    // softmax over the 64 keys of my query: 32 values here, 32 in lane ^ 32
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(sacc[kt][e] - mx);
        sum += pv;
        sacc[kt][e] = pv * PSCALE;
      }
    sum += __shfl_xor(sum, 32, 64);
    // O^T[dim][query] = V^T . P^T
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kt = s >> 1, base = 8 * (s & 1);
      const f32x4 a = {sacc[kt][base], sacc[kt][base + 1], sacc[kt][base + 2], sacc[kt][base + 3]};
      const f32x4 c = {sacc[kt][base + 4], sacc[kt][base + 5], sacc[kt][base + 6], sacc[kt][base + 7]};
      f16x8 ph, pl;
      ctk_split8(a, c, ph, pl);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const _Float16* va = vimg + (dt * 32 + r32) * VP + s * 16 + half * 8;
        const f16x8 vh = *reinterpret_cast<const f16x8*>(va);
        const f16x8 vl = *reinterpret_cast<const f16x8*>(va + 64 * VP);
        oacc[dt] = ctk_mma3(vh, vl, ph, pl, oacc[dt]);
      }
    }
    const float inv = 1.0f / (sum * PSCALE);
    if (p.o_split) {
      attn_store_sh_head(oacc, inv, oimg, lane, head, [&](int row) -> _Float16* {
        const int qr = q0 + row;
        return qr < p.n1 ? reinterpret_cast<_Float16*>(p.out) + ((long)b * p.o_bs + (long)qr * p.o_is) * p.o_ld : nullptr;
      });
    } else if (q0 + r32 < p.n1) {
      const long orow = ((long)b * p.o_bs + (long)qi * p.o_is) * p.o_ld;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int d = dt * 32 + q * 8 + half * 4;
          if (d < HD) {
            const f32x4 t = {oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv, oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv};
            *reinterpret_cast<f32x4*>(p.out + orow + head * HD + d) = t;
          }
        }
    }
  }
}

// ---- n1 == 64 queries (virtual <- points): workgroup = (key split, head, frame b), 4 waves --------------------
// The 64 pre-scaled queries live in registers as the B operand for the whole key loop; wave w walks the split's
// 32-key tiles w, w+4, ...: K fragment and the (permuted) V^T fragment come straight from global memory (each
// element is used by exactly one lane; the next tile's loads are issued before this tile's MFMAs), online
// softmax per query with the running max shared by the lane pair, then the four waves' (m, l, acc) states are
// merged through LDS and written either as the final rows (one split) or as a partial for attention_merge_kernel.
__global__ __launch_bounds__(256) void attention_q64_kernel(AttnP p) {
  // (16-byte aligned: the per-wave slices double as the V-transpose image and are written with f32x4 stores through `vt`)
  __shared__ __attribute__((aligned(16))) float red[4][64][HD + 2];
  static_assert(VT_BYTES <= (int)sizeof(float) * 64 * (HD + 2), "the V transpose image fits a wave's slice of red");
  static_assert((VT_PITCH * 4) % 16 == 0 && (64 * (HD + 2) * 4) % 16 == 0, "16-byte rows / slices for ds_write_b128");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r32 = lane & 31, half = lane >> 5;
  const int split = blockIdx.x, head = blockIdx.y, b = blockIdx.z;

  f16x8 qh[2][3], ql[2][3];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float* qp = p.q + ((long)b * p.q_bs + (long)(qt * 32 + r32) * p.q_is) * p.q_ld + head * HD + half * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * j) * p.scale2;
      const f32x4 c = *reinterpret_cast<const f32x4*>(qp + 16 * j + 4) * p.scale2;
      ctk_split8(a, c, qh[qt][j], ql[qt][j]);
    }
  }
  const int kbeg = split * p.keys_per_split;
  const int kend = min(p.n2, kbeg + p.keys_per_split);
  const int ntiles = (kend - kbeg + 31) >> 5;
  const float* kbase = p.k + (long)b * p.kv_bs * p.kv_ld + head * HD + half * 8;
  const float* vbase = p.v + (long)b * p.kv_bs * p.kv_ld + head * HD;
  const long kstride = p.kv_is * p.kv_ld;
  const int vd0 = r32, vd1 = min(32 + r32, HD - 1);  // dims of my V^T rows (rows 48..63 of the 2nd tile: unused outputs)

  // Round 5: the V rows of a tile travel like its K rows (6 row-contiguous 16-byte loads per lane instead of 32 scalar loads) and
  // are transposed into the V^T operand order through a wave-private LDS image that borrows this wave's slice of `red` (written
  // only after the key loop): 12 memory instructions per 32-key tile instead of 38.  Same values, same bits.
  f32x4 kraw[6], vrow[6];
  float* vt = &red[wave][0][0];  // [32 keys][VT_PITCH floats] = 6656 B of this wave's 12800
  auto load_tile = [&](int tile) {
    const int k0 = kbeg + tile * 32;
    const long row = (long)min(k0 + r32, kend - 1) * kstride;
    const float* kp = kbase + row;
    const float* vp = vbase + row + half * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      kraw[2 * j] = *reinterpret_cast<const f32x4*>(kp + 16 * j);
      kraw[2 * j + 1] = *reinterpret_cast<const f32x4*>(kp + 16 * j + 4);
      vrow[2 * j] = *reinterpret_cast<const f32x4*>(vp + 16 * j);
      vrow[2 * j + 1] = *reinterpret_cast<const f32x4*>(vp + 16 * j + 4);
    }
  };

  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.0f, 0.0f};
  f32x16 oacc[2][2];  // [dim tile][query tile]
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][qt][e] = 0.0f;

  if (wave < ntiles) load_tile(wave);
  for (int tile = wave; tile < ntiles; tile += 4) {
    const int k0 = kbeg + tile * 32;
    f16x8 kh[3], kl[3], vh[2][2], vl[2][2];  // V: [dim tile][k-step]
#pragma unroll
    for (int j = 0; j < 3; ++j) ctk_split8(kraw[2 * j], kraw[2 * j + 1], kh[j], kl[j]);
    {  // V rows -> LDS image -> V^T fragments: element (k-step s, e) of lane (dim, half) = V[key 16 s + 8 (e >> 2) + 4 half + (e & 3)][dim]
      float* wrow = vt + r32 * VT_PITCH + half * 8;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        *reinterpret_cast<f32x4*>(wrow + 16 * j) = vrow[2 * j];
        *reinterpret_cast<f32x4*>(wrow + 16 * j + 4) = vrow[2 * j + 1];
      }
      __builtin_amdgcn_wave_barrier();  // (one wave, in-order LDS: the reads below see every lane's writes)
      float vraw[2][16];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int key = 16 * s + 8 * (e >> 2) + 4 * half + (e & 3);
          vraw[0][s * 8 + e] = vt[key * VT_PITCH + vd0];
          vraw[1][s * 8 + e] = vt[key * VT_PITCH + vd1];
        }
      __builtin_amdgcn_wave_barrier();  // (the next tile's writes stay behind these reads)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f32x4 a = {vraw[dt][s * 8], vraw[dt][s * 8 + 1], vraw[dt][s * 8 + 2], vraw[dt][s * 8 + 3]};
          const f32x4 c = {vraw[dt][s * 8 + 4], vraw[dt][s * 8 + 5], vraw[dt][s * 8 + 6], vraw[dt][s * 8 + 7]};
          ctk_split8(a, c, vh[dt][s], vl[dt][s]);
        }
    }
    if (tile + 4 < ntiles) load_tile(tile + 4);  // raw registers are free again: prefetch behind the MFMAs

    f32x16 sacc[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[qt][e] = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) sacc[qt] = ctk_mma3(kh[j], kl[j], qh[qt][j], ql[qt][j], sacc[qt]);
    }
    if (p.kmask || p.qmask) {  // CoTracker2 masks (finite bias: a fully masked row stays a uniform softmax)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = min(k0 + 8 * (e >> 2) + 4 * half + (e & 3), kend - 1);
        const bool km = p.kmask && !p.kmask[key];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          if (km || (p.qmask && !p.qmask[qt * 32 + r32])) sacc[qt][e] = NEG_MAX;
      }
    }
    if (k0 + 32 > kend) {  // ragged last tile: keys past the split's range get probability 0
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k0 + 8 * (e >> 2) + 4 * half + (e & 3);
        if (key >= kend) {
          sacc[0][e] = -INFINITY;
          sacc[1][e] = -INFINITY;
        }
      }
    }
    f16x8 ph[2][2], pl[2][2];  // [query tile][k-step]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float tmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[qt][e]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mnew = fmaxf(m[qt], tmax);       // finite: every tile holds at least one valid key
      const float alpha = __builtin_amdgcn_exp2f(m[qt] - mnew);      // m = -inf on the first tile -> 0
      m[qt] = mnew;
      float psum = 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(sacc[qt][e] - mnew);
        psum += pv;
        sacc[qt][e] = pv * PSCALE;
      }
      l[qt] = l[qt] * alpha + psum;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        oacc[0][qt][e] *= alpha;
        oacc[1][qt][e] *= alpha;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 a = {sacc[qt][8 * s], sacc[qt][8 * s + 1], sacc[qt][8 * s + 2], sacc[qt][8 * s + 3]};
        const f32x4 c = {sacc[qt][8 * s + 4], sacc[qt][8 * s + 5], sacc[qt][8 * s + 6], sacc[qt][8 * s + 7]};
        ctk_split8(a, c, ph[qt][s], pl[qt][s]);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = ctk_mma3(vh[dt][s], vl[dt][s], ph[qt][s], pl[qt][s], oacc[dt][qt]);
  }

  // merge the four waves' states
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float lsum = l[qt] + __shfl_xor(l[qt], 32, 64);
    float* rr = &red[wave][qt * 32 + r32][0];
    if (half == 0) {
      rr[0] = m[qt];
      rr[1] = lsum;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + q * 8 + half * 4;
        if (d < HD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) rr[2 + d + e] = oacc[dt][qt][4 * q + e] * (1.0f / PSCALE);
        }
      }
  }
  __syncthreads();
  {
    const int qi = tid >> 2, part = tid & 3;  // 64 queries x 4 groups of 12 dims
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red[w][qi][0]);
    float L = 0.0f, o[12];
#pragma unroll
    for (int d = 0; d < 12; ++d) o[d] = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float wm = red[w][qi][0];
      const float wgt = (wm == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(wm - mm);  // waves that saw no tile carry m = -inf, l = 0
      L += wgt * red[w][qi][1];
#pragma unroll
      for (int d = 0; d < 12; ++d) o[d] += wgt * red[w][qi][2 + part * 12 + d];
    }
    if (p.splits == 1) {
      const float inv = 1.0f / L;
      const long orow = ((long)b * p.o_bs + (long)qi * p.o_is) * p.o_ld;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x4 t = {o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv};
        const int col = head * HD + part * 12 + 4 * i;
        if (p.o_split) {
          f16x4 hi, lo;
          ctk_split4(t, hi, lo);
          _Float16* dst = reinterpret_cast<_Float16*>(p.out) + orow + ctk_sh_col(col);
          *reinterpret_cast<f16x4*>(dst) = hi;
          *reinterpret_cast<f16x4*>(dst + 32) = lo;
        } else {
          *reinterpret_cast<f32x4*>(p.out + orow + col) = t;
        }
      }
    } else {
      float* pp = p.partial + ((((long)split * p.nbatch + b) * CTK_HEADS + head) * p.n1 + qi) * (HD + 2);
      if (part == 0) {
        pp[0] = mm;
        pp[1] = L;
      }
#pragma unroll
      for (int d = 0; d < 12; ++d) pp[2 + part * 12 + d] = o[d];
    }
  }
}

// ---- n1 == n2 (time attention: S = 16 sliding / streaming, S = T offline) -------------------------------------
// One WAVE per (batch group, head, 32-query tile); no LDS, no barrier, so occupancy hides the load latency.
// S <= 16: two batches share one 32 x 32 score tile (block diagonal: slots 0..15 = batch 2g, 16..31 = batch
// 2g+1, cross-batch scores masked to -inf so their probabilities are exactly 0); S > 16: one batch, 32-key tiles
// with online softmax.  K and V^T fragments come straight from global memory as in attention_q64_kernel.
template <int PACK>  // batches per score tile: 2 when n1 <= 16, else 1
__global__ __launch_bounds__(256, 2) void attention_self_kernel(AttnP p) {
  __shared__ __attribute__((aligned(16))) unsigned char oimg_all[4 * OIMG_BYTES];  // SH output images (attn_store_sh_head), one per wave
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r32 = lane & 31, half = lane >> 5;
  unsigned char* oimg = oimg_all + wave * OIMG_BYTES;
  const int head = blockIdx.y;
  constexpr int pack = PACK;
  constexpr int spb = 32 / pack;     // tile slots per batch
  const long job = (long)blockIdx.x * 4 + wave;
  const long njobs = (long)((p.nbatch + pack - 1) / pack) * p.qtiles;
  if (job >= njobs) return;          // wave-uniform
  const int g = (int)(job / p.qtiles), qtile = (int)(job - (long)g * p.qtiles);
  const int b0 = g * pack;

  const int qb = r32 / spb, qi = qtile * spb + r32 % spb;
  const bool qvalid = (b0 + qb < p.nbatch) && (qi < p.n1);
  const int qbc = min(b0 + qb, p.nbatch - 1), qic = min(qi, p.n1 - 1);
  f16x8 qh[3], ql[3];
  {
    const float* qp = p.q + ((long)qbc * p.q_bs + (long)qic * p.q_is) * p.q_ld + head * HD + half * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * j) * p.scale2;
      const f32x4 c = *reinterpret_cast<const f32x4*>(qp + 16 * j + 4) * p.scale2;
      ctk_split8(a, c, qh[j], ql[j]);
    }
  }
  const int vd0 = r32, vd1 = min(32 + r32, HD - 1);
  float m = -INFINITY, l = 0.0f;
  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.0f;

  const int nkt = (pack == 2) ? 1 : (p.n2 + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    f16x8 kh[3], kl[3], vh[2][2], vl[2][2];
    {
      const int kb = min(b0 + r32 / spb, p.nbatch - 1), ki = min(kt * spb + r32 % spb, p.n2 - 1);
      const float* kp = p.k + ((long)kb * p.kv_bs + (long)ki * p.kv_is) * p.kv_ld + head * HD + half * 8;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        ctk_split8(*reinterpret_cast<const f32x4*>(kp + 16 * j), *reinterpret_cast<const f32x4*>(kp + 16 * j + 4), kh[j], kl[j]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v0[8], v1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int slot = 16 * s + 8 * (e >> 2) + 4 * half + (e & 3);
        const int kb = min(b0 + slot / spb, p.nbatch - 1), ki = min(kt * spb + slot % spb, p.n2 - 1);
        const float* vp = p.v + ((long)kb * p.kv_bs + (long)ki * p.kv_is) * p.kv_ld + head * HD;
        v0[e] = vp[vd0];
        v1[e] = vp[vd1];
      }
      ctk_split8(f32x4{v0[0], v0[1], v0[2], v0[3]}, f32x4{v0[4], v0[5], v0[6], v0[7]}, vh[0][s], vl[0][s]);
      ctk_split8(f32x4{v1[0], v1[1], v1[2], v1[3]}, f32x4{v1[4], v1[5], v1[6], v1[7]}, vh[1][s], vl[1][s]);
    }
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) sacc = ctk_mma3(kh[j], kl[j], qh[j], ql[j], sacc);
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int slot = 8 * (e >> 2) + 4 * half + (e & 3);
      const bool ok = (slot / spb == qb) && (b0 + slot / spb < p.nbatch) && (kt * spb + slot % spb < p.n2);
      if (ok && ((p.kmask && !p.kmask[min(kt * spb + slot % spb, p.n2 - 1)]) || (p.qmask && !p.qmask[qic]))) sacc[e] = NEG_MAX;
      sacc[e] = ok ? sacc[e] : -INFINITY;
      tmax = fmaxf(tmax, sacc[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    float mnew = fmaxf(m, tmax);
    if (mnew == -INFINITY) mnew = 0.0f;  // a query slot with no valid key (padding slot): all probabilities 0, never stored
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    m = mnew;
    float psum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(sacc[e] - mnew);
      psum += pv;
      sacc[e] = pv * PSCALE;
    }
    l = l * alpha + psum;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      oacc[0][e] *= alpha;
      oacc[1][e] *= alpha;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 ph, pl;
      ctk_split8(f32x4{sacc[8 * s], sacc[8 * s + 1], sacc[8 * s + 2], sacc[8 * s + 3]},
                 f32x4{sacc[8 * s + 4], sacc[8 * s + 5], sacc[8 * s + 6], sacc[8 * s + 7]}, ph, pl);
      oacc[0] = ctk_mma3(vh[0][s], vl[0][s], ph, pl, oacc[0]);
      oacc[1] = ctk_mma3(vh[1][s], vl[1][s], ph, pl, oacc[1]);
    }
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / (l * PSCALE);
  if (p.o_split) {
    attn_store_sh_head(oacc, inv, oimg, lane, head, [&](int row) -> _Float16* {
      const int ob = b0 + row / spb, oi = qtile * spb + row % spb;
      return (ob < p.nbatch && oi < p.n1) ? reinterpret_cast<_Float16*>(p.out) + ((long)ob * p.o_bs + (long)oi * p.o_is) * p.o_ld : nullptr;
    });
  } else if (qvalid) {
    const long orow = ((long)qbc * p.o_bs + (long)qic * p.o_is) * p.o_ld;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int d = dt * 32 + q * 8 + half * 4;
        if (d < HD) {
          const f32x4 t = {oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv, oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv};
          *reinterpret_cast<f32x4*>(p.out + orow + head * HD + d) = t;
        }
      }
  }
}

// ---- time attention, S <= 16, no masks (CoTracker3 sliding / streaming): persistent waves with register prefetch ---
// attention_self_kernel<2> is latency bound: at 2 waves per SIMD a wave loads its q / k / v rows (18 KB), waits, splits,
// runs 63 MFMAs and exits -- 204 us per launch at C3 where the MFMAs alone would take ~25 us.  Here a wave walks over
// jobs (batch pair, head) with a stride of the whole grid and issues the NEXT job's raw f32 loads before it splits and
// multiplies the current one, so the memory latency of job i+1 hides behind the arithmetic of job i.  Jobs are numbered
// head-fastest: neighbouring waves read neighbouring 192-byte head slices of the same token rows.  The arithmetic (split,
// MFMA order, log2-domain softmax) is that of attention_self_kernel<2>, instruction for instruction: same bits out.
// Round 5: the V rows travel like the K rows -- 6 row-contiguous 16-byte loads per lane -- and are transposed into the V^T
// operand order through a wave-private LDS image ([32 token slots][52 floats]: conflict-free 16-byte writes, the 32 column
// reads of a lane hit 32 consecutive banks).  Until round 4 every lane fetched its 32 V^T elements with 32 scalar
// global_load_dword: 44 memory instructions per job, and the request stream, not the DRAM pins, bounded the kernel
// (profiles/r04_sq_counters.txt: 65 % of a wave's life waiting for operands at 4.0 TB/s).  Now 18.  Same values, same bits.
struct TimeRaw {
  f32x4 q[6], k[6], v[6];
};

__global__ __launch_bounds__(256, 2) void attention_time16_kernel(AttnP p, long njobs) {
  __shared__ __attribute__((aligned(16))) unsigned char oimg_all[4 * (OIMG_BYTES + VT_BYTES)];  // per wave: SH output image (attn_store_sh_head) | V transpose image
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r32 = lane & 31, half = lane >> 5;
  unsigned char* oimg = oimg_all + wave * (OIMG_BYTES + VT_BYTES);
  float* vt = reinterpret_cast<float*>(oimg + OIMG_BYTES);
  const long stride = (long)gridDim.x * 4;
  long job = (long)blockIdx.x * 4 + wave;
  if (job >= njobs) return;  // wave-uniform
  const int vd0 = r32, vd1 = min(32 + r32, HD - 1);
  const int rb = r32 >> 4, ri = r32 & 15;  // batch of the pair / frame slot of my q and k row

  // Round 5, second step: ROW-CONTIGUOUS loads.  A head's slice of a token row is 192 contiguous bytes = 12 sixteen-byte chunks;
  // load i of lane l fetches chunk (64 i + l) % 12 of token slot (64 i + l) / 12, so the 64 lanes of an instruction walk along
  // 5.3 rows instead of touching 32 rows x 16 bytes each: ~11 cache lines per instruction instead of 32 (every line of a slice
  // was requested by up to three instructions).  The chunks go through the wave's LDS image ([32 slots][52 floats], one matrix at
  // a time) and come back in the MFMA operand order; q, k and v all take this path.  Same values, same bits.
  // (token slot, chunk) of my load i -- recomputed where needed: six more live registers spill
  // (hipcc hoists the 6 LDS offsets and the 12 lane-dependent address parts out of the job loop and spills them -- a scratch
  // reload inside the loop waits for the prefetched loads of the next job as well; `ln` is the lane id laundered through an
  // empty asm inside the loop, so the arithmetic stays where it is used: ~12 VALU per load)
  auto slot_of = [&](const int i, const int ln) { return ((64 * i + ln) * 43691) >> 19; };  // idx / 12, exact for idx < 384
  auto chunk_of = [&](const int i, const int ln) { return (64 * i + ln) - 12 * slot_of(i, ln); };
  auto load = [&](long jb, TimeRaw& r) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int head = (int)(jb % CTK_HEADS);
    const int b0 = (int)(jb / CTK_HEADS) * 2;
    const float* qb = p.q + head * HD;
    const float* kb = p.k + head * HD;
    const float* vb = p.v + head * HD;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = slot_of(i, ln), c = chunk_of(i, ln);
      const int bc = min(b0 + (row >> 4), p.nbatch - 1), fr = row & 15;
      const long qo = ((long)bc * p.q_bs + (long)min(fr, p.n1 - 1) * p.q_is) * p.q_ld + c * 4;
      const long ko = ((long)bc * p.kv_bs + (long)min(fr, p.n2 - 1) * p.kv_is) * p.kv_ld + c * 4;
      r.q[i] = *reinterpret_cast<const f32x4*>(qb + qo);
      r.k[i] = *reinterpret_cast<const f32x4*>(kb + ko);
      r.v[i] = *reinterpret_cast<const f32x4*>(vb + ko);
      __builtin_amdgcn_sched_barrier(0);  // (one load's address arithmetic at a time: all 18 at once spill)
    }
  };
  // raw chunks of one matrix -> the wave's LDS image (token slot major, 52 floats per slot)
  auto stage = [&](const f32x4 (&m)[6]) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(vt + slot_of(i, ln) * VT_PITCH + chunk_of(i, ln) * 4) = m[i];
    __builtin_amdgcn_wave_barrier();  // (one wave, in-order LDS: the reads that follow see every lane's writes)
  };
  // image -> q / k operand rows: lane (slot r32, half) takes columns half*8 + 16 j + 0..7
  auto frag_rows = [&](f32x4 (&f)[6]) {
    const float* rr = vt + r32 * VT_PITCH + half * 8;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f[2 * j] = *reinterpret_cast<const f32x4*>(rr + 16 * j);
      f[2 * j + 1] = *reinterpret_cast<const f32x4*>(rr + 16 * j + 4);
    }
    __builtin_amdgcn_wave_barrier();  // (the next matrix' writes stay behind these reads)
  };

  TimeRaw raw;
  load(job, raw);
  while (true) {
    const int head = (int)(job % CTK_HEADS);
    const int b0 = (int)(job / CTK_HEADS) * 2;
    // raw f32 -> split-half fragments (the raw registers are dead afterwards and take the next job's loads)
    f16x8 qh[3], ql[3], kh[3], kl[3], vh[2][2], vl[2][2];
    {
      f32x4 f[6];
      stage(raw.q);
      frag_rows(f);
#pragma unroll
      for (int j = 0; j < 3; ++j) ctk_split8(f[2 * j] * p.scale2, f[2 * j + 1] * p.scale2, qh[j], ql[j]);
      stage(raw.k);
      frag_rows(f);
#pragma unroll
      for (int j = 0; j < 3; ++j) ctk_split8(f[2 * j], f[2 * j + 1], kh[j], kl[j]);
    }
    {  // V image -> V^T fragments: element (k-step s, e) of lane (dim, half) = V[slot 16 s + 8 (e >> 2) + 4 half + (e & 3)][dim]
      stage(raw.v);
      float v0[16], v1[16];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int slot = 16 * s + 8 * (e >> 2) + 4 * half + (e & 3);
          v0[8 * s + e] = vt[slot * VT_PITCH + vd0];
          v1[8 * s + e] = vt[slot * VT_PITCH + vd1];
        }
      __builtin_amdgcn_wave_barrier();  // (the next job's writes stay behind these reads)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        ctk_split8(f32x4{v0[8 * s], v0[8 * s + 1], v0[8 * s + 2], v0[8 * s + 3]}, f32x4{v0[8 * s + 4], v0[8 * s + 5], v0[8 * s + 6], v0[8 * s + 7]},
                   vh[0][s], vl[0][s]);
        ctk_split8(f32x4{v1[8 * s], v1[8 * s + 1], v1[8 * s + 2], v1[8 * s + 3]}, f32x4{v1[8 * s + 4], v1[8 * s + 5], v1[8 * s + 6], v1[8 * s + 7]},
                   vh[1][s], vl[1][s]);
      }
    }
    const long next = job + stride;
    const bool more = next < njobs;  // wave-uniform
    if (more) load(next, raw);

    const bool qvalid = (b0 + rb < p.nbatch) && (ri < p.n1);
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) sacc = ctk_mma3(kh[j], kl[j], qh[j], ql[j], sacc);
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int slot = 8 * (e >> 2) + 4 * half + (e & 3);
      const bool ok = ((slot >> 4) == rb) && (b0 + (slot >> 4) < p.nbatch) && ((slot & 15) < p.n2);
      sacc[e] = ok ? sacc[e] : -INFINITY;
      tmax = fmaxf(tmax, sacc[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = (tmax == -INFINITY) ? 0.0f : tmax;  // padding slot: all probabilities 0, never stored
    float l = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(sacc[e] - mnew);
      l += pv;
      sacc[e] = pv * PSCALE;
    }
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 ph, pl;
      ctk_split8(f32x4{sacc[8 * s], sacc[8 * s + 1], sacc[8 * s + 2], sacc[8 * s + 3]},
                 f32x4{sacc[8 * s + 4], sacc[8 * s + 5], sacc[8 * s + 6], sacc[8 * s + 7]}, ph, pl);
      oacc[0] = ctk_mma3(vh[0][s], vl[0][s], ph, pl, oacc[0]);
      oacc[1] = ctk_mma3(vh[1][s], vl[1][s], ph, pl, oacc[1]);
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / (l * PSCALE);
    if (p.o_split) {
      attn_store_sh_head(oacc, inv, oimg, lane, head, [&](int row) -> _Float16* {
        const int ob = b0 + (row >> 4), oi = row & 15;
        return (ob < p.nbatch && oi < p.n1) ? reinterpret_cast<_Float16*>(p.out) + ((long)ob * p.o_bs + (long)oi * p.o_is) * p.o_ld : nullptr;
      });
    } else if (qvalid) {
      const long orow = ((long)(b0 + rb) * p.o_bs + (long)ri * p.o_is) * p.o_ld;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int d = dt * 32 + q * 8 + half * 4;
          if (d < HD) {
            const f32x4 t = {oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv, oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv};
            *reinterpret_cast<f32x4*>(p.out + orow + head * HD + d) = t;
          }
        }
    }
    if (!more) break;
    job = next;
  }
}

__global__ void attention_merge_kernel(AttnP p) {
  // one thread per (batch, head, query, dim)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.nbatch * CTK_HEADS * p.n1 * HD;
  if (i >= total) return;
  const int d = i % HD;
  long r = i / HD;
  const int qi = r % p.n1; r /= p.n1;
  const int head = r % CTK_HEADS;
  const int b = r / CTK_HEADS;
  const long stride = (long)p.nbatch * CTK_HEADS * p.n1 * (HD + 2);
  const float* base = p.partial + (((long)b * CTK_HEADS + head) * p.n1 + qi) * (HD + 2);
  float m = -INFINITY;
  for (int s = 0; s < p.splits; ++s) m = fmaxf(m, base[s * stride]);
  float l = 0.0f, a = 0.0f;
  for (int s = 0; s < p.splits; ++s) {
    const float w = p.log2m ? __builtin_amdgcn_exp2f(base[s * stride] - m) : expf(base[s * stride] - m);
    l += w * base[s * stride + 1];
    a += w * base[s * stride + 2 + d];
  }
  const float o = a / l;
  if (p.o_split) {
    _Float16* oh = reinterpret_cast<_Float16*>(p.out) + (b * p.o_bs + qi * p.o_is) * p.o_ld + ctk_sh_col(head * HD + d);
    const _Float16 hi = (_Float16)o;
    oh[0] = hi;
    oh[32] = (_Float16)(o - (float)hi);
  } else {
    p.out[(b * p.o_bs + qi * p.o_is) * p.o_ld + head * HD + d] = o;
  }
}

// CTK_OPT_ATTENTION_VALU (include/ctk.h): 0 = MFMA kernels for the 64-key, 64-query and square shapes | 1: the VALU kernel everywhere.
int attn_backend() { return ctk_opt(CTK_OPT_ATTENTION_VALU); }

}  // namespace

extern "C" int ctk_attention(const ctk_attn_args* a, void* stream) {
  if (!a || !a->q || !a->k || !a->v || !a->out) return CTK_E_NULL;
  if (a->nbatch <= 0 || a->n1 <= 0 || a->n2 <= 0) return CTK_E_SHAPE;
  if ((a->q_ld % 4) || (a->kv_ld % 4) || (a->o_ld % 4)) return CTK_E_ALIGN;
  if (!ctk_aligned16(a->q) || !ctk_aligned16(a->k) || !ctk_aligned16(a->v) || !ctk_aligned16(a->out)) return CTK_E_ALIGN;
  AttnP p;
  p.q = a->q; p.q_ld = a->q_ld; p.q_bs = a->q_bs; p.q_is = a->q_is;
  p.k = a->k; p.v = a->v; p.kv_ld = a->kv_ld; p.kv_bs = a->kv_bs; p.kv_is = a->kv_is;
  p.out = static_cast<float*>(a->out); p.o_ld = a->o_ld; p.o_bs = a->o_bs; p.o_is = a->o_is;
  p.o_split = a->o_split;
  if (p.o_split && (a->o_ld % 64)) return CTK_E_ALIGN;
  p.nbatch = a->nbatch; p.n1 = a->n1; p.n2 = a->n2;
  p.splits = a->splits > 1 ? a->splits : 1;
  if (p.splits > 1 && !a->partial) return CTK_E_NULL;
  p.partial = a->partial;
  p.kmask = a->key_mask;
  p.qmask = a->query_mask;
  p.scale = 0.14433756729740643f;  // 48 ** -0.5 (blocks.py:372)
  p.scale2 = 0.14433756729740643f * 1.4426950408889634f;
  p.log2m = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double flops = 4.0 * a->nbatch * (double)a->n1 * a->n2 * CTK_HID;
  const double bytes = 4.0 * a->nbatch * ((double)a->n1 * 2 + (double)a->n2 * 2) * CTK_HID;
  const bool mfma = attn_backend() == 0;  // CTK_ATTN=1 forces the VALU kernel everywhere (A/B knob)

  if (mfma && a->n2 == CTK_VIRT) {
    // points <- virtual / virtual self: all 64 keys in one pass, no key split
    p.splits = 1;
    p.keys_per_split = a->n2;
    p.bpw = 1;
    p.qtiles = 1;
    CtkProfScope ps(a->n1 > CTK_VIRT ? "attention_p2v" : "attention_vself", flops, bytes, s);
    const unsigned gx = (unsigned)((a->n1 + 128 * QT_PER_WG - 1) / (128 * QT_PER_WG));
    hipLaunchKernelGGL(attention_kv64_kernel, dim3(gx, CTK_HEADS, (unsigned)a->nbatch), dim3(256), 0, s, p);
    CTK_HIP_CHECK_LAUNCH();
    return CTK_OK;
  }
  if (mfma && a->n1 == CTK_VIRT && a->n2 > CTK_VIRT) {
    // virtual <- points: key range split over workgroups, 32-key tiles
    p.keys_per_split = ((a->n2 + p.splits - 1) / p.splits + 31) / 32 * 32;
    p.splits = (a->n2 + p.keys_per_split - 1) / p.keys_per_split;  // drop empty splits
    p.bpw = 1;
    p.qtiles = 1;
    p.log2m = 1;
    CtkProfScope ps("attention_v2p", flops, bytes, s);
    hipLaunchKernelGGL(attention_q64_kernel, dim3((unsigned)p.splits, CTK_HEADS, (unsigned)a->nbatch), dim3(256), 0, s, p);
    CTK_HIP_CHECK_LAUNCH();
    if (p.splits > 1) {
      const long total = (long)p.nbatch * CTK_HEADS * p.n1 * HD;
      hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
      CTK_HIP_CHECK_LAUNCH();
    }
    return CTK_OK;
  }

  if (mfma && a->n1 == a->n2 && p.splits == 1) {
    // time attention (and any other square shape): one wave per (batch pair | batch, head, 32-query tile)
    p.keys_per_split = a->n2;
    p.bpw = a->n1 <= 16 ? 2 : 1;
    p.qtiles = p.bpw == 2 ? 1 : (a->n1 + 31) / 32;
    const long njobs = (long)((a->nbatch + p.bpw - 1) / p.bpw) * p.qtiles;
    CtkProfScope ps(a->q_is == 1 ? "attention_time" : "attention_vself", flops, bytes, s);
    const bool persistent = ctk_opt(CTK_OPT_ATTENTION_TIME_PERSISTENT) != 0;  // 0 = the non-persistent kernel (bit-identical)
    if (p.bpw == 2 && !p.kmask && !p.qmask && persistent) {
      // persistent waves: 2 workgroups per CU, each wave walks over ~njobs*8/2048 (batch pair, head) jobs
      const long total = njobs * CTK_HEADS;
      const unsigned blocks = (unsigned)((total + 3) / 4 < 512 ? (total + 3) / 4 : 512);
      hipLaunchKernelGGL(attention_time16_kernel, dim3(blocks), dim3(256), 0, s, p, total);
    } else if (p.bpw == 2) hipLaunchKernelGGL(attention_self_kernel<2>, dim3((unsigned)((njobs + 3) / 4), CTK_HEADS), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attention_self_kernel<1>, dim3((unsigned)((njobs + 3) / 4), CTK_HEADS), dim3(256), 0, s, p);
    CTK_HIP_CHECK_LAUNCH();
    return CTK_OK;
  }

  p.keys_per_split = ((a->n2 + p.splits - 1) / p.splits + KC - 1) / KC * KC;
  p.splits = (a->n2 + p.keys_per_split - 1) / p.keys_per_split;  // drop empty splits
  unsigned gx;
  if (a->n1 >= 64) {
    p.bpw = 1;
    p.qtiles = (a->n1 + 63) / 64;
    gx = (unsigned)(p.qtiles * a->nbatch);
  } else {
    p.bpw = 64 / a->n1;
    if (p.bpw > MAXB) p.bpw = MAXB;
    p.qtiles = 1;
    gx = (unsigned)((a->nbatch + p.bpw - 1) / p.bpw);
  }
  // one recorder row per use of the kernel on the path: time axis / points<-virtual / virtual<-points / virtual self
  const char* pname = p.splits > 1 ? "attention_v2p" : (a->n1 > CTK_VIRT && a->n2 == CTK_VIRT) ? "attention_p2v"
                      : (a->n1 == CTK_VIRT && a->n2 == CTK_VIRT && a->q_is != 1) ? "attention_vself" : "attention_time";
  CtkProfScope ps(pname, flops, bytes, s);
  const size_t lds_bytes = (size_t)2 * p.bpw * (KC * HD + BPAD) * sizeof(float);
  hipLaunchKernelGGL(attention_kernel, dim3(gx, CTK_HEADS, (unsigned)p.splits), dim3(64), lds_bytes, s, p);
  CTK_HIP_CHECK_LAUNCH();
  if (p.splits > 1) {
    const long total = (long)p.nbatch * CTK_HEADS * p.n1 * HD;
    hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
    CTK_HIP_CHECK_LAUNCH();
  }
  return CTK_OK;
}
