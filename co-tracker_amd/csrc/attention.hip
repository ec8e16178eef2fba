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
#include "ctk_profile.h"
#include "gemm_params.h"

namespace {

constexpr int HD = CTK_HEAD_DIM;  // 48
constexpr int KC = 16;            // keys per LDS chunk
constexpr int MAXB = 8;           // max batches packed into one wave (n1 >= 8)
constexpr int BPAD = 4;           // floats between batches in LDS (bank spread)

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
  float scale;
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
    const float w = expf(base[s * stride] - m);
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
  p.keys_per_split = ((a->n2 + p.splits - 1) / p.splits + KC - 1) / KC * KC;
  p.splits = (a->n2 + p.keys_per_split - 1) / p.keys_per_split;  // drop empty splits
  p.scale = 0.14433756729740643f;  // 48 ** -0.5 (blocks.py:372)
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtkProfScope ps("attention", 4.0 * a->nbatch * (double)a->n1 * a->n2 * CTK_HID,
                  4.0 * a->nbatch * ((double)a->n1 * 2 + (double)a->n2 * 2) * CTK_HID, s);
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
