// Row kernels of the CoTracker2 iteration (cotracker.py:86-173) around CorrBlock and the update former:
// token assembly (flow embedding + concat + positional embedding), the state update (coords += delta[:2],
// GroupNorm of the 128 feature deltas), the visibility head and the 4-D (grid_sampler_2d) feature sampler used
// for the positional embedding.  All HBM-bound, one pass over their rows.
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"

namespace {

constexpr int V2_IN = 456;    // input_dim (cotracker.py:47)
constexpr int V2_FLOW = 130;  // get_2d_embedding(flows, 64, cat_coords=True): 2 + 64 + 64
constexpr int V2_CORR = 196;  // 4 levels x 49 taps
constexpr int V2_C = 128;

// x[n*S+t][0..in_ld) = cat(flow_emb, fcorrs, track_feat, track_mask, vis) + pos_emb[n]   (cotracker.py:139-150;
// the time embedding of :150 is folded into the input projection's per-frame bias rows)
__global__ void v2_assemble_kernel(const float* coords, const float* fcorrs, const float* track_feat, const float* track_mask,
                                   const float* vis, const float* pos, int S, int N, int in_ld, float* x, int x_split) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)S * N * in_ld;
  if (i >= total) return;
  const int c = i % in_ld;
  const long row = i / in_ld;  // n*S + t
  const int t = row % S;
  const int n = row / S;
  float v = 0.0f;
  if (c < V2_IN) {
    if (c < V2_FLOW) {
      // flows = coords - coords[:, 0:1]   (cotracker.py:135); get_2d_embedding (embeddings.py:87-120)
      const int axis = c < 2 ? c : ((c - 2) >> 6);
      const float f = __fsub_rn(coords[((long)t * N + n) * 2 + axis], coords[(long)n * 2 + axis]);
      if (c < 2) {
        v = f;
      } else {
        const int k = (c - 2) & 63;                                      // position inside pe_x / pe_y
        const float div = __fmul_rn((float)(k & ~1), 1000.0f / 64.0f);   // arange(0, C, 2) * (1000 / C)
        const float arg = __fmul_rn(f, div);
        v = (k & 1) ? cosf(arg) : sinf(arg);
      }
    } else if (c < V2_FLOW + V2_CORR) {
      v = fcorrs[row * V2_CORR + (c - V2_FLOW)];
    } else if (c < V2_FLOW + V2_CORR + V2_C) {
      v = track_feat[((long)t * N + n) * V2_C + (c - V2_FLOW - V2_CORR)];
    } else if (c == V2_IN - 2) {
      v = track_mask[(long)t * N + n];
    } else {
      v = vis[(long)t * N + n];
    }
    v = __fadd_rn(v, pos[(long)n * V2_IN + c]);
  }
  if (x_split) {
    _Float16* xh = reinterpret_cast<_Float16*>(x) + row * (2L * in_ld) + ctk_sh_col(c);
    const _Float16 hi = (_Float16)v;
    xh[0] = hi;
    xh[32] = (_Float16)(v - (float)hi);
  } else {
    x[row * in_ld + c] = v;
  }
}

// coords[t,n] += delta[n*S+t][0:2]; normed[t*N+n][:] = GroupNorm(1,128)(delta[n*S+t][2:130])  (cotracker.py:157-167)
// one wavefront per row, 2 channels per lane
__global__ __launch_bounds__(256) void v2_apply_delta_kernel(const float* delta, int out_ld, int S, int N, float* coords,
                                                              const float* gamma, const float* beta, float eps, float* normed) {
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // n*S + t
  const int lane = threadIdx.x & 63;
  if (row >= (long)S * N) return;
  const int t = row % S;
  const int n = row / S;
  const float* d = delta + row * out_ld;
  const long tn = (long)t * N + n;
  if (lane < 2) coords[tn * 2 + lane] += d[lane];
  float2 v = make_float2(d[2 + lane * 2], d[3 + lane * 2]);
  const float mean = ctk_wave_sum(v.x + v.y) * (1.0f / V2_C);
  v.x -= mean;
  v.y -= mean;
  const float var = ctk_wave_sum(v.x * v.x + v.y * v.y) * (1.0f / V2_C);
  const float rstd = 1.0f / sqrtf(var + eps);
  float2 o;
  o.x = v.x * rstd * gamma[lane * 2] + beta[lane * 2];
  o.y = v.y * rstd * gamma[lane * 2 + 1] + beta[lane * 2 + 1];
  *reinterpret_cast<float2*>(normed + tn * V2_C + lane * 2) = o;
}

// vis[r] = <track_feat[r], w> + b   (vis_predictor, cotracker.py:81-83,172)
__global__ __launch_bounds__(256) void v2_vis_head_kernel(const float* tf, const float* w, const float* b, long R, float* out) {
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float2 v = *reinterpret_cast<const float2*>(tf + row * V2_C + lane * 2);
  const float2 ww = *reinterpret_cast<const float2*>(w + lane * 2);
  const float s = ctk_wave_sum(v.x * ww.x + v.y * ww.y);
  if (lane == 0) out[row] = s + b[0];
}

// sample_features4d (model_utils.py:258-290) of a channels-last map [H,W,C] at (x, y): bilinear_sampler's 4-D path =
// ATen grid_sampler_2d (align_corners, border): weights nw = s*e, ne = s*w, sw = n*e, se = n*w, accumulated with FMA
__global__ void sample4d_kernel(const float* map, int H, int W, int C, const float* coords, int N, float sx, float sy, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * C) return;
  const int c = i % C;
  const int n = i / C;
  const CtkTap tx = ctk_tap(coords[n * 2], W, sx), ty = ctk_tap(coords[n * 2 + 1], H, sy);
  const float* m = map + c;
  // an out-of-range corner (index clamped, weight 0) contributes exactly 0 in the reference as well
  float o = __fmul_rn(m[((long)ty.i0 * W + tx.i0) * C], __fmul_rn(ty.w0, tx.w0));
  o = __fmaf_rn(m[((long)ty.i0 * W + tx.i1) * C], __fmul_rn(ty.w0, tx.w1), o);
  o = __fmaf_rn(m[((long)ty.i1 * W + tx.i0) * C], __fmul_rn(ty.w1, tx.w0), o);
  o = __fmaf_rn(m[((long)ty.i1 * W + tx.i1) * C], __fmul_rn(ty.w1, tx.w1), o);
  out[i] = o;
}

}  // namespace

extern "C" int ctk_v2_assemble(int32_t S, int32_t N, const float* coords, const float* fcorrs, const float* track_feat,
                               const float* track_mask, const float* vis, const float* pos, int32_t in_ld, void* x,
                               int32_t x_split, void* stream) {
  if (!coords || !fcorrs || !track_feat || !track_mask || !vis || !pos || !x) return CTK_E_NULL;
  if (S <= 0 || N <= 0 || in_ld < V2_IN || (in_ld % 32)) return CTK_E_SHAPE;
  const long total = (long)S * N * in_ld;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtkProfScope ps("v2_assemble", 0.0, 8.0 * total, s);
  hipLaunchKernelGGL(v2_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, coords, fcorrs, track_feat, track_mask,
                     vis, pos, S, N, in_ld, static_cast<float*>(x), x_split);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_v2_apply_delta(int32_t S, int32_t N, const float* delta, int32_t out_ld, float* coords, const float* gamma,
                                  const float* beta, float eps, float* normed, void* stream) {
  if (!delta || !coords || !gamma || !beta || !normed) return CTK_E_NULL;
  if (S <= 0 || N <= 0 || out_ld < 2 + V2_C) return CTK_E_SHAPE;
  const long rows = (long)S * N;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtkProfScope ps("v2_apply_delta", 0.0, 4.0 * rows * (out_ld + V2_C), s);
  hipLaunchKernelGGL(v2_apply_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, delta, out_ld, S, N, coords, gamma, beta, eps,
                     normed);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_v2_vis_head(const float* track_feat, int64_t R, const float* w, const float* b, float* out, void* stream) {
  if (!track_feat || !w || !b || !out) return CTK_E_NULL;
  if (R <= 0) return CTK_E_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(v2_vis_head_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, track_feat, w, b, (long)R, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_sample_features4d(const float* map, int32_t H, int32_t W, int32_t C, const float* coords, int32_t N, float* out,
                                     void* stream) {
  if (!map || !coords || !out) return CTK_E_NULL;
  if (H <= 0 || W <= 0 || C <= 0 || N <= 0) return CTK_E_SHAPE;
  const long total = (long)N * C;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sample4d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, map, H, W, C, coords, N,
                     ctk_sampler_scale(W), ctk_sampler_scale(H), out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
