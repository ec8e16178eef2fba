// Row-wise kernels of the update iteration: LayerNorm, token assembly (posenc), virtual-token
// broadcast and the output heads fused with the (coords, vis, conf) state update.
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"

namespace {

// ---- LayerNorm over 384 channels: one HALF-wave (32 lanes) per row, 3 float4 per lane ------------
// nn.LayerNorm(384, elementwise_affine=False, eps=1e-6)  blocks.py:411,416 / cotracker.py:539,549
// nn.LayerNorm(384) (affine, eps=1e-5)                    cotracker.py:540 (norm_context)
// 16-byte loads, 16-byte f32 / 8-byte SH stores (the earlier wave-per-row version moved 8 / 4 bytes per lane);
// the two-pass mean / variance is unchanged.
__device__ __forceinline__ float ctk_half_sum(float v) {  // sum over the 32 lanes of my half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// DUAL (round 5): a second, parameter-free normalisation of the same rows with its own eps goes to y2 from the same read --
// the point tokens are normalised twice per depth from the same tokens (norm_context of the virtual<-points block,
// cotracker.py:540/574, and norm1 of the points<-virtual block, :539/573); same arithmetic, same bits as two launches.
template <bool DUAL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, float* y, long R, const float* gamma,
                                                         const float* beta, float eps, int out_split, float* y2, float eps2) {
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int j = threadIdx.x & 31;
  if (row >= R) return;  // whole half-waves leave together (xor-shuffles stay inside a half)
  const float* xr = x + row * CTK_HID + 4 * j;
  f32x4 v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = *reinterpret_cast<const f32x4*>(xr + 128 * k);
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
  const float mean = ctk_half_sum(s) * (1.0f / CTK_HID);
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v[k] -= mean;
    q += (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3]);
  }
  const float var = ctk_half_sum(q) * (1.0f / CTK_HID);
  auto store = [&](float* yo, const float rstd, const float* ga, const float* be) {
    float* yr = yo + row * CTK_HID + 4 * j;
    _Float16* yh = reinterpret_cast<_Float16*>(yo) + row * (2 * CTK_HID);  // SH row: 12 tiles x 64 halves
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = 128 * k + 4 * j;
      f32x4 o = v[k] * rstd;
      if (ga) o = o * *reinterpret_cast<const f32x4*>(ga + c) + *reinterpret_cast<const f32x4*>(be + c);
      if (out_split) {
        // Whole lines: lanes j and j ^ 1 hold 8 consecutive columns; the even lane stores their 8 hi halves (own quad + the
        // neighbour's), the odd lane their 8 lo halves -- 16 bytes each, and the 32 lanes of a row cover 512 consecutive bytes
        // (4 SH lines) per instruction instead of eight 64-byte half-lines in two instructions (round 3, same bits).
        f16x4 hi, lo;
        ctk_split4(o, hi, lo);
        const bool odd = j & 1;
        const f32x2 give = __builtin_bit_cast(f32x2, odd ? hi : lo);
        const f32x2 got = {__shfl_xor(give[0], 1, 64), __shfl_xor(give[1], 1, 64)};
        const f16x4 theirs = __builtin_bit_cast(f16x4, got);
        const f16x8 piece = odd ? ctk_cat8(theirs, lo) : ctk_cat8(hi, theirs);
        _Float16* dst = yh + ctk_sh_col(c & ~7) + (odd ? 32 : 0);
        *reinterpret_cast<f16x8*>(dst) = piece;
      } else {
        *reinterpret_cast<f32x4*>(yr + 128 * k) = o;
      }
    }
  };
  store(y, 1.0f / sqrtf(var + eps), gamma, beta);
  if (DUAL) store(y2, 1.0f / sqrtf(var + eps2), nullptr, nullptr);
}

// ---- token assembly: x[n*S+t][1024..1119] = [vis, conf, posenc(rel fwd/bwd coords), 0-pad] ---
// cotracker3_online.py:212-245 and posenc :19-39.  The time embedding (:247) is folded into the
// input projection's per-frame bias (ctk_model_weights.in_bias_t).
// One thread = 8 consecutive columns of a row (12 threads per row): a 16-byte hi piece and a 16-byte lo piece per thread.  Round 2
// wrote every element with two 2-byte stores, and at ~100 cycles per store instruction and CU (tools/gemm_lab.cpp) the 300 k store
// instructions of a launch were its whole 54 us; same values, same bits.
__device__ __forceinline__ float assemble_value(const float* coords, const float* vis, const float* conf, int S, int N, float scale_x,
                                                float scale_y, int t, int n, int e) {
  if (e == 0) return vis[(long)t * N + n];
  if (e == 1) return conf[(long)t * N + n];
  if (e >= 2 + 84) return 0.0f;
  const int k = e - 2;  // posenc element 0..83
  int comp, deg;
  bool shift = false;
  if (k < 4) { comp = k; deg = -1; }
  else if (k < 44) { comp = (k - 4) & 3; deg = (k - 4) >> 2; }
  else { comp = (k - 44) & 3; deg = (k - 44) >> 2; shift = true; }
  // comp 0,1 = forward (c[t]-c[t+1]) x,y ; comp 2,3 = backward (c[t]-c[t-1]) x,y
  const int axis = comp & 1;
  const bool fwd = comp < 2;
  const int tn = fwd ? t + 1 : t - 1;
  float rel = 0.0f;
  if (tn >= 0 && tn < S) rel = __fsub_rn(coords[((long)t * N + n) * 2 + axis], coords[((long)tn * N + n) * 2 + axis]);
  rel = __fdiv_rn(rel, axis == 0 ? scale_x : scale_y);
  if (deg < 0) return rel;
  float a = __fmul_rn(rel, (float)(1 << deg));
  if (shift) a = __fadd_rn(a, 1.57079637050628662109375f);  // f32(0.5*pi)
  return sinf(a);
}

__global__ void assemble_kernel(const float* coords, const float* vis, const float* conf, int S, int N, float scale_x,
                                float scale_y, float* x, int x_split) {
  constexpr int EW = CTK_X_LD - CTK_X_VIS;  // 96 columns written per row
  constexpr int G8 = EW / 8;                // 12 column octets
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)S * N * G8;
  if (i >= total) return;
  const int e0 = (int)(i % G8) * 8;
  const long row = i / G8;  // n*S + t
  const int t = row % S;
  const int n = row / S;
  f32x4 v[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k >> 2][k & 3] = assemble_value(coords, vis, conf, S, N, scale_x, scale_y, t, n, e0 + k);
  if (x_split) {  // SH row: 35 tiles x 64 halves
    _Float16* xh = reinterpret_cast<_Float16*>(x) + row * (2 * CTK_X_LD) + ctk_sh_col(CTK_X_VIS + e0);
    f16x8 hi, lo;
    ctk_split8(v[0], v[1], hi, lo);
    *reinterpret_cast<f16x8*>(xh) = hi;
    *reinterpret_cast<f16x8*>(xh + 32) = lo;
  } else {
    float* xp = x + row * CTK_X_LD + CTK_X_VIS + e0;
    *reinterpret_cast<f32x4*>(xp) = v[0];
    *reinterpret_cast<f32x4*>(xp + 4) = v[1];
  }
}

// ---- virtual tokens: tokens[(N+v)*S + t] = virual_tracks[v]   (cotracker.py:487-488) --------
__global__ void virtual_init_kernel(const float* vt, int S, float* dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index over 64*S*96
  const long total = (long)CTK_VIRT * S * (CTK_HID / 4);
  if (i >= total) return;
  const int c4 = i % (CTK_HID / 4);
  const int v = i / ((long)(CTK_HID / 4) * S);
  reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(vt)[v * (CTK_HID / 4) + c4];
}

// ---- heads: delta = tokens @ [flow_head; vis_conf_head]^T + b  (cotracker.py:526-529) -------
// optionally fused with coords += d[:2]; vis += d[2]; conf += d[3] (cotracker3_online.py:252-259)
// Round 3: a half-wave per row with 16-byte loads (a row's 32 lanes read 512 consecutive bytes per instruction), the 4 x 384 head
// weights held in registers while the half-wave walks over its rows (they were re-read from cache for every row: 12 of the 15
// loads), 68 -> ~35 us per launch.  Same sums up to the order of the f32 additions.
__global__ __launch_bounds__(256) void heads_kernel(const float* tokens, const float* hw, const float* hb, int S, int N,
                                                     float* delta, float* coords, float* vis, float* conf) {
  const int j = threadIdx.x & 31;
  const long hw_id = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;  // half-wave index
  const long nhw = ((long)gridDim.x * blockDim.x) >> 5;
  const long rows = (long)S * N;
  f32x4 w[4][3];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int k = 0; k < 3; ++k) w[o][k] = *reinterpret_cast<const f32x4*>(hw + o * CTK_HID + 128 * k + 4 * j);
  const float bias[4] = {hb[0], hb[1], hb[2], hb[3]};
  for (long row = hw_id; row < rows; row += nhw) {  // (the bound is uniform over a half-wave: xor-shuffles stay inside it)
    const float* xr = tokens + row * CTK_HID + 4 * j;
    f32x4 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = *reinterpret_cast<const f32x4*>(xr + 128 * k);
    float d[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += (v[k][0] * w[o][k][0] + v[k][1] * w[o][k][1]) + (v[k][2] * w[o][k][2] + v[k][3] * w[o][k][3]);
      d[o] = ctk_half_sum(acc) + bias[o];
    }
    if (j == 0) {
      if (delta) {
        const f32x4 t = {d[0], d[1], d[2], d[3]};
        *reinterpret_cast<f32x4*>(delta + row * 4) = t;
      }
      if (coords) {
        const int t = row % S;
        const int n = row / S;
        const long sn = (long)t * N + n;
        coords[sn * 2] += d[0];
        coords[sn * 2 + 1] += d[1];
        vis[sn] += d[2];
        conf[sn] += d[3];
      }
    }
  }
}

}  // namespace

extern "C" int ctk_layernorm(const float* x, void* y, int64_t R, const float* gamma, const float* beta, float eps,
                             int32_t out_split, void* stream) {
  if (!x || !y) return CTK_E_NULL;
  if (R <= 0) return CTK_E_SHAPE;
  if ((gamma == nullptr) != (beta == nullptr)) return CTK_E_NULL;
  CtkProfScope ps("layernorm", 8.0 * R * CTK_HID, 8.0 * R * CTK_HID, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)((R + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     static_cast<float*>(y), (long)R, gamma, beta, eps, out_split, static_cast<float*>(nullptr), 0.0f);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

// y = LayerNorm(x; gamma, beta, eps) and y2 = LayerNorm(x; no affine, eps2) from ONE read of x (api.hip: the point tokens of a depth)
int ctk_launch_layernorm2(const float* x, void* y, long R, const float* gamma, const float* beta, float eps, void* y2, float eps2,
                          int out_split, hipStream_t s) {
  if (!x || !y || !y2) return CTK_E_NULL;
  if (R <= 0) return CTK_E_SHAPE;
  if ((gamma == nullptr) != (beta == nullptr)) return CTK_E_NULL;
  CtkProfScope ps("layernorm", 12.0 * R * CTK_HID, 12.0 * R * CTK_HID, s);
  hipLaunchKernelGGL(layernorm_kernel<true>, dim3((unsigned)((R + 7) / 8)), dim3(256), 0, s, x, static_cast<float*>(y), R, gamma, beta, eps,
                     out_split, static_cast<float*>(y2), eps2);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_assemble_tokens(const ctk_window_args* a, void* x, int32_t x_split, void* stream) {
  if (!a || !a->coords || !a->vis || !a->conf || !x) return CTK_E_NULL;
  if (a->S <= 0 || a->N <= 0 || !(a->scale_x > 0.f) || !(a->scale_y > 0.f)) return CTK_E_SHAPE;
  const long total = (long)a->S * a->N * ((CTK_X_LD - CTK_X_VIS) / 8);  // threads: 8 columns each
  CtkProfScope ps("assemble_tokens", 0.0, 32.0 * total, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a->coords, a->vis, a->conf, a->S, a->N, a->scale_x, a->scale_y,
                     static_cast<float*>(x), x_split);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

int ctk_launch_virtual_init(const float* vt, int S, float* dst, hipStream_t s) {
  const long total = (long)CTK_VIRT * S * (CTK_HID / 4);
  CtkProfScope ps("virtual_init", 0.0, 16.0 * total, s);
  hipLaunchKernelGGL(virtual_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, vt, S, dst);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

int ctk_launch_heads(const float* tokens, const float* hw, const float* hb, int S, int N, float* delta, float* coords,
                     float* vis, float* conf, hipStream_t s) {
  const long rows = (long)S * N;
  CtkProfScope ps("heads_update", 8.0 * rows * CTK_HID, 4.0 * rows * CTK_HID, s);
  const long want = (rows + 7) / 8;  // one row per half-wave at most; 2048 workgroups walk over the rest
  hipLaunchKernelGGL(heads_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, tokens, hw, hb, S, N, delta, coords,
                     vis, conf);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
