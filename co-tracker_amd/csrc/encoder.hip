// Row / pixel kernels around the encoder's convolutions (conv_pp.hip): everything of BasicEncoder.forward
// (cotracker/models/core/cotracker/blocks.py:184-219) and ResidualBlock.forward (:128-138) that is not a convolution.
// Activations are NHWC; a convolution reads them in SH format ([pixel][C/32] lines of 128 B) and writes f32 [pixel][C].
//   ctk_enc_stem_im2col   2*(v/255)-1 (cotracker3_online.py:320) + the 7x7/2 patch of the 3-channel frame -> SH rows of
//                         160 columns ((ky,kx,c) order, 147 used): conv1 becomes a plain K=160 GEMM
//   ctk_enc_inorm_stats   InstanceNorm2d statistics (no affine, eps 1e-5, biased variance) per (frame, channel)
//   ctk_enc_inorm_apply   y = relu((x - mean) * rstd); optionally out = relu(skip + y) with skip either an f32 tensor or a
//                         second raw convolution output with its own statistics (the 1x1 downsample branch); writes SH
//                         (next convolution's input) and / or f32 (next unit's skip, the multi-scale fusion)
//   ctk_enc_fuse          F.interpolate(bilinear, align_corners=True) of the four stage outputs to H/4 x W/4 + channel concat
//                         (416) -> SH, the input of conv2
//   ctk_enc_l2norm        channel L2 normalisation of the 128-channel NHWC output (cotracker3_online.py:384-394)
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"

namespace {

// ---- stem: scale + im2col of the 7x7 stride-2 pad-3 patch ------------------------------------------------------------
__global__ void enc_stem_im2col_kernel(const float* frames /*[F,3,H,W] 0..255*/, int F, int H, int W, int Ho, int Wo, _Float16* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (pixel, 8-column group): 20 groups per pixel
  const long total = (long)F * Ho * Wo * 20;
  if (i >= total) return;
  const int g8 = (int)(i % 20);
  long r = i / 20;
  const int ox = (int)(r % Wo);
  r /= Wo;
  const int oy = (int)(r % Ho);
  const long f = r / Ho;
  f32x4 v[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = g8 * 8 + e;  // (ky*7 + kx)*3 + c
    float x = 0.0f;
    if (k < 147) {
      const int c = k % 3, t = k / 3, kx = t % 7, ky = t / 7;
      const int iy = oy * 2 + ky - 3, ix = ox * 2 + kx - 3;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
        const float p = frames[((f * 3 + c) * H + iy) * (long)W + ix];
        x = __fsub_rn(__fmul_rn(2.0f, __fdiv_rn(p, 255.0f)), 1.0f);
      }
    }
    v[e >> 2][e & 3] = x;
  }
  f16x8 hi, lo;
  ctk_split8(v[0], v[1], hi, lo);
  // SH row of 160 columns = 5 lines; 8-column group g8 -> line g8 / 4, chunk g8 % 4 (hi) and 4 + g8 % 4 (lo)
  _Float16* row = out + ((f * Ho + oy) * (long)Wo + ox) * 320 + (g8 >> 2) * 64 + (g8 & 3) * 8;
  *reinterpret_cast<f16x8*>(row) = hi;
  *reinterpret_cast<f16x8*>(row + 32) = lo;
}

// ---- instance-norm statistics: partial sums per (frame, pixel block, channel) in f64, then the final mean / rstd ---------
constexpr int ST_ROWS = 512;  // pixels per partial block
__global__ __launch_bounds__(256) void enc_inorm_partial_kernel(const float* x, long HW, int C, int nblk, double* part /*[F][nblk][C][2]*/) {
  const int f = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  const int C4 = C / 4, groups = 256 / C4;             // row groups working in parallel on the block's pixels
  const int c4 = tid % C4, grp = tid / C4;
  __shared__ double red[256][8];
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (grp < groups) {
    const long p0 = (long)b * ST_ROWS, p1 = min(p0 + ST_ROWS, HW);
    const float* base = x + ((long)f * HW) * C + c4 * 4;
    // 4 rows in flight per thread (a single dependent load per iteration left the kernel latency-bound at 1.9 TB/s); f32 partial
    // sums over at most 4 values, f64 beyond
    long p = p0 + grp;
    for (; p + 3 * groups < p1; p += 4 * groups) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (p + (long)u * groups) * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] += (double)((v[0][e] + v[1][e]) + (v[2][e] + v[3][e]));
        ss[e] += (double)v[0][e] * (double)v[0][e] + (double)v[1][e] * (double)v[1][e] + (double)v[2][e] * (double)v[2][e] + (double)v[3][e] * (double)v[3][e];
      }
    }
    for (; p < p1; p += groups) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + p * C);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += (double)v[e]; ss[e] += (double)v[e] * (double)v[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid][e] = s[e]; red[tid][4 + e] = ss[e]; }
  __syncthreads();
  if (tid < C4) {  // fixed summation order over the row groups: deterministic
    for (int gi = 1; gi < groups; ++gi)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[tid][e] += red[gi * C4 + tid][e];
    double* o = part + (((long)f * nblk + b) * C + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = red[tid][e]; o[2 * e + 1] = red[tid][4 + e]; }
  }
}
__global__ void enc_inorm_final_kernel(const double* part, long HW, int C, int nblk, int FC, float eps, float* stats /*[F][C][2] mean, rstd*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (f, c)
  if (i >= FC) return;
  const int f = i / C, c = i % C;
  double s = 0, ss = 0;
  for (int b = 0; b < nblk; ++b) {
    const double* o = part + (((long)f * nblk + b) * C + c) * 2;
    s += o[0];
    ss += o[1];
  }
  const double mean = s / (double)HW;
  const double var = fmax(ss / (double)HW - mean * mean, 0.0);  // biased (F.instance_norm / batch_norm training statistics)
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- normalise (+ ReLU) (+ skip, ReLU) ------------------------------------------------------------------------------
// one thread = 8 consecutive channels of one pixel
__global__ void enc_inorm_apply_kernel(const float* x, const float* stats, const float* skip, const float* skip_stats, long HW, int C, long total8,
                                       _Float16* out_sh, float* out_f32) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int C8 = C / 8;
  const int c8 = (int)(i % C8);
  const long pixel = i / C8;
  const long f = pixel / HW;
  const float* st = stats + (f * C + c8 * 8) * 2;
  const float* xp = x + pixel * C + c8 * 8;
  f32x4 v[2] = {*reinterpret_cast<const f32x4*>(xp), *reinterpret_cast<const f32x4*>(xp + 4)};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float y = __fmul_rn(__fsub_rn(v[e >> 2][e & 3], st[2 * e]), st[2 * e + 1]);
    y = fmaxf(y, 0.0f);
    v[e >> 2][e & 3] = y;
  }
  if (skip) {
    const float* sp = skip + pixel * C + c8 * 8;
    f32x4 k[2] = {*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4)};
    if (skip_stats) {  // the downsample branch: InstanceNorm of the raw 1x1 convolution, no ReLU (blocks.py:123-126)
      const float* ks = skip_stats + (f * C + c8 * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) k[e >> 2][e & 3] = __fmul_rn(__fsub_rn(k[e >> 2][e & 3], ks[2 * e]), ks[2 * e + 1]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = fmaxf(__fadd_rn(k[e >> 2][e & 3], v[e >> 2][e & 3]), 0.0f);  // relu(x + y)
  }
  if (out_f32) {
    float* op = out_f32 + pixel * C + c8 * 8;
    *reinterpret_cast<f32x4*>(op) = v[0];
    *reinterpret_cast<f32x4*>(op + 4) = v[1];
  }
  if (out_sh) {
    f16x8 hi, lo;
    ctk_split8(v[0], v[1], hi, lo);
    _Float16* row = out_sh + pixel * 2 * C + (c8 >> 2) * 64 + (c8 & 3) * 8;
    *reinterpret_cast<f16x8*>(row) = hi;
    *reinterpret_cast<f16x8*>(row + 32) = lo;
  }
}

// ---- multi-scale fusion: bilinear (align_corners=True) resize to Ho x Wo + concat -> SH [F][Ho][Wo][13 lines] ---------------
struct EncFuseP {
  const float* src[4];
  int H[4], W[4], C[4], c0[4];  // source sizes, channels, first channel in the concatenation
  int F, Ho, Wo, Ctot;
};
__global__ void enc_fuse_kernel(EncFuseP p, _Float16* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (pixel, 8-channel group)
  const int C8 = p.Ctot / 8;
  const long total = (long)p.F * p.Ho * p.Wo * C8;
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  long r = i / C8;
  const int ox = (int)(r % p.Wo);
  r /= p.Wo;
  const int oy = (int)(r % p.Ho);
  const long f = r / p.Ho;
  const int ch = c8 * 8;
  int s = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (ch >= p.c0[k]) s = k;
  const int H = p.H[s], W = p.W[s], C = p.C[s], c = ch - p.c0[s];
  // ATen upsample_bilinear2d, align_corners=True: scale = (in - 1) / (out - 1) in float, src = scale * dst
  const float sy = p.Ho > 1 ? (float)(H - 1) / (float)(p.Ho - 1) : 0.0f, sx = p.Wo > 1 ? (float)(W - 1) / (float)(p.Wo - 1) : 0.0f;
  const float fy = __fmul_rn(sy, (float)oy), fx = __fmul_rn(sx, (float)ox);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = __fsub_rn(fy, (float)y0), lx1 = __fsub_rn(fx, (float)x0);
  const float ly0 = __fsub_rn(1.0f, ly1), lx0 = __fsub_rn(1.0f, lx1);
  const float* base = p.src[s] + (f * H * (long)W) * C + c;
  f32x4 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x0) * C + 4 * h);
    const f32x4 b = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x1) * C + 4 * h);
    const f32x4 cc = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x0) * C + 4 * h);
    const f32x4 d = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x1) * C + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e)  // h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (w0lambda * c + w1lambda * d)
      v[h][e] = __fadd_rn(__fmul_rn(ly0, __fadd_rn(__fmul_rn(lx0, a[e]), __fmul_rn(lx1, b[e]))),
                          __fmul_rn(ly1, __fadd_rn(__fmul_rn(lx0, cc[e]), __fmul_rn(lx1, d[e]))));
  }
  f16x8 hi, lo;
  ctk_split8(v[0], v[1], hi, lo);
  _Float16* row = out + ((f * p.Ho + oy) * (long)p.Wo + ox) * 2 * p.Ctot + (c8 >> 2) * 64 + (c8 & 3) * 8;
  *reinterpret_cast<f16x8*>(row) = hi;
  *reinterpret_cast<f16x8*>(row + 32) = lo;
}

// ---- channel L2 normalisation of NHWC [P][128] (cotracker3_online.py:384-394): one wave per pixel -------------------------
__global__ __launch_bounds__(256) void enc_l2norm_kernel(const float* x, long P, float* out) {
  const long pixel = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pixel >= P) return;
  const int lane = threadIdx.x & 63;
  const float a = x[pixel * 128 + lane], b = x[pixel * 128 + 64 + lane];
  float ss = ctk_wave_sum(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)));
  const float n = sqrtf(fmaxf(ss, 1e-12f));
  out[pixel * 128 + lane] = a / n;
  out[pixel * 128 + 64 + lane] = b / n;
}

}  // namespace

extern "C" int ctk_enc_stem_im2col(const float* frames, int32_t F, int32_t H, int32_t W, void* out_sh, void* stream) {
  if (!frames || !out_sh) return CTK_E_NULL;
  if (F <= 0 || H < 7 || W < 7) return CTK_E_SHAPE;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const long total = (long)F * Ho * Wo * 20;
  CtkProfScope ps("enc_stem_im2col", 0.0, 4.0 * (3.0 * F * H * W + 160.0 * F * Ho * Wo), static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(enc_stem_im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), frames, F, H, W, Ho,
                     Wo, static_cast<_Float16*>(out_sh));
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_enc_inorm_workspace_bytes(int32_t F, int64_t HW, int32_t C, size_t* out_bytes) {
  if (!out_bytes) return CTK_E_NULL;
  if (F <= 0 || HW <= 0 || C <= 0) return CTK_E_SHAPE;
  *out_bytes = (size_t)F * ((HW + ST_ROWS - 1) / ST_ROWS) * C * 2 * sizeof(double);
  return CTK_OK;
}

extern "C" int ctk_enc_inorm_stats(const float* x, int32_t F, int64_t HW, int32_t C, float eps, float* stats, void* workspace, void* stream) {
  if (!x || !stats || !workspace) return CTK_E_NULL;
  if (F <= 0 || HW <= 0 || C <= 0 || (C % 8) || C > 1024 || 256 / (C / 4) < 1) return CTK_E_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nblk = (int)((HW + ST_ROWS - 1) / ST_ROWS);
  CtkProfScope ps("enc_inorm_stats", 0.0, 4.0 * F * (double)HW * C, s);
  hipLaunchKernelGGL(enc_inorm_partial_kernel, dim3((unsigned)nblk, (unsigned)F), dim3(256), 0, s, x, (long)HW, C, nblk, static_cast<double*>(workspace));
  CTK_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(enc_inorm_final_kernel, dim3((unsigned)((F * C + 255) / 256)), dim3(256), 0, s, static_cast<const double*>(workspace), (long)HW, C,
                     nblk, F * C, eps, stats);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_enc_inorm_apply(const float* x, const float* stats, const float* skip, const float* skip_stats, int32_t F, int64_t HW, int32_t C,
                                   void* out_sh, float* out_f32, void* stream) {
  if (!x || !stats || (!out_sh && !out_f32)) return CTK_E_NULL;
  if (F <= 0 || HW <= 0 || C <= 0 || (C % 32)) return CTK_E_SHAPE;
  const long total8 = (long)F * HW * (C / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtkProfScope ps("enc_inorm_apply", 0.0, 4.0 * F * (double)HW * C * (1.0 + (skip ? 1.0 : 0.0) + (out_sh ? 1.0 : 0.0) + (out_f32 ? 1.0 : 0.0)), s);
  hipLaunchKernelGGL(enc_inorm_apply_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, x, stats, skip, skip_stats, (long)HW, C, total8,
                     static_cast<_Float16*>(out_sh), out_f32);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_enc_fuse(const float* const* src, const int32_t* H, const int32_t* W, const int32_t* C, int32_t F, int32_t Ho, int32_t Wo, void* out_sh,
                            void* stream) {
  if (!src || !H || !W || !C || !out_sh) return CTK_E_NULL;
  EncFuseP p;
  int c0 = 0;
  for (int k = 0; k < 4; ++k) {
    if (!src[k]) return CTK_E_NULL;
    if (H[k] <= 0 || W[k] <= 0 || C[k] <= 0 || (C[k] % 8)) return CTK_E_SHAPE;
    p.src[k] = src[k]; p.H[k] = H[k]; p.W[k] = W[k]; p.C[k] = C[k]; p.c0[k] = c0;
    c0 += C[k];
  }
  if (F <= 0 || Ho <= 0 || Wo <= 0 || (c0 % 32)) return CTK_E_SHAPE;
  p.F = F; p.Ho = Ho; p.Wo = Wo; p.Ctot = c0;
  const long total = (long)F * Ho * Wo * (c0 / 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtkProfScope ps("enc_fuse", 0.0, 8.0 * F * (double)Ho * Wo * c0, s);
  hipLaunchKernelGGL(enc_fuse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, static_cast<_Float16*>(out_sh));
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_enc_l2norm(const float* x, int64_t P, float* out, void* stream) {
  if (!x || !out) return CTK_E_NULL;
  if (P <= 0) return CTK_E_SHAPE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtkProfScope ps("enc_l2norm", 0.0, 8.0 * (double)P * 128, s);
  hipLaunchKernelGGL(enc_l2norm_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, s, x, (long)P, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
