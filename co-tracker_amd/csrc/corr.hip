// Fused correlation sampler: bilinear 7x7x128 patch gather + 49x49 correlation (fp32 MFMA).
//
// Replaces, per pyramid level and iteration (cotracker3_online.py:192-204):
//   get_correlation_feat (:130-143) -> bilinear_sampler (model_utils.py:191-255)
//   torch.einsum("btnhwc,bnijc->btnhwij")  (:202-204)
// without ever materialising the sampled patches (19.3 GB/level at T=120, N=6400).
//
// Work decomposition.  A workgroup (4 waves) owns one (point n, level l, chunk of <=TC frames):
// the point's support patch [49,128] is staged once in LDS (rows 49..63 zero) and reused for
// every frame of the chunk -- it is the B operand.  The A operand is the sampled patch:
// rows r = (t,p) of the chunk flattened (TC*49 rows), 16 rows per MFMA tile, so a 16-frame
// window is exactly 49 tiles with no row padding.  A lane needs A[row = lane&15][channel
// 16j+4g+e] (g = lane>>4) for MFMA step (j,e): it loads the four bilinear corners of ITS row
// at channels [16j+4g, 16j+4g+4) as float4 straight from the NHWC pyramid (a pixel's 128
// channels are 512 contiguous bytes; 4 lanes cover 64 B of one pixel), blends them in
// registers with the reference's exact float32 sequence, and feeds v_mfma_f32_16x16x4_f32.
// The k index supplied by lane group g at step (j,e) is 16j+4g+e for both operands.
// D (16 rows x 64 cols, cols 49..63 dropped) goes to the correlation buffer row (l, n, t),
// column p*49+q  ==  the reference's (h,w,i,j) row-major flattening (:205).
#include "ctk_common.h"
#include "ctk_profile.h"

namespace {

constexpr int TC = 16;          // frames per workgroup
constexpr int SUP_PITCH = 132;  // floats; 132 % 64 = 4 -> rows land on distinct 4-bank groups

struct CorrP {
  const float* fmaps[CTK_LEVELS];
  const float* support[CTK_LEVELS];
  int H[CTK_LEVELS], W[CTK_LEVELS];
  float sx[CTK_LEVELS], sy[CTK_LEVELS];
  const float* coords;  // [S,N,2]
  const uint8_t* mask;  // [N] or null
  float* out;           // [L][nchunk*S][ld]
  long out_level_stride;
  int ld;
  int S, N, n0, ncount, tchunks;
};

__global__ __launch_bounds__(256) void corr_volume_kernel(CorrP p) {
  __shared__ __attribute__((aligned(16))) float sup[64 * SUP_PITCH];

  unsigned bid = ctk_xcd_remap(blockIdx.x, gridDim.x);
  const int tc = bid % p.tchunks;
  bid /= p.tchunks;
  const int lvl = bid % CTK_LEVELS;
  const int nl = bid / CTK_LEVELS;  // local point index
  const int n = p.n0 + nl;
  const int t0 = tc * TC;
  const int nt = min(TC, p.S - t0);
  const int rows = nt * CTK_TAPS;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* out_base = p.out + (long)lvl * p.out_level_stride + ((long)nl * p.S + t0) * p.ld;

  // zero the K padding columns [2401, ld) of this chunk's rows
  const int padw = p.ld - CTK_CORR_K;
  for (int i = tid; i < nt * padw; i += 256) out_base[(long)(i / padw) * p.ld + CTK_CORR_K + i % padw] = 0.0f;

  const bool live = p.mask ? (p.mask[n] != 0) : true;
  if (!live) {  // support features of not-yet-queried tracks are zeroed (cotracker3_online.py:493-496)
    for (int i = tid; i < nt * CTK_CORR_K; i += 256) out_base[(long)(i / CTK_CORR_K) * p.ld + i % CTK_CORR_K] = 0.0f;
    return;
  }

  // stage support patch [49][128] -> LDS [64][132], rows 49..63 zero
  const float* sp = p.support[lvl] + (long)n * CTK_TAPS * CTK_C;
  for (int i = tid; i < 64 * 32; i += 256) {
    const int row = i >> 5, c4 = i & 31;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < CTK_TAPS) v = *reinterpret_cast<const f32x4*>(sp + row * CTK_C + c4 * 4);
    *reinterpret_cast<f32x4*>(&sup[row * SUP_PITCH + c4 * 4]) = v;
  }
  __syncthreads();

  const int H = p.H[lvl], W = p.W[lvl];
  const float sx = p.sx[lvl], sy = p.sy[lvl];
  const float inv = 1.0f / (float)(1 << lvl);  // coords / 2**i : exact
  const float* fm = p.fmaps[lvl];
  const int li = lane & 15, g = lane >> 4;

  // Software pipeline over (m-tile, channel-quarter) steps: the 8 corner float4 of the NEXT step are
  // in flight while the current step blends + feeds 32 MFMAs, so the matrix pipe does not wait on
  // the gather.  Two small register buffers alternate statically (quarters 0,2 -> buf0; 1,3 -> buf1).
  const int mtiles = (rows + 15) >> 4;
  struct Row {
    unsigned o00, o10, o01, o11;  // element offsets of the four corners (incl. frame and lane-group offset)
    float w00, w10, w01, w11;
    bool valid;
  };
  auto setup = [&](int mt) {
    Row r;
    const int ri = mt * 16 + li;
    r.valid = ri < rows;
    const int rr = r.valid ? ri : rows - 1;
    const int tl = rr / CTK_TAPS, pp = rr - tl * CTK_TAPS;
    const int hx = pp / 7, wy = pp - hx * 7;  // first 7-index = x offset, second = y (cotracker3_online.py:102-104)
    const float* cptr = p.coords + ((long)(t0 + tl) * p.N + n) * 2;
    const float cx = __fmul_rn(cptr[0], inv), cy = __fmul_rn(cptr[1], inv);
    const CtkTap tx = ctk_tap(__fadd_rn(cx, (float)(hx - 3)), W, sx);
    const CtkTap ty = ctk_tap(__fadd_rn(cy, (float)(wy - 3)), H, sy);
    r.w00 = __fmul_rn(tx.w0, ty.w0); r.w10 = __fmul_rn(tx.w1, ty.w0);
    r.w01 = __fmul_rn(tx.w0, ty.w1); r.w11 = __fmul_rn(tx.w1, ty.w1);
    const unsigned fo = (unsigned)(t0 + tl) * (unsigned)(H * W) ;
    r.o00 = (fo + (unsigned)(ty.i0 * W + tx.i0)) * CTK_C + g * 4;
    r.o10 = (fo + (unsigned)(ty.i0 * W + tx.i1)) * CTK_C + g * 4;
    r.o01 = (fo + (unsigned)(ty.i1 * W + tx.i0)) * CTK_C + g * 4;
    r.o11 = (fo + (unsigned)(ty.i1 * W + tx.i1)) * CTK_C + g * 4;
    return r;
  };
  struct Buf { f32x4 v[2][4]; };  // [j within quarter][corner]
  auto gather = [&](const Row& r, int quarter, Buf& bf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned off = (quarter * 2 + j) * 16;
      bf.v[j][0] = *reinterpret_cast<const f32x4*>(fm + r.o00 + off);
      bf.v[j][1] = *reinterpret_cast<const f32x4*>(fm + r.o10 + off);
      bf.v[j][2] = *reinterpret_cast<const f32x4*>(fm + r.o01 + off);
      bf.v[j][3] = *reinterpret_cast<const f32x4*>(fm + r.o11 + off);
    }
  };
  f32x4 acc[4];
  auto consume = [&](const Row& r, int quarter, const Buf& bf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // corner order of ATen grid_sampler_3d: (x0,y0),(x1,y0),(x0,y1),(x1,y1).  The fused path may
        // use FMA here (values feed a 128-term dot product anyway); the standalone sampler
        // (sample_patches_kernel) keeps the bit-exact mul+add sequence.
        float s_ = bf.v[j][0][e] * r.w00;
        s_ = fmaf(bf.v[j][1][e], r.w10, s_);
        s_ = fmaf(bf.v[j][2][e], r.w01, s_);
        s_ = fmaf(bf.v[j][3][e], r.w11, s_);
        a[e] = r.valid ? s_ : 0.0f;
      }
      f32x4 b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        b[q] = *reinterpret_cast<const f32x4*>(&sup[(q * 16 + li) * SUP_PITCH + (quarter * 2 + j) * 16 + g * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[q][e], acc[q], 0, 0, 0);
    }
  };

  Buf buf0, buf1;
  Row cur = setup(min(wave, mtiles - 1));
  if (wave < mtiles) gather(cur, 0, buf0);
  for (int mt = wave; mt < mtiles; mt += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    gather(cur, 1, buf1);
    consume(cur, 0, buf0);
    gather(cur, 2, buf0);
    consume(cur, 1, buf1);
    gather(cur, 3, buf1);
    consume(cur, 2, buf0);
    const Row nxt = setup(min(mt + 4, mtiles - 1));
    if (mt + 4 < mtiles) gather(nxt, 0, buf0);
    consume(cur, 3, buf1);

    // D layout (16x16): col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int orow = mt * 16 + g * 4 + reg;
      if (orow < rows) {
        const int otl = orow / CTK_TAPS, opp = orow - otl * CTK_TAPS;
        float* dst = out_base + (long)otl * p.ld + opp * CTK_TAPS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = q * 16 + li;
          if (col < CTK_TAPS) dst[col] = acc[q][reg];
        }
      }
    }
    cur = nxt;
  }
}

// ----- tap indices (bit-exactness probe) -------------------------------------------------
__global__ void tap_indices_kernel(CorrP p, int32_t* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over S*N*L*2*7
  const long total = (long)p.S * p.N * CTK_LEVELS * 14;
  if (i >= total) return;
  const int k = i % 7;
  const int axis = (i / 7) % 2;
  const int lvl = (i / 14) % CTK_LEVELS;
  const long tn = i / (14 * CTK_LEVELS);
  const float inv = 1.0f / (float)(1 << lvl);
  const float c = __fmul_rn(p.coords[tn * 2 + axis], inv);
  const CtkTap t = axis == 0 ? ctk_tap(__fadd_rn(c, (float)(k - 3)), p.W[lvl], p.sx[lvl])
                             : ctk_tap(__fadd_rn(c, (float)(k - 3)), p.H[lvl], p.sy[lvl]);
  out[i] = t.i0;
}

// ----- get_correlation_feat standalone: out [S,N,49,128] ---------------------------------
__global__ __launch_bounds__(256) void sample_patches_kernel(const float* fm, int S, int H, int W, float sx, float sy,
                                                              const float* coords, int N, int level, float* out) {
  // one wave per (t,n,tap): lane handles 2 channels
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long total = (long)S * N * CTK_TAPS;
  if (wid >= total) return;
  const int pp = wid % CTK_TAPS;
  const long tn = wid / CTK_TAPS;
  const int t = tn / N;
  const int hx = pp / 7, wy = pp - hx * 7;
  const float inv = 1.0f / (float)(1 << level);
  const float cx = __fmul_rn(coords[tn * 2], inv), cy = __fmul_rn(coords[tn * 2 + 1], inv);
  const CtkTap tx = ctk_tap(__fadd_rn(cx, (float)(hx - 3)), W, sx);
  const CtkTap ty = ctk_tap(__fadd_rn(cy, (float)(wy - 3)), H, sy);
  const float w00 = __fmul_rn(tx.w0, ty.w0), w10 = __fmul_rn(tx.w1, ty.w0);
  const float w01 = __fmul_rn(tx.w0, ty.w1), w11 = __fmul_rn(tx.w1, ty.w1);
  const float* frame = fm + (long)t * H * W * CTK_C + lane * 2;
  const float2 v00 = *reinterpret_cast<const float2*>(frame + ((long)ty.i0 * W + tx.i0) * CTK_C);
  const float2 v10 = *reinterpret_cast<const float2*>(frame + ((long)ty.i0 * W + tx.i1) * CTK_C);
  const float2 v01 = *reinterpret_cast<const float2*>(frame + ((long)ty.i1 * W + tx.i0) * CTK_C);
  const float2 v11 = *reinterpret_cast<const float2*>(frame + ((long)ty.i1 * W + tx.i1) * CTK_C);
  float2 o;
  o.x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v00.x, w00), __fmul_rn(v10.x, w10)), __fmul_rn(v01.x, w01)),
                  __fmul_rn(v11.x, w11));
  o.y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v00.y, w00), __fmul_rn(v10.y, w10)), __fmul_rn(v01.y, w01)),
                  __fmul_rn(v11.y, w11));
  *reinterpret_cast<float2*>(out + wid * CTK_C + lane * 2) = o;
}

// ----- get_track_feat: trilinear support patches, out [N,49,128] --------------------------
__global__ __launch_bounds__(256) void sample_support_kernel(const float* fm, int T, int H, int W, float sx, float sy,
                                                              float sz, const float* frames, const float* coords, int N,
                                                              float* out) {
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long total = (long)N * CTK_TAPS;
  if (wid >= total) return;
  const int pp = wid % CTK_TAPS;
  const int n = wid / CTK_TAPS;
  const int hx = pp / 7, wy = pp - hx * 7;
  const CtkTap tx = ctk_tap(__fadd_rn(coords[n * 2], (float)(hx - 3)), W, sx);
  const CtkTap ty = ctk_tap(__fadd_rn(coords[n * 2 + 1], (float)(wy - 3)), H, sy);
  const CtkTap tz = ctk_tap(__fadd_rn(frames[n], 0.0f), T, sz);
  float2 o = make_float2(0.f, 0.f);
  const int zi[2] = {tz.i0, tz.i1};
  const float zw[2] = {tz.w0, tz.w1};
  const int yi[2] = {ty.i0, ty.i1};
  const float yw[2] = {ty.w0, ty.w1};
  const int xi[2] = {tx.i0, tx.i1};
  const float xw[2] = {tx.w0, tx.w1};
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // weight (wx*wy)*wz, corners z0{(x0,y0),(x1,y0),(x0,y1),(x1,y1)} then z1 (ATen grid_sampler_3d)
        const float w = __fmul_rn(__fmul_rn(xw[dx], yw[dy]), zw[dz]);
        const float2 v = *reinterpret_cast<const float2*>(fm + (((long)zi[dz] * H + yi[dy]) * W + xi[dx]) * CTK_C + lane * 2);
        o.x = __fadd_rn(o.x, __fmul_rn(v.x, w));
        o.y = __fadd_rn(o.y, __fmul_rn(v.y, w));
      }
  *reinterpret_cast<float2*>(out + wid * CTK_C + lane * 2) = o;
}

// ----- channel L2-normalise + NCHW -> NHWC (cotracker3_online.py:384-394) ------------------
__global__ __launch_bounds__(256) void normalize_nhwc_kernel(const float* in, long HW, float* out) {
  // block: 64 pixels x 128 channels of frame blockIdx.y
  __shared__ float tile[CTK_C][65];
  __shared__ float rnorm[64];
  const long f = blockIdx.y;
  const long p0 = (long)blockIdx.x * 64;
  const int tid = threadIdx.x;
  const float* src = in + f * CTK_C * HW;
  for (int i = tid; i < CTK_C * 64; i += 256) {
    const int c = i >> 6, px = i & 63;
    tile[c][px] = (p0 + px < HW) ? src[(long)c * HW + p0 + px] : 0.0f;
  }
  __syncthreads();
  if (tid < 64) {
    float ss = 0.0f;
    for (int c = 0; c < CTK_C; ++c) ss += tile[c][tid] * tile[c][tid];
    rnorm[tid] = sqrtf(fmaxf(ss, 1e-12f));
  }
  __syncthreads();
  float* dst = out + (f * HW + p0) * CTK_C;
  for (int i = tid; i < 64 * CTK_C; i += 256) {
    const int px = i >> 7, c = i & 127;
    if (p0 + px < HW) dst[(long)px * CTK_C + c] = tile[c][px] / rnorm[px];
  }
}

// ----- 2x2 average pooling on NHWC (F.avg_pool2d(2, stride=2), cotracker3_online.py:401-409)
__global__ void avg_pool2_nhwc_kernel(const float* in, int F, int H, int W, float* out) {
  const int Ho = H / 2, Wo = W / 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  const long total = (long)F * Ho * Wo * (CTK_C / 4);
  if (i >= total) return;
  const int c4 = i % (CTK_C / 4);
  long r = i / (CTK_C / 4);
  const int xo = r % Wo; r /= Wo;
  const int yo = r % Ho;
  const long f = r / Ho;
  const float* base = in + ((f * H + 2 * yo) * W + 2 * xo) * CTK_C + c4 * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(base);
  const f32x4 b = *reinterpret_cast<const f32x4*>(base + CTK_C);
  const f32x4 c = *reinterpret_cast<const f32x4*>(base + (long)W * CTK_C);
  const f32x4 d = *reinterpret_cast<const f32x4*>(base + (long)W * CTK_C + CTK_C);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(a[e], b[e]), c[e]), d[e]), 0.25f);
  *reinterpret_cast<f32x4*>(out + i * 4) = o;
}

int fill_corr_params(const ctk_window_args* a, CorrP& p) {
  if (!a) return CTK_E_NULL;
  if (a->S <= 0 || a->N <= 0) return CTK_E_SHAPE;
  for (int l = 0; l < CTK_LEVELS; ++l) {
    if (!a->fmaps[l] || !a->support[l]) return CTK_E_NULL;
    if (a->H[l] <= 0 || a->W[l] <= 0) return CTK_E_SHAPE;
    if (!ctk_aligned16(a->fmaps[l]) || !ctk_aligned16(a->support[l])) return CTK_E_ALIGN;
    p.fmaps[l] = a->fmaps[l];
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
  return CTK_OK;
}

}  // namespace

// Internal launcher shared with api.hip: correlation volumes of points [n0, n0+ncount) into
// out[l][(n-n0)*S + t][ld].
int ctk_launch_corr_volume(const ctk_window_args* a, int n0, int ncount, float* out, long level_stride, int ld,
                           hipStream_t s) {
  CorrP p;
  int rc = fill_corr_params(a, p);
  if (rc) return rc;
  if (!out) return CTK_E_NULL;
  p.out = out;
  p.out_level_stride = level_stride;
  p.ld = ld;
  p.n0 = n0;
  p.ncount = ncount;
  p.tchunks = (a->S + TC - 1) / TC;
  const long blocks = (long)ncount * CTK_LEVELS * p.tchunks;
  // algorithmic work per (t,n,level): 2*49*49*128 flop; (2r+2)^2*128*4 B footprint + support/S + volume out
  const double units = (double)ncount * a->S * CTK_LEVELS;
  CtkProfScope ps("corr_volume", units * 2.0 * 49 * 49 * 128,
                  units * (64.0 * 128 * 4 + 49.0 * 128 * 4 / a->S + 2.0 + 2401.0 * 4), s);
  hipLaunchKernelGGL(corr_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_corr_volume(const ctk_window_args* a, float* out, void* stream) {
  if (!a) return CTK_E_NULL;
  return ctk_launch_corr_volume(a, 0, a->N, out, (long)a->N * a->S * CTK_CORR_LD, CTK_CORR_LD,
                                static_cast<hipStream_t>(stream));
}

extern "C" int ctk_tap_indices(const ctk_window_args* a, int32_t* out, void* stream) {
  CorrP p;
  ctk_window_args tmp = *a;
  static const float dummy = 0.f;
  for (int l = 0; l < CTK_LEVELS; ++l) {  // indices need only the level sizes
    if (!tmp.fmaps[l]) tmp.fmaps[l] = reinterpret_cast<const float*>(16);
    if (!tmp.support[l]) tmp.support[l] = reinterpret_cast<const float*>(16);
  }
  (void)dummy;
  int rc = fill_corr_params(&tmp, p);
  if (rc) return rc;
  if (!out) return CTK_E_NULL;
  const long total = (long)a->S * a->N * CTK_LEVELS * 14;
  hipLaunchKernelGGL(tap_indices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_sample_patches(const float* fmap, int32_t S, int32_t H, int32_t W, const float* coords, int32_t N,
                                  int32_t level, float* out, void* stream) {
  if (!fmap || !coords || !out) return CTK_E_NULL;
  if (S <= 0 || H <= 0 || W <= 0 || N <= 0 || level < 0 || level > 16) return CTK_E_SHAPE;
  const long waves = (long)S * N * CTK_TAPS;
  hipLaunchKernelGGL(sample_patches_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), fmap, S, H, W, ctk_sampler_scale(W), ctk_sampler_scale(H), coords,
                     N, level, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_sample_support(const float* fmap, int32_t T, int32_t H, int32_t W, const float* frames,
                                  const float* coords, int32_t N, float* out, void* stream) {
  if (!fmap || !frames || !coords || !out) return CTK_E_NULL;
  if (T <= 0 || H <= 0 || W <= 0 || N <= 0) return CTK_E_SHAPE;
  const long waves = (long)N * CTK_TAPS;
  hipLaunchKernelGGL(sample_support_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), fmap, T, H, W, ctk_sampler_scale(W), ctk_sampler_scale(H),
                     ctk_sampler_scale(T), frames, coords, N, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_normalize_to_nhwc(const float* in, int32_t F, int32_t H, int32_t W, float* out, void* stream) {
  if (!in || !out) return CTK_E_NULL;
  if (F <= 0 || H <= 0 || W <= 0) return CTK_E_SHAPE;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(normalize_nhwc_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)F), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, HW, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}

extern "C" int ctk_avg_pool2_nhwc(const float* in, int32_t F, int32_t H, int32_t W, float* out, void* stream) {
  if (!in || !out) return CTK_E_NULL;
  if (F <= 0 || H < 2 || W < 2) return CTK_E_SHAPE;
  const long total = (long)F * (H / 2) * (W / 2) * (CTK_C / 4);
  hipLaunchKernelGGL(avg_pool2_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, F, H, W, out);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
