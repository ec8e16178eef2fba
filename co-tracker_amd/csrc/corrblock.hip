// CorrBlock.corr + CorrBlock.sample fused (CoTracker2's 4D correlation-volume sampler, blocks.py:284-362).
//
// The reference materialises, per pyramid level, the full volume corrs[b,s,n,h,w] = <track_feat[b,s,n,:],
// fmaps_l[b,s,:,h,w]> / sqrt(C) (blocks.py:342-362; 12 288 floats per (s,n) at level 0) and then reads 4 bilinear
// corners x 49 taps of it back through bilinear_sampler / grid_sampler_2d (blocks.py:309-340, padding "border",
// align_corners=True).  Only the <= 9x9 pixels under the 7x7 tap lattice are ever used, so this kernel computes
// exactly those dot products and blends them: the volume never exists in memory.  HBM traffic per (s,n,level) is
// the footprint (<= 81 pixels x 512 B, contiguous per pixel in the NHWC pyramid) + 196 B of output, against
// H*W*4 B written and re-read by the reference -- an HBM-bound gather, no MFMA.
//
// Exactness: tap indices / weights are ctk_tap (same coordinate pipeline as grid_sampler_2d: the vectorised CPU
// kernel forms (g+1)*((size-1)/2), which rounds identically to ((g+1)/2)*(size-1)); the blend is
// nw*w_nw -> fma(ne) -> fma(sw) -> fma(se) in ATen's corner order; the division by sqrt(C) is an IEEE divide as
// in blocks.py:361.  Only the order of the 128-term dot product differs from the reference's BLAS matmul.
//
// Workgroup = one (frame s, point n), 4 waves = the 4 pyramid levels.  Lane (h = lane>>5, c = lane&31) owns
// channels 4c..4c+3 of the track feature; footprint pixels are processed two at a time (one per half-wave,
// 512 contiguous bytes per half-wave load), the 32 partial sums of a pixel are folded with xor-shuffles.
#include "ctk_common.h"
#include "ctk_profile.h"

namespace {

constexpr int FPMAX = 9;  // footprint is at most 9 x 9 pixels (7 taps spaced 1.0 apart + f32 round-trip slack)

struct CorrBlockP {
  const float* fm[CTK_LEVELS];  // NHWC [S,H_l,W_l,128]
  int H[CTK_LEVELS], W[CTK_LEVELS];
  float sx[CTK_LEVELS], sy[CTK_LEVELS];
  const float* targets;  // [S,N,128]
  const float* coords;   // [S,N,2] level-0 units
  float* out;            // [N,S,196]
  int S, N;
  float sqrt_c;
};

struct LevelTab {  // per-wave scratch in LDS
  int x0[7], x1[7], y0[7], y1[7];
  float wx0[7], wx1[7], wy0[7], wy1[7];
  float c[FPMAX * FPMAX + 3];
};

__global__ __launch_bounds__(256) void corrblock_sample_kernel(CorrBlockP p) {
  __shared__ LevelTab tabs[CTK_LEVELS];
  const int sn = blockIdx.x;  // s*N + n
  const int s = sn / p.N, n = sn - s * p.N;
  const int lvl = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, c = lane & 31;
  LevelTab& tab = tabs[lvl];
  const int H = p.H[lvl], W = p.W[lvl];

  // centroid_lvl = coords / 2**i (exact), coords_lvl = centroid + delta (one f32 add)   blocks.py:326-328
  const float inv = 1.0f / (float)(1 << lvl);
  if (lane < 14) {
    const int axis = lane / 7, k = lane - axis * 7;
    const float cc = __fmul_rn(p.coords[(long)sn * 2 + axis], inv);
    const float v = __fadd_rn(cc, (float)(k - 3));
    const CtkTap t = axis == 0 ? ctk_tap(v, W, p.sx[lvl]) : ctk_tap(v, H, p.sy[lvl]);
    if (axis == 0) { tab.x0[k] = t.i0; tab.x1[k] = t.i1; tab.wx0[k] = t.w0; tab.wx1[k] = t.w1; }
    else { tab.y0[k] = t.i0; tab.y1[k] = t.i1; tab.wy0[k] = t.w0; tab.wy1[k] = t.w1; }
  }
  const f32x4 f = *reinterpret_cast<const f32x4*>(p.targets + (long)sn * CTK_C + c * 4);
  __syncthreads();

  // footprint = bounding box of all tap corners (indices are monotone in the tap number)
  int xb = tab.x0[0], xe = tab.x1[0], yb = tab.y0[0], ye = tab.y1[0];
#pragma unroll
  for (int k = 1; k < 7; ++k) {
    xb = min(xb, tab.x0[k]); xe = max(xe, tab.x1[k]);
    yb = min(yb, tab.y0[k]); ye = max(ye, tab.y1[k]);
  }
  const int fw = min(xe - xb + 1, FPMAX), fh = min(ye - yb + 1, FPMAX);
  const int npix = fw * fh;
  const float* fm = p.fm[lvl] + (long)s * H * W * CTK_C + c * 4;

  for (int j0 = 0; j0 < npix; j0 += 16) {  // 8 pixel pairs in flight
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pix = min(j0 + 2 * j + half, npix - 1);
      const int py = pix / fw, px = pix - py * fw;
      v[j] = *reinterpret_cast<const f32x4*>(fm + ((long)(yb + py) * W + (xb + px)) * CTK_C);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = v[j][0] * f[0] + v[j][1] * f[1] + v[j][2] * f[2] + v[j][3] * f[3];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);  // stays inside the 32-lane half
      const int pix = j0 + 2 * j + half;
      if (c == 0 && pix < npix) tab.c[pix] = __fdiv_rn(d, p.sqrt_c);  // corrs / sqrt(C)   blocks.py:361
    }
  }
  __syncthreads();

  if (lane < CTK_TAPS) {
    const int a = lane / 7, b = lane - a * 7;  // a: x offset index, b: y offset index (delta = (dy[a], dx[b]) added to (x, y))
    const int x0 = min(tab.x0[a] - xb, fw - 1), x1 = min(tab.x1[a] - xb, fw - 1);
    const int y0 = min(tab.y0[b] - yb, fh - 1), y1 = min(tab.y1[b] - yb, fh - 1);
    const float wx0 = tab.wx0[a], wx1 = tab.wx1[a], wy0 = tab.wy0[b], wy1 = tab.wy1[b];
    // ATen grid_sampler_2d: nw = s*e, ne = s*w, sw = n*e, se = n*w; interpolated = nw_val*nw + ne_val*ne + sw_val*sw + se_val*se
    float o = __fmul_rn(tab.c[y0 * fw + x0], __fmul_rn(wy0, wx0));
    o = __fmaf_rn(tab.c[y0 * fw + x1], __fmul_rn(wy0, wx1), o);
    o = __fmaf_rn(tab.c[y1 * fw + x0], __fmul_rn(wy1, wx0), o);
    o = __fmaf_rn(tab.c[y1 * fw + x1], __fmul_rn(wy1, wx1), o);
    p.out[((long)n * p.S + s) * (CTK_LEVELS * CTK_TAPS) + lvl * CTK_TAPS + lane] = o;
  }
}

}  // namespace

extern "C" int ctk_corrblock_sample(const float* const* fmaps, const int32_t* H, const int32_t* W, int32_t S, int32_t N,
                                    const float* targets, const float* coords, float* out, void* stream) {
  if (!fmaps || !H || !W || !targets || !coords || !out) return CTK_E_NULL;
  if (S <= 0 || N <= 0 || (long)S * N > 2000000000L) return CTK_E_SHAPE;
  CorrBlockP p;
  for (int l = 0; l < CTK_LEVELS; ++l) {
    if (!fmaps[l]) return CTK_E_NULL;
    if (H[l] <= 0 || W[l] <= 0) return CTK_E_SHAPE;
    if (!ctk_aligned16(fmaps[l])) return CTK_E_ALIGN;
    p.fm[l] = fmaps[l];
    p.H[l] = H[l];
    p.W[l] = W[l];
    p.sx[l] = ctk_sampler_scale(W[l]);
    p.sy[l] = ctk_sampler_scale(H[l]);
  }
  if (!ctk_aligned16(targets)) return CTK_E_ALIGN;
  p.targets = targets;
  p.coords = coords;
  p.out = out;
  p.S = S;
  p.N = N;
  p.sqrt_c = sqrtf((float)CTK_C);  // torch.sqrt(torch.tensor(C).float())
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double units = (double)S * N * CTK_LEVELS;
  CtkProfScope ps("corrblock_sample", units * 2.0 * 64 * CTK_C, units * (64.0 * CTK_C * 4 + CTK_C + 49.0 * 4), s);
  hipLaunchKernelGGL(corrblock_sample_kernel, dim3((unsigned)((long)S * N)), dim3(256), 0, s, p);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
