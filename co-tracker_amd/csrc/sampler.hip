// Op D of SURVEY 8b: the stand-alone bilinear_sampler (cotracker/models/core/model_utils.py:191-255) with the reference's
// full signature -- 4-D input [B,C,H,W] with coords [B,*,2] = (x, y), or 5-D input [B,C,T,H,W] with coords [B,*,3] =
// (t, x, y); align_corners in {True, False}; padding_mode in {"zeros", "border"} -- bit-identical to the reference's
// torch.nn.functional.grid_sample on the CPU (sampler_math.h holds the arithmetic and says how it was pinned).
// The hot path does not call this kernel (its samplers are fused into the correlation kernels and work on NHWC pyramids);
// this is the general operator for callers of model_utils.bilinear_sampler / sample_features4d / sample_features5d on the
// reference's own NCHW layout.
//
// HBM-bound gather.  One thread = one sample point of one batch element for a group of CG channels: the tap indices and
// weights are computed once per point, then each channel costs 4 (8) scattered 4-byte reads + one coalesced 4-byte write
// (neighbouring threads = neighbouring points: writes are contiguous in the [B,C,P] output, reads are as local as the
// coordinates are).  Algorithmic bytes per (point, channel): 4 or 8 taps x 4 B in + 4 B out.
#include "ctk_common.h"
#include "ctk_profile.h"
#include "sampler_math.h"

namespace {

constexpr int CG = 8;  // channels per thread pass (gridDim.y walks the channel groups)

struct SampP {
  const float* in;      // [B,C,(D,)H,W]
  const float* coords;  // [B,P,2|3]
  float* out;           // [B,C,P]
  int B, C, D, H, W;    // D == 0: 4-D input
  long P;
  int align, border;
  float sx, sy, sz;
};

__global__ __launch_bounds__(256) void bilinear_sampler_kernel(SampP p) {
  const long pt = (long)blockIdx.x * 256 + threadIdx.x;
  if (pt >= p.P) return;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CG, c1 = min(c0 + CG, p.C);
  const long HW = (long)p.H * p.W;
  if (p.D == 0) {
    const float* cp = p.coords + ((long)b * p.P + pt) * 2;
    const CtkAxis x = ctk_axis_vector(cp[0], p.W, p.sx, p.align, p.border);
    const CtkAxis y = ctk_axis_vector(cp[1], p.H, p.sy, p.align, p.border);
    const bool m_nw = x.in0 && y.in0, m_ne = x.in1 && y.in0, m_sw = x.in0 && y.in1, m_se = x.in1 && y.in1;
    const long o_nw = (long)y.i0 * p.W + x.i0;
    const float* base = p.in + ((long)b * p.C + c0) * HW;
    float* op = p.out + ((long)b * p.C + c0) * p.P + pt;
    for (int c = c0; c < c1; ++c, base += HW, op += p.P) {
      const float nw = m_nw ? base[o_nw] : 0.0f, ne = m_ne ? base[o_nw + 1] : 0.0f;
      const float sw = m_sw ? base[o_nw + p.W] : 0.0f, se = m_se ? base[o_nw + p.W + 1] : 0.0f;
      *op = ctk_blend2(nw, ne, sw, se, x, y);
    }
  } else {
    const float* cp = p.coords + ((long)b * p.P + pt) * 3;  // (t, x, y): model_utils.py:238-240 reorders to grid_sample's (x, y, t)
    const CtkAxis z = ctk_axis_scalar(cp[0], p.D, p.sz, p.align, p.border);
    const CtkAxis x = ctk_axis_scalar(cp[1], p.W, p.sx, p.align, p.border);
    const CtkAxis y = ctk_axis_scalar(cp[2], p.H, p.sy, p.align, p.border);
    const long DHW = HW * p.D;
    const long o0 = ((long)z.i0 * p.H + y.i0) * p.W + x.i0;
    const float* base = p.in + ((long)b * p.C + c0) * DHW;
    float* op = p.out + ((long)b * p.C + c0) * p.P + pt;
    for (int c = c0; c < c1; ++c, base += DHW, op += p.P) {
      *op = ctk_blend3(x, y, z, [&](int dz, int dy, int dx) { return base[o0 + (long)dz * HW + (long)dy * p.W + dx]; });
    }
  }
}

}  // namespace

extern "C" int ctk_bilinear_sampler(const float* input, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, const float* coords,
                                    int64_t P, int32_t align_corners, int32_t padding_mode, float* out, void* stream) {
  if (!input || !coords || !out) return CTK_E_NULL;
  if (B <= 0 || C <= 0 || D < 0 || H <= 0 || W <= 0 || P <= 0 || B > 65535) return CTK_E_SHAPE;
  if (padding_mode != CTK_PAD_ZEROS && padding_mode != CTK_PAD_BORDER) return CTK_E_SHAPE;  // "reflection" is not implemented
  if ((long)(D > 0 ? D : 1) * H * W > 2000000000L || (P + 255) / 256 > 16777215L) return CTK_E_SHAPE;  // gridDim.x * blockDim.x must stay below 2^32
  SampP p;
  p.in = input; p.coords = coords; p.out = out;
  p.B = B; p.C = C; p.D = D; p.H = H; p.W = W; p.P = P;
  p.align = align_corners != 0;
  p.border = padding_mode == CTK_PAD_BORDER;
  p.sx = ctk_sm_prescale(W, p.align);
  p.sy = ctk_sm_prescale(H, p.align);
  p.sz = D > 0 ? ctk_sm_prescale(D, p.align) : 0.0f;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double taps = D > 0 ? 8.0 : 4.0;
  CtkProfScope ps("bilinear_sampler", 0.0, (double)B * C * P * 4.0 * (taps + 1.0), s);
  const dim3 grid((unsigned)((P + 255) / 256), (unsigned)((C + CG - 1) / CG), (unsigned)B);
  if (grid.y > 65535) return CTK_E_SHAPE;
  hipLaunchKernelGGL(bilinear_sampler_kernel, grid, dim3(256), 0, s, p);
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
