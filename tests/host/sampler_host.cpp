// Host build of co-tracker_amd/csrc/sampler_math.h (test infrastructure): the SAME arithmetic the device kernel of
// csrc/sampler.hip runs, looped on the CPU, so tests/test_sampler_math_host.py can compare it bit for bit with
// torch.nn.functional.grid_sample without a GPU.  Build: g++ -O2 -ffp-contract=off -shared -fPIC.
#include <stdint.h>
#include "../../co-tracker_amd/csrc/sampler_math.h"

extern "C" int host_bilinear_sampler(const float* in, int B, int C, int D, int H, int W, const float* coords, long P, int align,
                                     int border, float* out) {
  const long HW = (long)H * W;
  const float sx = ctk_sm_prescale(W, align), sy = ctk_sm_prescale(H, align), sz = D > 0 ? ctk_sm_prescale(D, align) : 0.0f;
  for (int b = 0; b < B; ++b)
    for (long pt = 0; pt < P; ++pt) {
      if (D == 0) {
        const float* cp = coords + ((long)b * P + pt) * 2;
        const CtkAxis x = ctk_axis_vector(cp[0], W, sx, align, border), y = ctk_axis_vector(cp[1], H, sy, align, border);
        const long o = (long)y.i0 * W + x.i0;
        for (int c = 0; c < C; ++c) {
          const float* base = in + ((long)b * C + c) * HW;
          const float nw = (x.in0 && y.in0) ? base[o] : 0.0f, ne = (x.in1 && y.in0) ? base[o + 1] : 0.0f;
          const float sw = (x.in0 && y.in1) ? base[o + W] : 0.0f, se = (x.in1 && y.in1) ? base[o + W + 1] : 0.0f;
          out[((long)b * C + c) * P + pt] = ctk_blend2(nw, ne, sw, se, x, y);
        }
      } else {
        const float* cp = coords + ((long)b * P + pt) * 3;
        const CtkAxis z = ctk_axis_scalar(cp[0], D, sz, align, border), x = ctk_axis_scalar(cp[1], W, sx, align, border),
                      y = ctk_axis_scalar(cp[2], H, sy, align, border);
        const long o0 = ((long)z.i0 * H + y.i0) * W + x.i0;
        for (int c = 0; c < C; ++c) {
          const float* base = in + ((long)b * C + c) * HW * D;
          out[((long)b * C + c) * P + pt] =
              ctk_blend3(x, y, z, [&](int dz, int dy, int dx) { return base[o0 + (long)dz * HW + (long)dy * W + dx]; });
        }
      }
    }
  return 0;
}
