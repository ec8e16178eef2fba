// Arithmetic of bilinear_sampler (model_utils.py:191-255) + ATen's CPU grid_sample, restated ONCE for the device kernel
// (sampler.hip) and for a host build of the same code (tests/test_sampler_math_host.py compiles this header with g++ and
// compares it, bit for bit, with torch.nn.functional.grid_sample on the CPU -- no GPU needed to pin the arithmetic).
//
// Two pipelines, because ATen has two:
//   * 5-D input  -> grid_sampler_3d_cpu_impl (aten/src/ATen/native/GridSampler.cpp, scalar code built WITHOUT FMA):
//       u = align ? ((g + 1) / 2) * (size - 1) : ((g + 1) * size - 1) / 2;  border: clip to [0, size-1];
//       i0 = floor(u);  w0 = (i0 + 1) - u;  w1 = u - i0;  weight of a corner = (wx * wy) * wz;
//       out = 0; out += v * w for the in-range corners in the order tnw, tne, tsw, tse, bnw, bne, bsw, bse (mul, then add).
//   * 4-D input  -> the vectorised kernel (aten/src/ATen/native/cpu/GridSamplerKernel.cpp, built WITH FMA, and the
//     compiler contracts a*b+c): u = align ? (g + 1) * ((size - 1) / 2) : fma(g + 1, size / 2, -0.5);  same clip;
//       w = u - floor(u);  e = 1 - w;  weights nw = s*e, ne = s*w, sw = n*e, se = n*w (y weight first);
//       out = fma(se_v, se, fma(sw_v, sw, fma(ne_v, ne, nw_v * nw)));  out-of-range corners read as 0;
//       with border padding the west / north corners are always in range, east / south are tested against the size.
//   Both: g = c * f32(2 / max(size - 1, 1))  (align)  or  c * f32(2 / size)  (not), then g - 1, each separately rounded
//   (model_utils.py:242-251: a float32 tensor multiply, then `coords -= 1`).
// Which contractions ATen's build has was established empirically against torch 2.10 CPU (tools/probe_grid_sample_fma.py).
// Every operation below is an explicitly rounded single operation: compile with -ffp-contract=off.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define CTK_SM_HD __host__ __device__ __forceinline__
#else
#define CTK_SM_HD static inline
#endif

struct CtkAxis {
  int i0;        // floor index (may be outside [0, size) with zeros padding)
  float w0, w1;  // weights of i0 and i0 + 1
  bool in0, in1; // whether i0 / i0 + 1 are inside [0, size)
};

CTK_SM_HD float ctk_sm_prescale(int size, int align) {  // python double -> float32 tensor element
  return align ? (float)(2.0 / (double)(size - 1 > 1 ? size - 1 : 1)) : (float)(2.0 / (double)size);
}

// grid_sampler_3d_cpu_impl's per-axis arithmetic (5-D input)
CTK_SM_HD CtkAxis ctk_axis_scalar(float c, int size, float prescale, int align, int border) {
  float g = c * prescale;
  g = g - 1.0f;
  float u;
  if (align) {
    u = ((g + 1.0f) / 2.0f) * (float)(size - 1);
  } else {
    u = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
  }
  if (border) u = fminf((float)(size - 1), fmaxf(u, 0.0f));
  const float f = floorf(u);
  CtkAxis a;
  a.i0 = (int)f;
  a.w0 = (f + 1.0f) - u;
  a.w1 = u - f;
  a.in0 = a.i0 >= 0 && a.i0 < size;
  a.in1 = a.i0 + 1 >= 0 && a.i0 + 1 < size;
  return a;
}

// the vectorised 2-D kernel's per-axis arithmetic (4-D input)
CTK_SM_HD CtkAxis ctk_axis_vector(float c, int size, float prescale, int align, int border) {
  float g = c * prescale;
  g = g - 1.0f;
  float u;
  if (align) {
    u = (g + 1.0f) * ((float)(size - 1) / 2.0f);
  } else {
    u = fmaf(g + 1.0f, (float)size / 2.0f, -0.5f);
  }
  if (border) u = fminf((float)(size - 1), fmaxf(u, 0.0f));
  const float f = floorf(u);
  CtkAxis a;
  a.i0 = (int)f;
  a.w1 = u - f;
  a.w0 = 1.0f - a.w1;
  a.in0 = border ? true : (a.i0 > -1 && a.i0 < size);
  a.in1 = border ? (a.i0 + 1 < size) : (a.i0 + 1 > -1 && a.i0 + 1 < size);
  return a;
}

// 4-D blend: nw, ne, sw, se values (0 where out of range) -> FMA chain in ATen's order
CTK_SM_HD float ctk_blend2(float nw_v, float ne_v, float sw_v, float se_v, const CtkAxis& x, const CtkAxis& y) {
  const float nw = y.w0 * x.w0, ne = y.w0 * x.w1, sw = y.w1 * x.w0, se = y.w1 * x.w1;
  float o = nw_v * nw;
  o = fmaf(ne_v, ne, o);
  o = fmaf(sw_v, sw, o);
  o = fmaf(se_v, se, o);
  return o;
}

// 5-D blend: accumulate the in-range corners, multiply then add, corner order z-major / y / x, weight (wx * wy) * wz
template <typename Load>  // Load(dz, dy, dx) -> value of corner (z0 + dz, y0 + dy, x0 + dx); only called for in-range corners
CTK_SM_HD float ctk_blend3(const CtkAxis& x, const CtkAxis& y, const CtkAxis& z, Load load) {
  float o = 0.0f;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const bool ok = (dx ? x.in1 : x.in0) && (dy ? y.in1 : y.in0) && (dz ? z.in1 : z.in0);
        if (ok) {
          const float w = ((dx ? x.w1 : x.w0) * (dy ? y.w1 : y.w0)) * (dz ? z.w1 : z.w0);
          const float p = load(dz, dy, dx) * w;
          o = o + p;
        }
      }
  return o;
}
