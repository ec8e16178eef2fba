// Launch parameters shared by the two GEMM back ends (gemm.hip: exact f32 MFMA; gemm_f16x3.hip: split-half MFMA).
#pragma once
#include "ctk_common.h"

struct CtkGemmP {
  const float* A; long lda; int M;
  const float* W; long ldw; int N; int K;
  const unsigned short* Wp;   // packed split-half weights (gemm_f16x3.hip) or null
  float* C; long ldc;
  const float* bias;
  const float* bias_rows; int bias_period;
  const float* resid; long ldr;
  int act;
  int batch; long a_bs; long c_bs;
  int mblocks, nblocks;
};

// gemm_f16x3.hip
int ctk_launch_gemm_f16x3(CtkGemmP& g, double flops, double bytes, hipStream_t s);
