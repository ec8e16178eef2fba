// fp32 MFMA GEMM with fused epilogues:  C = act(A @ W^T + bias + bias_rows) + resid
//
// Every Linear on the hot path (corr_mlp cotracker3_online.py:84/205, input_transform
// cotracker.py:484, to_q/to_kv/to_out blocks.py:375-377, mlp.fc1/fc2 blocks.py:61-67) runs
// through this kernel.  Parity (1e-3 px / 1e-4 logit) rules out bf16/f16 inputs,
// so the contraction uses v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate, 157 TF peak.
//
// Tiling (wave64, 4 waves as 2x2): block tile (64*MR) x (64*NR), BK = 32; each wave owns
// MR x NR accumulators of 32x32.  A and W are both K-contiguous ("B^T" layout = torch Linear),
// staged global -> VGPR -> LDS with a 36-float row pitch so that the ds_read_b128 fragment
// reads (16-lane groups of rows distinct mod 16) are bank-conflict free.  A lane's float4
// covers four MFMA k-steps: the k index fed to step (j,e) by lane-half h is 8j+4h+e for
// both operands, a permutation of the K order that leaves the dot product unchanged.
// One barrier per K-tile (double-buffered LDS); global loads for tile k+1 are issued
// before the MFMAs of tile k.
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"

namespace {

constexpr int BK = 32;
constexpr int PITCH = BK + 4;  // floats

template <int MR, int NR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(CtkGemmP g) {
  constexpr int BM = 64 * MR, BN = 64 * NR;
  constexpr int A_LD4 = BM / 32, W_LD4 = BN / 32;  // float4 loads per thread per K-tile
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * PITCH];

  const unsigned nblk = gridDim.x;
  unsigned tile = ctk_xcd_remap(blockIdx.x, nblk);
  const int nb = tile % g.nblocks;
  tile /= g.nblocks;
  const int mb = tile % g.mblocks;
  const int bz = tile / g.mblocks;

  const float* A = static_cast<const float*>(g.A) + (long)bz * g.a_bs;
  float* C = static_cast<float*>(g.C) + (long)bz * g.c_bs;
  const int m0 = mb * BM, n0 = nb * BN;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, half = lane >> 5;

  // staging assignment: thread -> (row r + 32*i, float4 column kq)
  const int lr = tid >> 3, kq = tid & 7;
  const float* a_src[A_LD4];
  const float* w_src[W_LD4];
#pragma unroll
  for (int i = 0; i < A_LD4; ++i) {
    int row = min(m0 + lr + 32 * i, g.M - 1);  // clamp: rows >= M are never stored
    a_src[i] = A + (long)row * g.lda + kq * 4;
  }
#pragma unroll
  for (int i = 0; i < W_LD4; ++i) w_src[i] = g.W + (long)(n0 + lr + 32 * i) * g.ldw + kq * 4;

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  f32x4 sa[A_LD4], sw[W_LD4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) sa[i] = *reinterpret_cast<const f32x4*>(a_src[i] + kt * BK);
#pragma unroll
    for (int i = 0; i < W_LD4; ++i) sw[i] = *reinterpret_cast<const f32x4*>(w_src[i] + kt * BK);
  };
  auto lstore = [&](int buf) {
    float* la = lds[buf];
    float* lw = lds[buf] + BM * PITCH;
#pragma unroll
    for (int i = 0; i < A_LD4; ++i) *reinterpret_cast<f32x4*>(la + (lr + 32 * i) * PITCH + kq * 4) = sa[i];
#pragma unroll
    for (int i = 0; i < W_LD4; ++i) *reinterpret_cast<f32x4*>(lw + (lr + 32 * i) * PITCH + kq * 4) = sw[i];
  };

  const int KT = g.K / BK;
  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const float* la = lds[buf] + (wm * 32 * MR + r32) * PITCH + half * 4;
    const float* lw = lds[buf] + BM * PITCH + (wn * 32 * NR + r32) * PITCH + half * 4;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      f32x4 fa[MR], fb[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) fa[i] = *reinterpret_cast<const f32x4*>(la + i * 32 * PITCH + j * 8);
#pragma unroll
      for (int i = 0; i < NR; ++i) fb[i] = *reinterpret_cast<const f32x4*>(lw + i * 32 * PITCH + j * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < MR; ++mi)
#pragma unroll
          for (int ni = 0; ni < NR; ++ni)
            // operands swapped on purpose: D' = W_tile . A_tile^T, so a lane ends up with 4 CONSECUTIVE
            // output columns (n) of one output row (m) per register quad -> float4 epilogue I/O
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[ni][e], fa[mi][e], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue.  With the swapped operands the 32x32 accumulator holds D'[n][m]:
  //   m (output row)    = lane & 31
  //   n (output column) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // so registers 4q..4q+3 are output columns 8q + 4*half + 0..3 of row m: one float4 per quad.
  // Residual values (which may alias C: in-place "x += f(x)") are loaded up front, all at once.
  const bool has_res = g.resid != nullptr;
  const float* Rz = has_res ? g.resid + (long)bz * g.c_bs : nullptr;
  f32x4 res[MR][NR][4];
#pragma unroll
  for (int mi = 0; mi < MR; ++mi) {
    const int row = min(m0 + wm * 32 * MR + mi * 32 + r32, g.M - 1);
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 32 * NR + ni * 32 + q * 8 + half * 4;
        res[mi][ni][q] = has_res ? *reinterpret_cast<const f32x4*>(Rz + (long)row * g.ldr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
#pragma unroll
  for (int mi = 0; mi < MR; ++mi) {
    const int row = m0 + wm * 32 * MR + mi * 32 + r32;
    const float* brow = g.bias_rows ? g.bias_rows + (long)(min(row, g.M - 1) % g.bias_period) * g.N : nullptr;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + wn * 32 * NR + ni * 32 + q * 8 + half * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][q * 4 + e];
        if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
        if (brow) v += *reinterpret_cast<const f32x4*>(brow + col);
        if (g.act == CTK_ACT_GELU_ERF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_erf(v[e]);
        } else if (g.act == CTK_ACT_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctk_gelu_tanh(v[e]);
        }
        v += res[mi][ni][q];
        if (row < g.M) *reinterpret_cast<f32x4*>(C + (long)row * g.ldc + col) = v;
      }
    }
  }
}

}  // namespace

extern "C" int ctk_gemm(const ctk_gemm_args* a, void* stream) {
  if (!a || !a->A || (!a->W && !a->Wp) || !a->C) return CTK_E_NULL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->N % 64) || (a->K % BK)) return CTK_E_SHAPE;
  if ((a->lda % 4) || !ctk_aligned16(a->A)) return CTK_E_ALIGN;
  if (a->Wp ? !ctk_aligned16(a->Wp) : ((a->ldw % 4) || !ctk_aligned16(a->W))) return CTK_E_ALIGN;
  if (a->bias_rows && a->bias_period <= 0) return CTK_E_SHAPE;
  const int batch = a->batch > 0 ? a->batch : 1;
  if (batch > 1 && ((a->a_bs % 4) || (a->c_bs % 4))) return CTK_E_ALIGN;
  if ((a->ldc % 4) || !ctk_aligned16(a->C) || (a->resid && ((a->ldr % 4) || !ctk_aligned16(a->resid)))) return CTK_E_ALIGN;
  if ((a->bias && !ctk_aligned16(a->bias)) || (a->bias_rows && !ctk_aligned16(a->bias_rows))) return CTK_E_ALIGN;
  CtkGemmP g;
  g.A = a->A; g.lda = a->lda; g.M = a->M;
  g.W = a->W; g.ldw = a->ldw; g.N = a->N; g.K = a->K;
  g.Wp = static_cast<const unsigned short*>(a->Wp);
  g.C = a->C; g.ldc = a->ldc;
  g.bias = a->bias; g.bias_rows = a->bias_rows; g.bias_period = a->bias_period;
  g.resid = a->resid; g.ldr = a->ldr; g.act = a->act;
  g.batch = batch; g.a_bs = a->a_bs; g.c_bs = a->c_bs;
  g.a_split = a->a_split; g.c_split = a->c_split;
  if ((g.a_split || g.c_split) && !g.Wp) return CTK_E_SHAPE;  // SH operands exist only on the split-half back end
  if (g.c_split && (g.resid || (a->ldc % 64))) return CTK_E_SHAPE;
  if (g.a_split && (a->lda % 64)) return CTK_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double kv = (double)(a->k_valid > 0 ? a->k_valid : a->K);
  const double flops = 2.0 * a->M * (double)a->N * kv * batch;
  const double bytes = 4.0 * batch * ((double)a->M * kv + (double)a->M * a->N * (a->resid ? 2.0 : 1.0)) + 4.0 * a->N * kv;
  if (g.Wp) return ctk_launch_gemm_f16x3(g, flops, bytes, s);  // split-half MFMA back end (gemm_f16x3.hip)
  // 128x128 tiles when they fill the chip, 64x64 tiles for the small (virtual-track) GEMMs.
  const long big_blocks = (long)((a->M + 127) / 128) * (a->N / 128) * batch;
  if ((a->N % 128) == 0 && big_blocks >= 384) {
    g.mblocks = (a->M + 127) / 128; g.nblocks = a->N / 128;
    CtkProfScope ps("gemm_f32_128x128", flops, bytes, s);
    hipLaunchKernelGGL((gemm_f32_kernel<2, 2>), dim3((unsigned)big_blocks), dim3(256), 0, s, g);
  } else {
    g.mblocks = (a->M + 63) / 64; g.nblocks = a->N / 64;
    const long blocks = (long)g.mblocks * g.nblocks * batch;
    CtkProfScope ps("gemm_f32_64x64", flops, bytes, s);
    hipLaunchKernelGGL((gemm_f32_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, s, g);
  }
  CTK_HIP_CHECK_LAUNCH();
  return CTK_OK;
}
