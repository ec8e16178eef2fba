// Dev tool (round 5): does a ds_write2_b32 issued right behind a packed FP32 op store a stale first data register?
//   hipcc -O2 --offload-arch=gfx950 tools/pk_hazard_repro.cpp -o tools/pk_hazard_repro && tools/pk_hazard_repro
// The sampler's blend hit this with  v_pk_fma_f32 vD, vA, vW, vC op_sel:[0,1,0]  +  ds_write2_b32 vaddr, vD.lo, vD.hi  (profiles/
// r05_sampler_v3_pk_hazard.txt).  Each wave repeats that pair with fixed registers (inline asm), reads the two dwords back
// and compares with fmaf; a stale register shows as the PREVIOUS iteration's value.  Variants: 0 = op_sel:[0,1,0] (low lane reads
// the high half of the weight pair), 1 = op_sel_hi:[1,0,1] (both lanes read the low half), 2 = plain operands with a (w, w) pair;
// `gap` = number of s_nop 0 between the two instructions; `busy` = other waves of the workgroup hammer the LDS meanwhile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR, int GAP>
__global__ __launch_bounds__(256) void repro(unsigned* bad_lane_hist, unsigned* bad_elem, int iters, int busy) {
  __shared__ float lds[256 * 2 + 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* mine = lds + tid * 2;
  float* junk = lds + 512 + wave * 1024;
  const unsigned addr = (unsigned)(size_t)mine;  // LDS byte address (low 32 bits of the generic pointer are the offset)
  unsigned bad0 = 0, bad1 = 0;
  if (busy && wave != 0) {  // LDS traffic from the other waves
    float acc = 0.0f;
    for (int it = 0; it < iters * 4; ++it) {
      junk[(lane * 17 + it) & 1023] = acc;
      acc += junk[(lane * 5 + it * 3) & 1023];
    }
    if (acc == 12345.678f) bad_elem[2] = 1;
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const float a0 = (float)(lane + 1) + 0.25f * it, a1 = (float)(2 * lane + 3) - 0.5f * it;
    const float w0 = 0.5f + 0.001f * (it & 15), w1 = 0.25f + 0.002f * (it & 7);
    const float c0 = 1000.0f + it, c1 = -500.0f - it;
    const float wsel = (VAR == 0) ? w1 : w0;  // the weight both result lanes use
    const float e0 = fmaf(a0, wsel, c0), e1 = fmaf(a1, wsel, c1);
    float p0 = a0, p1 = a1, q0 = (VAR == 2) ? wsel : w0, q1 = (VAR == 2) ? wsel : w1, r0 = c0, r1 = c1;
    asm volatile(
        "v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\t"
        "v_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
        "v_mov_b32 v14, %5\n\tv_mov_b32 v15, %6\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 7\n\t"
        ".if %7 == 0\n\tv_pk_fma_f32 v[16:17], v[10:11], v[12:13], v[14:15] op_sel:[0,1,0]\n\t.endif\n\t"
        ".if %7 == 1\n\tv_pk_fma_f32 v[16:17], v[10:11], v[12:13], v[14:15] op_sel_hi:[1,0,1]\n\t.endif\n\t"
        ".if %7 == 2\n\tv_pk_fma_f32 v[16:17], v[10:11], v[12:13], v[14:15]\n\t.endif\n\t"
        ".rept %8\n\ts_nop 0\n\t.endr\n\t"
        "ds_write2_b32 %0, v16, v17 offset1:1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        :
        : "v"(addr), "v"(p0), "v"(p1), "v"(q0), "v"(q1), "v"(r0), "v"(r1), "n"(VAR), "n"(GAP)
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "memory");
    const float g0 = mine[0], g1 = mine[1];
    if (g0 != e0) { ++bad0; atomicAdd(&bad_lane_hist[lane], 1u); }
    if (g1 != e1) { ++bad1; atomicAdd(&bad_lane_hist[64 + lane], 1u); }
  }
  if (bad0) atomicAdd(&bad_elem[0], bad0);
  if (bad1) atomicAdd(&bad_elem[1], bad1);
}


// The sequence of the sampler's blend as hipcc emitted it (one j step): four ds_read_b128 of the corner vectors, counted waits,
// four DEPENDENT packed ops accumulating in place (w00 * a, + b * w10, + c * w01, + d * w11) and the ds_write2_b32 of the result.
// FLAVOR 0 = the failing copy's operand forms (weights packed as (w00, w10), (w01, w11): op_sel:[0,1,0] on the 2nd and 4th op),
// FLAVOR 1 = the never-failing copy's (one register per weight, op_sel_hi only).  Lanes with lane % 5 == 4 are masked off (EXEC).
// CTX bit 0: eight global_load_dwordx4 are in flight (landing in OTHER registers) while the chain runs, as the next frame's A
// fragments are in the sampler; bit 1: sixteen MFMAs (other registers) are issued right in front of the chain.
template <int FLAVOR, int CTX>
__global__ __launch_bounds__(256) void repro_chain(unsigned* bad_lane_hist, unsigned* bad_elem, int iters, int busy, const float* gsrc) {
  __shared__ float lds[256 * 16 + 256 * 2 + 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* rows = lds + tid * 16;               // a, b, c, d (4 floats each)
  float* mine = lds + 256 * 16 + tid * 2;     // the two outputs
  float* junk = lds + 256 * 18 + wave * 1024;
  const unsigned raddr = (unsigned)(size_t)rows, waddr = (unsigned)(size_t)mine;
  unsigned bad0 = 0, bad1 = 0;
  if (busy && wave != 0) {
    float acc = 0.0f;
    for (int it = 0; it < iters * 6; ++it) {
      junk[(lane * 17 + it) & 1023] = acc;
      acc += junk[(lane * 5 + it * 3) & 1023];
    }
    if (acc == 12345.678f) bad_elem[2] = 1;
    return;
  }
  if (lane % 5 == 4) return;  // partial EXEC, as in the blend
  for (int it = 0; it < iters; ++it) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (float)((lane * 7 + k * 13 + it * 3) % 97) * 0.03125f - 1.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) rows[k] = v[k];
    const float w00 = 0.25f + 0.001f * (it & 31), w10 = 0.75f - 0.001f * (it & 31), w01 = 0.125f + 0.002f * (it & 15), w11 = 0.5f - 0.002f * (it & 15);
    const float e0 = fmaf(v[12], w11, fmaf(v[8], w01, fmaf(v[4], w10, v[0] * w00)));
    const float e1 = fmaf(v[13], w11, fmaf(v[9], w01, fmaf(v[5], w10, v[1] * w00)));
    mine[0] = -7.0f;
    mine[1] = -7.0f;
    const float* gp = gsrc + ((size_t)blockIdx.x * 256 + tid) * 4 + (size_t)(it & 63) * 2048 * 256 * 4;
    if (CTX & 1) {
      asm volatile(
          "global_load_dwordx4 v[100:103], %0, off\n\tglobal_load_dwordx4 v[104:107], %0, off offset:16\n\t"
          "global_load_dwordx4 v[108:111], %0, off offset:32\n\tglobal_load_dwordx4 v[112:115], %0, off offset:48\n\t"
          "global_load_dwordx4 v[116:119], %0, off offset:64\n\tglobal_load_dwordx4 v[120:123], %0, off offset:80\n\t"
          "global_load_dwordx4 v[124:127], %0, off offset:96\n\tglobal_load_dwordx4 v[128:131], %0, off offset:112\n\t"
          :
          : "v"(gp)
          : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",
            "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "memory");
    }
    if (CTX & 2) {
      asm volatile(
          ".rept 16\n\tv_mfma_f32_16x16x32_f16 v[140:143], v[132:135], v[136:139], v[140:143]\n\t.endr\n\t"
          :
          :
          : "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "memory");
    }
    if (FLAVOR == 2) {  // the (e1, e2) pair exactly as hipcc emitted it in the failing build: register shuffles, in-place src0, crossed op_sel
      const float x0 = fmaf(v[13], w11, fmaf(v[9], w01, fmaf(v[5], w10, v[1] * w00)));
      const float x1 = fmaf(v[14], w11, fmaf(v[10], w01, fmaf(v[6], w10, v[2] * w00)));
      asm volatile(
          "v_mov_b32 v40, %2\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %3\n\t"
          "v_mov_b32 v44, %4\n\tv_mov_b32 v45, %4\n\tv_mov_b32 v46, %5\n\tv_mov_b32 v47, %5\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read_b128 v[20:23], %1\n\tds_read_b128 v[24:27], %1 offset:16\n\tds_read_b128 v[28:31], %1 offset:32\n\tds_read_b128 v[32:35], %1 offset:48\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mov_b32 v50, v21\n\tv_mov_b32 v51, v22\n\t"
          "v_pk_mul_f32 v[52:53], v[40:41], v[50:51] op_sel:[1,0] op_sel_hi:[0,1]\n\t"
          "v_mov_b32 v24, v25\n\tv_mov_b32 v25, v26\n\t"
          "v_pk_fma_f32 v[24:25], v[24:25], v[42:43], v[52:53] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
          "v_mov_b32 v28, v29\n\tv_mov_b32 v29, v30\n\t"
          "v_pk_fma_f32 v[28:29], v[28:29], v[44:45], v[24:25] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
          "v_mov_b32 v54, v33\n\tv_mov_b32 v55, v34\n\t"
          "v_pk_fma_f32 v[34:35], v[54:55], v[46:47], v[28:29] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
          "v_add_u32 v33, 0, %0\n\t"
          "ds_write2_b32 v33, v34, v35 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          :
          : "v"(waddr), "v"(raddr), "v"(w00), "v"(w10), "v"(w01), "v"(w11)
          : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v40", "v41", "v42", "v43", "v44", "v45",
            "v46", "v47", "v50", "v51", "v52", "v53", "v54", "v55", "memory");
      if (CTX & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float h0 = mine[0], h1 = mine[1];
      if (h0 != x0) { ++bad0; atomicAdd(&bad_lane_hist[lane], 1u); }
      if (h1 != x1) { ++bad1; atomicAdd(&bad_lane_hist[64 + lane], 1u); }
      continue;
    }
    if (FLAVOR == 0) {
      asm volatile(
          "v_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\tv_mov_b32 v14, %4\n\tv_mov_b32 v15, %5\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read_b128 v[20:23], %1\n\tds_read_b128 v[24:27], %1 offset:16\n\tds_read_b128 v[28:31], %1 offset:32\n\tds_read_b128 v[32:35], %1 offset:48\n\t"
          "s_waitcnt lgkmcnt(3)\n\t"
          "v_pk_mul_f32 v[16:17], v[12:13], v[20:21] op_sel_hi:[0,1]\n\t"
          "s_waitcnt lgkmcnt(2)\n\t"
          "v_pk_fma_f32 v[16:17], v[24:25], v[12:13], v[16:17] op_sel:[0,1,0]\n\t"
          "s_waitcnt lgkmcnt(1)\n\t"
          "v_pk_fma_f32 v[16:17], v[28:29], v[14:15], v[16:17] op_sel_hi:[1,0,1]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_pk_fma_f32 v[16:17], v[32:33], v[14:15], v[16:17] op_sel:[0,1,0]\n\t"
          "ds_write2_b32 %0, v16, v17 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          :
          : "v"(waddr), "v"(raddr), "v"(w00), "v"(w10), "v"(w01), "v"(w11)
          : "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "memory");
    } else {
      asm volatile(
          "v_mov_b32 v12, %2\n\tv_mov_b32 v36, %3\n\tv_mov_b32 v14, %4\n\tv_mov_b32 v38, %5\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "ds_read_b128 v[20:23], %1\n\tds_read_b128 v[24:27], %1 offset:16\n\tds_read_b128 v[28:31], %1 offset:32\n\tds_read_b128 v[32:35], %1 offset:48\n\t"
          "s_waitcnt lgkmcnt(3)\n\t"
          "v_pk_mul_f32 v[16:17], v[12:13], v[20:21] op_sel_hi:[0,1]\n\t"
          "s_waitcnt lgkmcnt(2)\n\t"
          "v_pk_fma_f32 v[16:17], v[24:25], v[36:37], v[16:17] op_sel_hi:[1,0,1]\n\t"
          "s_waitcnt lgkmcnt(1)\n\t"
          "v_pk_fma_f32 v[16:17], v[28:29], v[14:15], v[16:17] op_sel_hi:[1,0,1]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_pk_fma_f32 v[16:17], v[32:33], v[38:39], v[16:17] op_sel_hi:[1,0,1]\n\t"
          "ds_write2_b32 %0, v16, v17 offset1:1\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          :
          : "v"(waddr), "v"(raddr), "v"(w00), "v"(w10), "v"(w01), "v"(w11)
          : "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "memory");
    }
    if (CTX & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float g0 = mine[0], g1 = mine[1];
    if (g0 != e0) { ++bad0; atomicAdd(&bad_lane_hist[lane], 1u); }
    if (g1 != e1) { ++bad1; atomicAdd(&bad_lane_hist[64 + lane], 1u); }
  }
  if (bad0) atomicAdd(&bad_elem[0], bad0);
  if (bad1) atomicAdd(&bad_elem[1], bad1);
}

template <int VAR, int GAP>
void run(const char* tag, int busy) {
  unsigned *hist, *elem;
  CHECK(hipMalloc(&hist, 128 * 4));
  CHECK(hipMalloc(&elem, 4 * 4));
  CHECK(hipMemset(hist, 0, 128 * 4));
  CHECK(hipMemset(elem, 0, 4 * 4));
  const int iters = 2000, blocks = 2048;
  static float* gsrc = nullptr;
  if (!gsrc) { CHECK(hipMalloc(&gsrc, (size_t)64 * 2048 * 256 * 16 + 4096)); CHECK(hipMemset(gsrc, 0, (size_t)64 * 2048 * 256 * 16 + 4096)); }
  if (VAR >= 10) hipLaunchKernelGGL((repro_chain<(VAR >= 20 ? 2 : (VAR - 10) % 2), (VAR >= 20 ? VAR - 20 : (VAR - 10) / 2)>), dim3(blocks), dim3(256), 0, 0, hist, elem, iters, busy, (const float*)gsrc);
  else hipLaunchKernelGGL((repro<(VAR < 10 ? VAR : 0), GAP>), dim3(blocks), dim3(256), 0, 0, hist, elem, iters, busy);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned> h(128), el(4);
  CHECK(hipMemcpy(h.data(), hist, 128 * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(el.data(), elem, 4 * 4, hipMemcpyDeviceToHost));
  const double checks = (double)iters * blocks * (busy ? 64 : 256);
  printf("%-34s busy %d: wrong first register %u, wrong second register %u of %.3g checks each", tag, busy, el[0], el[1], checks);
  if (el[0] | el[1]) {
    printf("; lanes with a wrong FIRST register:");
    for (int l = 0; l < 64; ++l) if (h[l]) printf(" %d:%u", l, h[l]);
    printf("; wrong SECOND:");
    for (int l = 0; l < 64; ++l) if (h[64 + l]) printf(" %d:%u", l, h[64 + l]);
  }
  printf("\n");
  CHECK(hipFree(hist));
  CHECK(hipFree(elem));
}

int main() {
  for (int busy = 0; busy < 2; ++busy) {
    run<0, 0>("op_sel:[0,1,0], no gap", busy);
    run<0, 1>("op_sel:[0,1,0], 1 s_nop", busy);
    run<0, 4>("op_sel:[0,1,0], 4 s_nop", busy);
    run<1, 0>("op_sel_hi:[1,0,1], no gap", busy);
    run<2, 0>("plain (w, w) pair, no gap", busy);
    run<10, 0>("blend chain, failing copy's forms", busy);
    run<11, 0>("blend chain, op_sel_hi-only forms", busy);
    run<12, 0>("chain (failing forms) + loads in flight", busy);
    run<14, 0>("chain (failing forms) + MFMAs in front", busy);
    run<16, 0>("chain (failing forms) + loads + MFMAs", busy);
    run<17, 0>("chain (op_sel_hi forms) + loads + MFMAs", busy);
    run<20, 0>("exact (e1, e2) sequence of the failing build", busy);
    run<23, 0>("exact (e1, e2) sequence + loads + MFMAs", busy);
  }
  return 0;
}
