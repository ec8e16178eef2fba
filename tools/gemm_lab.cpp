// gemm_lab: stand-alone A/B of the split-half GEMM kernels through the C-ABI (no Python, no torch: a fresh GPU box pays
// nothing for imports).  For every Linear shape of the update path it
//   1. runs gemm_f16x3.hip's kernels (ctk_gemm_pp_mode(0)) and gemm_pp.hip's persistent ping-pong kernels (mode 1),
//   2. checks BOTH against an fp64 host reference on sampled rows (incl. the last rows of a ragged M) and against each other,
//   3. re-runs the new kernel several times and demands bit-identical output (an LDS race shows up as nondeterminism),
//   4. times both, interleaved, with HIP events on the launch stream.
// Build (against the DEV library, `make -C co-tracker_amd/csrc dev`: the trace / clock / jitter / no-store modes exist only there):
//   hipcc -O2 -std=c++17 tools/gemm_lab.cpp -o tools/gemm_lab -Iinclude -Lco-tracker_amd -lctk_hip_dev -Wl,-rpath,'$ORIGIN/../co-tracker_amd'
// Usage:  tools/gemm_lab [quick]   (quick: small shapes only, for a smoke run)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../include/ctk.h"

#define HIP_OK(x)                                                                   \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)
#define CTK_OKAY(x)                                                                 \
  do {                                                                              \
    int r_ = (x);                                                                   \
    if (r_ != 0) {                                                                  \
      fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, r_, ctk_error_string(r_)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

extern "C" int ctk_debug_pp_trace(unsigned long long* host_out, int n);  // gemm_pp.hip dev entry (not in ctk.h)
extern "C" int ctk_debug_pp_clock(unsigned long long* host_out4);

struct Shape {
  const char* name;
  long M;
  int K, N;
  int act;
  bool res, split_out, brows, bias;
  int batch;  // corr_mlp.fc2: one batch per pyramid level, output columns interleaved (c_bs = N)
};

static float half_to_float(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  float v;
  if (e == 0) v = std::ldexp((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = std::ldexp((float)(m | 1024), (int)e - 25);
  return s ? -v : v;
}
static double gelu_erf(double x) { return 0.5 * x * (1.0 + std::erf(x * 0.70710678118654752440)); }
static double gelu_tanh(double x) { return 0.5 * x * (1.0 + std::tanh(0.79788456080286535588 * (x + 0.044715 * x * x * x))); }

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  // "exp m1 m2 ...": timing only, one column per ctk_gemm_pp_mode value (bit 0 = new kernels, bit 1 = no stores, bits 8.. = start stagger)
  const bool exp_mode = argc > 1 && (!strcmp(argv[1], "exp") || !strcmp(argv[1], "quant"));
  std::vector<int> exp_modes;
  for (int i = 2; exp_mode && i < argc; ++i) exp_modes.push_back(atoi(argv[i]));
  const int reps = quick ? 3 : 12;
  HIP_OK(hipSetDevice(0));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  printf("ctk abi %d\n", ctk_abi_version());
  // LAB_OLD / LAB_NEW = the ctk_gemm_pp_mode values of the "old" / "new" columns
  const int old_mode = getenv("LAB_OLD") ? atoi(getenv("LAB_OLD")) : 0;
  const int new_mode = getenv("LAB_NEW") ? atoi(getenv("LAB_NEW")) : 1;

  const bool dense = argc > 1 && (!strcmp(argv[1], "dense") || !strcmp(argv[1], "stress"));  // the shapes of tests/test_sharding.py's dense-mode run (S = 8, 3840 points)
  std::vector<Shape> shapes;
  if (dense) {
    shapes.push_back({"fc1  dense    ", 31232, 384, 1536, CTK_ACT_GELU_TANH, false, true, false, true, 1});
    shapes.push_back({"kv   dense    ", 30720, 384, 768, CTK_ACT_NONE, false, false, false, true, 1});
    shapes.push_back({"cfc2 dense    ", 30720, 384, 256, CTK_ACT_NONE, false, true, false, true, 4});
    shapes.push_back({"cfc1 dense    ", 122880, 2432, 384, CTK_ACT_GELU_ERF, false, true, false, true, 1});
    shapes.push_back({"fc2  dense    ", 31232 * 2, 1536, 384, CTK_ACT_NONE, true, false, false, true, 1});
  }
  // C3 sliding window (S = 16, N = 6400 points + 64 virtual tracks): rows = 103 424 (tokens) / 102 400 (points)
  const long RT = quick ? 6144 + 100 : 103424, RP = quick ? 6144 : 102400;
  if (!dense) {
  shapes.push_back({"mlp.fc1       ", RT, 384, 1536, CTK_ACT_GELU_TANH, false, true, false, true, 1});
  shapes.push_back({"mlp.fc2       ", RT, 1536, 384, CTK_ACT_NONE, true, false, false, true, 1});
  shapes.push_back({"to_q          ", RT, 384, 384, CTK_ACT_NONE, false, false, false, true, 1});
  shapes.push_back({"to_out        ", RT, 384, 384, CTK_ACT_NONE, true, false, false, true, 1});
  shapes.push_back({"to_kv         ", RP, 384, 768, CTK_ACT_NONE, false, false, false, true, 1});
  shapes.push_back({"input_transf  ", RP, 1120, 384, CTK_ACT_NONE, false, false, true, false, 1});
  shapes.push_back({"corr_mlp.fc2  ", RP, 384, 256, CTK_ACT_NONE, false, true, false, true, 4});
  shapes.push_back({"corr_mlp.fc1  ", quick ? RP : 4 * RP, 2432, 384, CTK_ACT_GELU_ERF, false, true, false, true, 1});
  }
  if (argc > 1 && !strcmp(argv[1], "quant")) {  // round quantisation: 768 / 800 / 1024 tiles of 256 x 192 on 256 CUs
    shapes.clear();
    for (long m : {98304l, 102400l, 131072l}) {
      shapes.push_back({"fc2  quant    ", m, 1536, 384, CTK_ACT_NONE, true, false, false, true, 1});
      shapes.push_back({"to_q quant    ", m, 384, 384, CTK_ACT_NONE, false, false, false, true, 1});
    }
  }
  if (argc > 1 && !strcmp(argv[1], "small")) {  // the virtual-track Linears (64 x S rows): 64 x 64 tile kernels
    shapes.clear();
    shapes.push_back({"v.q/out       ", 1024, 384, 384, CTK_ACT_NONE, true, false, false, true, 1});
    shapes.push_back({"v.kv          ", 1024, 384, 768, CTK_ACT_NONE, false, false, false, true, 1});
    shapes.push_back({"v.fc1         ", 1024, 384, 1536, CTK_ACT_GELU_TANH, false, true, false, true, 1});
    shapes.push_back({"v.fc2         ", 1024, 1536, 384, CTK_ACT_NONE, true, false, false, true, 1});
    shapes.push_back({"v.q S=120     ", 7680, 384, 384, CTK_ACT_NONE, true, false, false, true, 1});
  }
  if (!quick && !dense && !(argc > 1 && (!strcmp(argv[1], "small") || !strcmp(argv[1], "quant")))) {
    // C2 (offline S = 48, N = 400) and C4 (S = 16, N = 1024) token counts: few tiles per CU
    shapes.push_back({"fc1   @C2     ", 22272, 384, 1536, CTK_ACT_GELU_TANH, false, true, false, true, 1});
    shapes.push_back({"to_out@C2     ", 22272, 384, 384, CTK_ACT_NONE, true, false, false, true, 1});
    shapes.push_back({"fc1   @C4     ", 17408, 384, 1536, CTK_ACT_GELU_TANH, false, true, false, true, 1});
    shapes.push_back({"to_out@C4     ", 17408, 384, 384, CTK_ACT_NONE, true, false, false, true, 1});
    shapes.push_back({"fc2 ragged M  ", 103424 - 77, 1536, 384, CTK_ACT_NONE, true, false, false, true, 1});
    shapes.push_back({"kv  ragged M  ", 102400 - 200, 384, 768, CTK_ACT_NONE, false, false, false, true, 1});
  }

  if (argc > 1 && (!strcmp(argv[1], "trace") || !strcmp(argv[1], "clock")) && shapes.size() > 8) shapes.resize(8);
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  int failures = 0;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));

  for (const Shape& sh : shapes) {
    const long M = sh.M;
    const int K = sh.K, N = sh.N, B = sh.batch;
    const long a_elems = (long)B * M * K;
    const long ldc = (long)N * B;  // batches write interleaved column groups (as corr_mlp.fc2 does into x)
    const long c_elems = M * ldc;
    // host data (A is generated on the device-side-cheap way: a small random table tiled, rows decorrelated by a roll)
    std::vector<float> hW((size_t)N * K), hb(N), hbr;
    for (auto& v : hW) v = nd(rng) / std::sqrt((float)K);
    for (auto& v : hb) v = 0.1f * nd(rng);
    const int period = 16;
    if (sh.brows) {
      hbr.resize((size_t)period * N);
      for (auto& v : hbr) v = 0.1f * nd(rng);
    }
    const long TAB = 1 << 22;
    std::vector<float> tab(TAB);
    for (auto& v : tab) v = nd(rng);
    std::vector<float> hA((size_t)a_elems);
    for (long i = 0; i < a_elems; ++i) hA[i] = tab[(i * 2654435761ul + (i >> 22) * 40503ul) & (TAB - 1)];
    std::vector<float> hR;
    if (sh.res) {
      hR.resize((size_t)c_elems);
      for (long i = 0; i < c_elems; ++i) hR[i] = 3.0f * tab[(i * 11400714819323198485ul >> 20) & (TAB - 1)];
    }

    float *dA, *dAsh, *dW, *db = nullptr, *dbr = nullptr, *dC0, *dC1, *dC2;
    void* dWp;
    HIP_OK(hipMalloc(&dA, a_elems * 4));
    HIP_OK(hipMalloc(&dAsh, a_elems * 4));
    HIP_OK(hipMalloc(&dW, (size_t)N * K * 4));
    HIP_OK(hipMalloc(&dC0, c_elems * 4));
    HIP_OK(hipMalloc(&dC1, c_elems * 4));
    HIP_OK(hipMalloc(&dC2, c_elems * 4));
    size_t wp_bytes = 0;
    CTK_OKAY(ctk_pack_weight_bytes(N, K, &wp_bytes));
    HIP_OK(hipMalloc(&dWp, wp_bytes));
    HIP_OK(hipMemcpy(dA, hA.data(), a_elems * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dW, hW.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
    if (sh.bias) {
      HIP_OK(hipMalloc(&db, N * 4));
      HIP_OK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    }
    if (sh.brows) {
      HIP_OK(hipMalloc(&dbr, (size_t)period * N * 4));
      HIP_OK(hipMemcpy(dbr, hbr.data(), (size_t)period * N * 4, hipMemcpyHostToDevice));
    }
    CTK_OKAY(ctk_pack_weight(dW, K, N, K, dWp, st));
    CTK_OKAY(ctk_split_rows(dA, K, (long)B * M, K, dAsh, st));
    HIP_OK(hipStreamSynchronize(st));

    auto run = [&](float* dC, int mode) {
      ctk_gemm_pp_mode(mode);
      if (sh.res) HIP_OK(hipMemcpyAsync(dC, hR.data(), c_elems * 4, hipMemcpyHostToDevice, st));  // x += Linear(.)
      ctk_gemm_args g;
      memset(&g, 0, sizeof(g));
      g.A = dAsh; g.lda = 2 * K; g.M = (int)M;
      g.W = nullptr; g.ldw = K; g.N = N; g.K = K; g.Wp = dWp;
      g.C = dC; g.ldc = sh.split_out ? 2 * ldc : ldc;
      g.bias = db; g.bias_rows = dbr; g.bias_period = sh.brows ? period : 0;
      g.resid = sh.res ? dC : nullptr; g.ldr = ldc;
      g.act = sh.act;
      g.batch = B; g.a_bs = 2 * M * K; g.c_bs = sh.split_out ? 2 * N : N;
      g.k_valid = 0; g.a_split = 1; g.c_split = sh.split_out ? 1 : 0;
      CTK_OKAY(ctk_gemm(&g, st));
    };
    auto time_mode = [&](float* dC, int mode) {
      // residual shapes accumulate in place while timing (values drift, timing does not care)
      ctk_gemm_pp_mode(mode);
      ctk_gemm_args g;
      memset(&g, 0, sizeof(g));
      g.A = dAsh; g.lda = 2 * K; g.M = (int)M;
      g.ldw = K; g.N = N; g.K = K; g.Wp = dWp;
      g.C = dC; g.ldc = sh.split_out ? 2 * ldc : ldc;
      g.bias = db; g.bias_rows = dbr; g.bias_period = sh.brows ? period : 0;
      g.resid = sh.res ? dC : nullptr; g.ldr = ldc;
      g.act = sh.act;
      g.batch = B; g.a_bs = 2 * M * K; g.c_bs = sh.split_out ? 2 * N : N;
      g.a_split = 1; g.c_split = sh.split_out ? 1 : 0;
      CTK_OKAY(ctk_gemm(&g, st));  // warm
      HIP_OK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CTK_OKAY(ctk_gemm(&g, st));
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      return ms / reps;
    };

    if (argc > 1 && !strcmp(argv[1], "clock")) {
      // the shader clock the PRODUCTION kernel runs at, after `reps` back-to-back launches (workgroup 0's lifetime of the last one)
      const int mode = argc > 2 ? atoi(argv[2]) : 33;
      const double ms = time_mode(dC1, mode);
      HIP_OK(hipStreamSynchronize(st));
      unsigned long long c[4];
      CTK_OKAY(ctk_debug_pp_clock(c));
      const double ticks = (double)(c[2] - c[0]), real = (double)(c[3] - c[1]);
      printf("%s M=%7ld K=%4d N=%4d B=%d | mode %d: %8.1f us per launch | workgroup 0 alive %.1f us = %.0f cycles -> %.3f GHz\n", sh.name, M, K, N, B, mode,
             ms * 1e3, real / 100.0, ticks, real > 0 ? ticks / real * 0.1 : 0.0);
      fflush(stdout);
      HIP_OK(hipFree(dA)); HIP_OK(hipFree(dAsh)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dC0)); HIP_OK(hipFree(dC1)); HIP_OK(hipFree(dC2));
      HIP_OK(hipFree(dWp));
      if (db) HIP_OK(hipFree(db));
      if (dbr) HIP_OK(hipFree(dbr));
      continue;
    }
    if (argc > 1 && !strcmp(argv[1], "trace")) {
      // wave timeline of the DBG kernel (mode bit 6): s_memtime at the start of every MFMA phase + around every epilogue
      time_mode(dC1, 1);  // warm caches and clocks on the production kernel
      run(dC1, 65);
      HIP_OK(hipStreamSynchronize(st));
      const int WGS = 4, ST = 128;
      std::vector<unsigned long long> tr((size_t)WGS * 8 * ST);
      CTK_OKAY(ctk_debug_pp_trace(tr.data(), (int)tr.size()));
      const bool t256 = (N % 256) == 0;
      const int ph = t256 ? 4 : 3, KT = K / 32, per_tile = ph * KT + 2;
      printf("%s M=%ld K=%d N=%d: %d phases/K-tile, %d K-tiles, %d stamps per tile (cycles of s_memtime)\n", sh.name, M, K, N, ph, KT, per_tile);
      for (int wg = 0; wg < 2; ++wg)
        for (int wv : {0, 3, 4, 7}) {
          const unsigned long long* t = &tr[((size_t)wg * 8 + wv) * ST];
          int n = 0;
          while (n < ST - 4 && t[n]) ++n;
          const double ticks = (double)(t[ST - 2] - t[ST - 4]), real = (double)(t[ST - 1] - t[ST - 3]);  // s_memtime / s_memrealtime (100 MHz)
          printf("  wg %d wave %d: %d stamps; kernel life %.0f ticks = %.1f us -> s_memtime at %.3f GHz;", wg, wv, n, ticks, real / 100.0, real > 0 ? ticks / real * 0.1 : 0.0);
          for (int tile = 0; tile * per_tile + per_tile <= n; ++tile) {
            const unsigned long long* b = t + tile * per_tile;
            // phase-to-phase periods by phase index, first K-tile apart
            double sum[4] = {0, 0, 0, 0};
            int cnt[4] = {0, 0, 0, 0};
            for (int i = ph; i + 1 < ph * KT; ++i) {  // skip the first K-tile
              sum[i % ph] += (double)(b[i + 1] - b[i]);
              cnt[i % ph]++;
            }
            printf(" | tile %d: first K-tile %llu, period by phase", tile, b[ph] - b[0]);
            for (int p = 0; p < ph; ++p) printf(" %.0f", cnt[p] ? sum[p] / cnt[p] : 0.0);
            printf(", main loop %llu, last phase->epi %llu, epilogue %llu", b[ph * KT - 1] - b[0], b[ph * KT] - b[ph * KT - 1], b[ph * KT + 1] - b[ph * KT]);
            if (tile * per_tile + per_tile < n) printf(", epi end->next phase 0 %llu", b[per_tile] - b[ph * KT + 1]);
          }
          printf("\n");
        }
      fflush(stdout);
      HIP_OK(hipFree(dA)); HIP_OK(hipFree(dAsh)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dC0)); HIP_OK(hipFree(dC1)); HIP_OK(hipFree(dC2));
      HIP_OK(hipFree(dWp));
      if (db) HIP_OK(hipFree(db));
      if (dbr) HIP_OK(hipFree(dbr));
      continue;
    }
    if (exp_mode) {
      printf("%s M=%7ld K=%4d N=%4d B=%d |", sh.name, M, K, N, B);
      std::vector<double> best(exp_modes.size(), 1e30);
      for (int r = 0; r < 3; ++r)
        for (size_t i = 0; i < exp_modes.size(); ++i) best[i] = std::min(best[i], (double)time_mode(dC1, exp_modes[i]));
      for (size_t i = 0; i < exp_modes.size(); ++i) printf(" mode %5d: %8.1f us (%.3f) |", exp_modes[i], best[i] * 1e3, 2.0 * M * N * (double)K * B / best[i] / 1e9 / 833.3);
      printf("\n");
      fflush(stdout);
      HIP_OK(hipFree(dA)); HIP_OK(hipFree(dAsh)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dC0)); HIP_OK(hipFree(dC1)); HIP_OK(hipFree(dC2));
      HIP_OK(hipFree(dWp));
      if (db) HIP_OK(hipFree(db));
      if (dbr) HIP_OK(hipFree(dbr));
      continue;
    }
    if (argc > 1 && !strcmp(argv[1], "stress")) {  // repeated runs of one mode, every output compared bitwise with the first
      const int mode = argc > 2 ? atoi(argv[2]) : 1, iters = argc > 3 ? atoi(argv[3]) : 200;
      std::vector<float> ref((size_t)c_elems), cur((size_t)c_elems);
      int bad = 0;
      const int burst = argc > 4 ? atoi(argv[4]) : 1;  // launches per synchronisation (back-to-back on the stream)
      for (int it = 0; it < iters; ++it) {
        for (int b = 0; b < (it ? burst : 1); ++b) run(dC1, mode);
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipMemcpy(it ? cur.data() : ref.data(), dC1, c_elems * 4, hipMemcpyDeviceToHost));
        if (it && memcmp(ref.data(), cur.data(), c_elems * 4)) {
          long nbad = 0, first = -1;
          for (long i = 0; i < c_elems; ++i)
            if (memcmp(&ref[i], &cur[i], 4)) { if (first < 0) first = i; ++nbad; }
          if (bad < 5) printf("  iter %d: %ld words differ, first at row %ld col %ld\n", it, nbad, first / ldc, first % ldc);
          ++bad;
        }
      }
      printf("%s mode %d: %d of %d runs differ from the first\n", sh.name, mode, bad, iters - 1);
      fflush(stdout);
      failures += bad != 0;
      HIP_OK(hipFree(dA)); HIP_OK(hipFree(dAsh)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dC0)); HIP_OK(hipFree(dC1)); HIP_OK(hipFree(dC2));
      HIP_OK(hipFree(dWp));
      if (db) HIP_OK(hipFree(db));
      if (dbr) HIP_OK(hipFree(dbr));
      continue;
    }
    run(dC0, old_mode);
    run(dC1, new_mode);
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> c0((size_t)c_elems), c1((size_t)c_elems), c2((size_t)c_elems);
    HIP_OK(hipMemcpy(c0.data(), dC0, c_elems * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(c1.data(), dC1, c_elems * 4, hipMemcpyDeviceToHost));
    // determinism of the new kernel
    int nondet = 0;
    for (int r = 0; r < (quick ? 2 : 4); ++r) {
      run(dC2, new_mode);
      HIP_OK(hipStreamSynchronize(st));
      HIP_OK(hipMemcpy(c2.data(), dC2, c_elems * 4, hipMemcpyDeviceToHost));
      if (memcmp(c1.data(), c2.data(), c_elems * 4)) ++nondet;
    }
    // decode outputs (SH -> f32) and compare
    auto value = [&](const std::vector<float>& c, long m, long col) -> double {
      if (!sh.split_out) return c[(size_t)m * ldc + col];
      const uint16_t* h = reinterpret_cast<const uint16_t*>(c.data()) + (size_t)m * 2 * ldc + (col >> 5) * 64 + (col & 31);
      return (double)half_to_float(h[0]) + (double)half_to_float(h[32]);
    };
    double max_old_new = 0, max_ref_old = 0, max_ref_new = 0, max_mag = 0;
    long cmp = 0;
    {
      const long stride = std::max<long>(1, c_elems / 4000000);
      for (long i = 0; i < M * ldc; i += stride) {
        const long m = i / ldc, col = i % ldc;
        const double a = value(c0, m, col), b = value(c1, m, col);
        max_old_new = std::max(max_old_new, std::fabs(a - b));
        ++cmp;
      }
    }
    std::vector<long> rows;
    for (int i = 0; i < 48; ++i) rows.push_back((long)((i * 7919l * 104729l) % M));
    for (long m = std::max<long>(0, M - 40); m < M; ++m) rows.push_back(m);
    rows.push_back(0); rows.push_back(255); rows.push_back(256); rows.push_back(127); rows.push_back(128);
    for (long m : rows) {
      if (m >= M) continue;
      for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
          double acc = 0;
          const float* a = &hA[((size_t)b * M + m) * K];
          const float* w = &hW[(size_t)n * K];
          for (int k = 0; k < K; ++k) acc += (double)a[k] * (double)w[k];
          if (sh.bias) acc += hb[n];
          if (sh.brows) acc += hbr[(size_t)(m % period) * N + n];
          if (sh.act == CTK_ACT_GELU_ERF) acc = gelu_erf(acc);
          else if (sh.act == CTK_ACT_GELU_TANH) acc = gelu_tanh(acc);
          const long col = (long)b * N + n;
          if (sh.res) acc += hR[(size_t)m * ldc + col];
          max_mag = std::max(max_mag, std::fabs(acc));
          max_ref_old = std::max(max_ref_old, std::fabs(value(c0, m, col) - acc));
          max_ref_new = std::max(max_ref_new, std::fabs(value(c1, m, col) - acc));
        }
    }
    const double tol = 2e-5 * std::max(1.0, max_mag);
    const bool ok = max_ref_new <= tol && max_ref_old <= tol && nondet == 0;
    if (!ok) ++failures;

    // timing, interleaved old / new / old / new
    double t_old = 1e30, t_new = 1e30;
    for (int r = 0; r < 2; ++r) {
      t_old = std::min(t_old, (double)time_mode(dC0, old_mode));
      t_new = std::min(t_new, (double)time_mode(dC1, new_mode));
    }
    const double flops = 2.0 * M * N * (double)K * B;
    printf("%s M=%7ld K=%4d N=%4d B=%d | old %8.1f us %6.1f TF (%.3f) | new %8.1f us %6.1f TF (%.3f) | x%.2f | err vs fp64 old %.2e new %.2e (tol %.1e) old-new %.2e nondet %d %s\n",
           sh.name, M, K, N, B, t_old * 1e3, flops / t_old / 1e9, flops / t_old / 1e9 / 833.3, t_new * 1e3, flops / t_new / 1e9,
           flops / t_new / 1e9 / 833.3, t_old / t_new, max_ref_old, max_ref_new, tol, max_old_new, nondet, ok ? "OK" : "FAIL");
    fflush(stdout);
    HIP_OK(hipFree(dA)); HIP_OK(hipFree(dAsh)); HIP_OK(hipFree(dW)); HIP_OK(hipFree(dC0)); HIP_OK(hipFree(dC1)); HIP_OK(hipFree(dC2));
    HIP_OK(hipFree(dWp));
    if (db) HIP_OK(hipFree(db));
    if (dbr) HIP_OK(hipFree(dbr));
  }
  printf("%s\n", failures ? "GEMM LAB: FAILURES" : "GEMM LAB: all shapes OK");
  return failures ? 1 : 0;
}
