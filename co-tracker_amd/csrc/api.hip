// C-ABI orchestration: one update iteration = corr_embed -> assemble_tokens -> update_former
// -> heads + state update, all enqueued on the caller's stream (no host sync, capturable).
#include "ctk_common.h"
#include "ctk_profile.h"
#include "gemm_params.h"
#include "ctk_options.h"
#include <cstdlib>
#include <mutex>
#include <new>

int ctk_launch_corr_volume(const ctk_window_args* a, int n0, int ncount, float* out, long level_stride, int ld,
                           hipStream_t s);
int ctk_launch_pyramid_split(const float* fmap, long pixels, void* out, int version, hipStream_t s);
int ctk_launch_corr_volume_sh(const ctk_window_args* a, const void* const* fm_sh, int n0, int ncount, void* out,
                              long level_stride_halves, int version, hipStream_t s);
int ctk_launch_virtual_init(const float* vt, int S, float* dst, hipStream_t s);
int ctk_launch_layernorm2(const float* x, void* y, long R, const float* gamma, const float* beta, float eps, void* y2, float eps2,
                          int out_split, hipStream_t s);
int ctk_launch_heads(const float* tokens, const float* hw, const float* hb, int S, int N, float* delta, float* coords,
                     float* vis, float* conf, hipStream_t s);

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

#define CTK_TRY(expr)        \
  do {                       \
    int rc__ = (expr);       \
    if (rc__) return rc__;   \
  } while (0)

// ---- two-stream overlap (ctk_window_args.aux_stream) --------------------------------------------------------
// Fork / join between the caller's stream and its auxiliary stream use timing-less events from a small ring.  An event
// may be re-recorded as soon as the hipStreamWaitEvent that consumes its previous record has been ENQUEUED (the wait
// binds to the record that precedes it), so a ring far longer than one fork/join sequence needs no host synchronisation.
// The events are host objects created on first use and kept for the life of the process (the second exception, after
// the profiler, to "no mutable global state"); both calls are legal during stream capture, where they become graph edges
// and pull the auxiliary stream into the capture.
// One ring per device (an event belongs to the device that was current when it was created: recording it on another
// device's stream is an invalid-handle error), creation errors are reported to the caller.
class EventRing {
 public:
  int next(hipEvent_t* out) {
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return (int)rc;
    if (dev < 0 || dev >= kMaxDev) return CTK_E_STATE;
    std::lock_guard<std::mutex> g(m_);
    PerDev& d = d_[dev];
    if (!d.ready) {
      for (int i = 0; i < kN; ++i) {
        rc = hipEventCreateWithFlags(&d.ev[i], hipEventDisableTiming);
        if (rc != hipSuccess) {
          for (int j = 0; j < i; ++j) (void)hipEventDestroy(d.ev[j]);
          return (int)rc;
        }
      }
      d.ready = true;
    }
    *out = d.ev[d.i];
    d.i = (d.i + 1) % kN;
    return CTK_OK;
  }

 private:
  static constexpr int kN = 256, kMaxDev = 64;
  struct PerDev {
    hipEvent_t ev[kN];
    int i = 0;
    bool ready = false;
  };
  std::mutex m_;
  PerDev d_[kMaxDev];
};
EventRing g_events;

// `to` waits for everything enqueued on `from` so far
int stream_follow(hipStream_t from, hipStream_t to) {
  hipEvent_t e;
  CTK_TRY(g_events.next(&e));
  hipError_t rc = hipEventRecord(e, from);
  if (rc != hipSuccess) return (int)rc;
  rc = hipStreamWaitEvent(to, e, 0);
  return rc == hipSuccess ? CTK_OK : (int)rc;
}

// CTK_OPT_OVERLAP (ctk_set_option; initial value from CTK_OVERLAP when the library is loaded): 0 = ignore aux_stream (everything on
// the caller's stream), bit 0 = software pipeline sampler || corr_mlp, bit 1 = points<-virtual query projection beside the
// virtual-track chain.  DEFAULT 0: measured on MI355X at C3 (profiles/r02_overlap_and_time_attention_ab.txt) the sampler and the
// corr_mlp GEMM do NOT complement each other -- run side by side each slows down by more than the other gains (sampler 2.70 -> 4 x
// 1.21 ms, fc1 2.24 -> 4 x 0.77 ms per iteration; step 1528.5 -> 1547.3 ms) -- and the side query projection is worth 0.15 %
// (1526.1 ms), inside run-to-run noise.  Results are bit-identical in every mode (tests), so the code stays as an opt-in for
// other shapes.  Two more placements were measured in round 4 and removed in round 6 (both +-0: the time blocks' q projection
// beside their kv projection, profiles/r04_overlap_qkv_ab.txt; the side projection on a limited number of CUs beside the small
// launches of the virtual-track chain, profiles/r04_overlap8_trace.txt).
int overlap_mode() { return ctk_opt(CTK_OPT_OVERLAP); }

// Joins `aux` back into `main` when a fork is still open at scope exit (an error return between fork and join would
// otherwise leave the auxiliary stream unjoined -- inside ctk_window_graph_create: stuck in a broken capture).
struct JoinGuard {
  hipStream_t main, aux;
  bool open = false;
  int fork() {
    const int rc = stream_follow(main, aux);
    open = rc == CTK_OK;
    return rc;
  }
  int join() {
    open = false;
    return stream_follow(aux, main);
  }
  ~JoinGuard() {
    if (open) (void)stream_follow(aux, main);
  }
};

// A Linear's weight: torch-layout f32 and/or the ctk_pack_weight blob (preferred when present).
struct WRef {
  const float* w;
  const void* p;
};

// lda / ldc / a_bs / c_bs are given in f32 ELEMENTS of the logical matrix; an SH operand stores two halves
// per element, so its strides double (a row of K columns = 2K halves).
int gemm(const float* A, long lda, int M, WRef W, long ldw, int N, int K, float* C, long ldc, const float* bias,
         int act, const float* resid, long ldr, hipStream_t s, const float* bias_rows = nullptr, int period = 0,
         int batch = 1, long a_bs = 0, long c_bs = 0, int k_valid = 0, bool a_split = false, bool c_split = false) {
  ctk_gemm_args g;
  g.A = A; g.lda = lda; g.M = M; g.W = W.w; g.Wp = W.p; g.ldw = ldw; g.N = N; g.K = K; g.C = C; g.ldc = ldc;
  g.bias = bias; g.bias_rows = bias_rows; g.bias_period = period; g.resid = resid; g.ldr = ldr; g.act = act;
  g.batch = batch; g.a_bs = a_bs; g.c_bs = c_bs; g.k_valid = k_valid;
  g.a_split = a_split; g.c_split = c_split;
  if (a_split) { g.lda *= 2; g.a_bs *= 2; }
  if (c_split) { g.ldc *= 2; g.c_bs *= 2; }
  return ctk_gemm(&g, s);
}

bool split_mode(const ctk_model_weights* w) { return w->in_p != nullptr; }

// ---- update-former workspace carve -------------------------------------------------------
struct UfWs {
  float* tokens;  // [R,384]
  float* xn;      // [R,384]
  float* xn2;     // [N*S,384]  norm1(points) of the points<-virtual block, produced on the auxiliary stream
  float* qkv;     // [R,1152]
  float* att;     // [R,384]
  float* hid;     // [R,1536]
  float* partial; // attention split-K partials
  size_t bytes;
};

int v2p_splits(int N) {
  int s = (N + 1023) / 1024;  // ~1024 keys (32 tiles of 32, 8 per wave) per 4-wave workgroup of attention_q64_kernel
  if (s < 1) s = 1;
  if (s > 32) s = 32;
  return s;
}

UfWs carve_uf(int S, int N, void* base) {
  const size_t R = (size_t)(N + CTK_VIRT) * S;
  UfWs w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* r = reinterpret_cast<float*>(p + off);
    off += align256(nfloat * sizeof(float));
    return r;
  };
  w.tokens = take(R * CTK_HID);
  w.xn = take(R * CTK_HID);
  w.xn2 = take((size_t)N * S * CTK_HID);
  w.qkv = take(R * 3 * CTK_HID);
  w.att = take(R * CTK_HID);
  w.hid = take(R * CTK_MLP);
  w.partial = take((size_t)v2p_splits(N) * S * CTK_HEADS * CTK_VIRT * (CTK_HEAD_DIM + 2));
  w.bytes = off;
  return w;
}

int attn(const float* q, long q_ld, long q_bs, long q_is, const float* k, const float* v, long kv_ld, long kv_bs,
         long kv_is, float* out, long o_bs, long o_is, int nbatch, int n1, int n2, int splits, float* partial,
         hipStream_t s, bool o_split, const uint8_t* key_mask = nullptr, const uint8_t* query_mask = nullptr) {
  ctk_attn_args a;
  a.key_mask = key_mask; a.query_mask = query_mask;
  a.q = q; a.q_ld = q_ld; a.q_bs = q_bs; a.q_is = q_is;
  a.k = k; a.v = v; a.kv_ld = kv_ld; a.kv_bs = kv_bs; a.kv_is = kv_is;
  a.out = out; a.o_ld = o_split ? 2 * CTK_HID : CTK_HID; a.o_bs = o_bs; a.o_is = o_is; a.o_split = o_split;
  a.nbatch = nbatch; a.n1 = n1; a.n2 = n2; a.splits = splits; a.partial = partial;
  return ctk_attention(&a, s);
}

// residual MLP: x += fc2(gelu_tanh(fc1(LN(x))))  on rows [r0, r0+R)   (blocks.py:437 / cotracker.py:576)
// In split mode (sp) the GEMM inputs xn / att / hid live in SH format (same bytes, same row offsets).
int mlp_block(const UfWs& ws, long r0, long R, const ctk_block_weights& b, hipStream_t s, bool sp) {
  float* tok = ws.tokens + r0 * CTK_HID;
  float* xn = ws.xn + r0 * CTK_HID;
  float* hid = ws.hid + r0 * CTK_MLP;
  CTK_TRY(ctk_layernorm(tok, xn, R, nullptr, nullptr, 1e-6f, sp, s));
  CTK_TRY(gemm(xn, CTK_HID, (int)R, WRef{b.w1, b.w1_p}, CTK_HID, CTK_MLP, CTK_HID, hid, CTK_MLP, b.b1, CTK_ACT_GELU_TANH, nullptr, 0, s,
               nullptr, 0, 1, 0, 0, 0, sp, sp));
  CTK_TRY(gemm(hid, CTK_MLP, (int)R, WRef{b.w2, b.w2_p}, CTK_MLP, CTK_HID, CTK_MLP, tok, CTK_HID, b.b2, CTK_ACT_NONE, tok, CTK_HID, s,
               nullptr, 0, 1, 0, 0, 0, sp, false));
  return CTK_OK;
}

int check_block(const ctk_block_weights& b, bool cross) {
  if (!b.bq || !b.bkv || !b.bo || !b.b1 || !b.b2) return CTK_E_NULL;
  if ((!b.wq && !b.wq_p) || (!b.wkv && !b.wkv_p) || (!b.wo && !b.wo_p) || (!b.w1 && !b.w1_p) || (!b.w2 && !b.w2_p)) return CTK_E_NULL;
  if (cross && (!b.ctx_gamma || !b.ctx_beta)) return CTK_E_NULL;
  return CTK_OK;
}

// What run_transformer needs of a model: CoTracker3 (ctk_model_weights: 3 layers, no mask) and CoTracker2
// (ctk_former_weights: 6 layers, per-point attention mask) share the block structure.
struct FormerRef {
  int depth;
  const ctk_block_weights* time_blocks;
  const ctk_block_weights* virtual2point;
  const ctk_block_weights* virtual_self;
  const ctk_block_weights* point2virtual;
  const float* virtual_tokens;
  const uint8_t* point_mask;  // CoTracker2 attention_mask per point (cotracker.py:343-345) or null
  bool split;                 // xn / att / hid are SH-format (split-half back end)
  hipStream_t aux;            // optional second stream (ctk_window_args.aux_stream) or null
  bool space_attn = true;     // false: add_space_attn=False (cotracker.py:496-502): the three space blocks are skipped
};

FormerRef former_of(const ctk_model_weights* w) {
  return FormerRef{CTK_DEPTH, w->time_blocks, w->virtual2point, w->virtual_self, w->point2virtual, w->virtual_tokens, nullptr,
                   w->in_p != nullptr, nullptr};
}

// EfficientUpdateFormer.forward (cotracker.py:483-531) on tokens already holding the input
// projection in rows [0, N*S).
int run_transformer(int S, int N, const FormerRef& fr, const UfWs& ws, hipStream_t s) {
  const FormerRef* w = &fr;
  const bool sp = fr.split;
  const long P = (long)N * S;             // point rows
  const long V = (long)CTK_VIRT * S;      // virtual rows
  const long R = P + V;
  const long QL = 3 * CTK_HID;            // qkv leading dimension
  float* tok = ws.tokens;
  float* xn = ws.xn;
  float* qkv = ws.qkv;
  float* att = ws.att;
  CTK_TRY(ctk_launch_virtual_init(w->virtual_tokens, S, tok + P * CTK_HID, s));  // cotracker.py:487-488

  for (int i = 0; i < fr.depth; ++i) {
    // ---- time attention over S for every track (incl. virtual)      cotracker.py:494-497
    {
      const ctk_block_weights& b = w->time_blocks[i];
      CTK_TRY(ctk_layernorm(tok, xn, R, nullptr, nullptr, 1e-6f, sp, s));
      CTK_TRY(gemm(xn, CTK_HID, (int)R, WRef{b.wq, b.wq_p}, CTK_HID, CTK_HID, CTK_HID, qkv, QL, b.bq, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(gemm(xn, CTK_HID, (int)R, WRef{b.wkv, b.wkv_p}, CTK_HID, 2 * CTK_HID, CTK_HID, qkv + CTK_HID, QL, b.bkv, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(attn(qkv, QL, S, 1, qkv + CTK_HID, qkv + 2 * CTK_HID, QL, S, 1, att, S, 1, N + CTK_VIRT, S, S, 1, nullptr, s, sp));
      CTK_TRY(gemm(att, CTK_HID, (int)R, WRef{b.wo, b.wo_p}, CTK_HID, CTK_HID, CTK_HID, tok, CTK_HID, b.bo, CTK_ACT_NONE, tok, CTK_HID, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(mlp_block(ws, 0, R, b, s, sp));
    }
    // The points<-virtual block's query side -- norm1(points) and to_q(points) -- depends only on the point tokens the
    // time block just produced, not on the virtual tracks: with an auxiliary stream it runs BESIDE the virtual-track
    // chain below (virtual<-points attention, two 1024-row MLPs, virtual self attention: ~16 launches that occupy a
    // fraction of the chip), into its own xn2 buffer and the (otherwise unused) q columns of the point rows of qkv.
    if (!fr.space_attn) continue;
    // (bit-identical to the single-stream order: same launches, same inputs; measured useless at C3 -- the side stream's persistent
    // to_q kernel and the main stream's persistent to_kv kernel each want every CU's whole LDS and serialise)
    const bool side_q = fr.aux != nullptr && (overlap_mode() & 2) != 0;
    JoinGuard side{s, fr.aux};
    auto side_work = [&]() -> int {
      const ctk_block_weights& b = w->point2virtual[i];
      CTK_TRY(side.fork());
      CTK_TRY(ctk_layernorm(tok, ws.xn2, P, nullptr, nullptr, 1e-6f, sp, fr.aux));                                          // norm1(points)
      CTK_TRY(gemm(ws.xn2, CTK_HID, (int)P, WRef{b.wq, b.wq_p}, CTK_HID, CTK_HID, CTK_HID, qkv, QL, b.bq, CTK_ACT_NONE, nullptr, 0, fr.aux, nullptr, 0, 1, 0, 0, 0, sp, false));
      return CTK_OK;
    };
    if (side_q) CTK_TRY(side_work());
    // ---- virtual <- points cross attention                          cotracker.py:510-512
    {
      const ctk_block_weights& b = w->virtual2point[i];
      CTK_TRY(ctk_layernorm(tok + P * CTK_HID, xn + P * CTK_HID, V, nullptr, nullptr, 1e-6f, sp, s));   // norm1(virtual)
      // norm_context(points) -- and, from the same read of the point tokens (the virtual-track chain below does not touch them),
      // norm1(points) of this depth's points<-virtual block into xn2 (round 5: one pass, two norms)
      if (!side_q) CTK_TRY(ctk_launch_layernorm2(tok, xn, P, b.ctx_gamma, b.ctx_beta, 1e-5f, ws.xn2, 1e-6f, sp, s));
      else CTK_TRY(ctk_layernorm(tok, xn, P, b.ctx_gamma, b.ctx_beta, 1e-5f, sp, s));                   // norm_context(points)
      CTK_TRY(gemm(xn + P * CTK_HID, CTK_HID, (int)V, WRef{b.wq, b.wq_p}, CTK_HID, CTK_HID, CTK_HID, qkv + P * QL, QL, b.bq, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(gemm(xn, CTK_HID, (int)P, WRef{b.wkv, b.wkv_p}, CTK_HID, 2 * CTK_HID, CTK_HID, qkv + CTK_HID, QL, b.bkv, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      // batch = frame t; query i = virtual track (row P + i*S + t); key j = point (row j*S + t)
      CTK_TRY(attn(qkv + P * QL, QL, 1, S, qkv + CTK_HID, qkv + 2 * CTK_HID, QL, 1, S, att + P * CTK_HID, 1, S, S, CTK_VIRT, N,
                   v2p_splits(N), ws.partial, s, sp, fr.point_mask, nullptr));  // mask over KEYS (cotracker.py:566-569)
      CTK_TRY(gemm(att + P * CTK_HID, CTK_HID, (int)V, WRef{b.wo, b.wo_p}, CTK_HID, CTK_HID, CTK_HID, tok + P * CTK_HID, CTK_HID, b.bo, CTK_ACT_NONE,
                   tok + P * CTK_HID, CTK_HID, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(mlp_block(ws, P, V, b, s, sp));
    }
    // ---- virtual self attention (AttnBlock over 64 virtual tracks per frame)  cotracker.py:514
    {
      const ctk_block_weights& b = w->virtual_self[i];
      CTK_TRY(ctk_layernorm(tok + P * CTK_HID, xn + P * CTK_HID, V, nullptr, nullptr, 1e-6f, sp, s));
      CTK_TRY(gemm(xn + P * CTK_HID, CTK_HID, (int)V, WRef{b.wq, b.wq_p}, CTK_HID, CTK_HID, CTK_HID, qkv + P * QL, QL, b.bq, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(gemm(xn + P * CTK_HID, CTK_HID, (int)V, WRef{b.wkv, b.wkv_p}, CTK_HID, 2 * CTK_HID, CTK_HID, qkv + P * QL + CTK_HID, QL, b.bkv, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(attn(qkv + P * QL, QL, 1, S, qkv + P * QL + CTK_HID, qkv + P * QL + 2 * CTK_HID, QL, 1, S, att + P * CTK_HID, 1, S, S,
                   CTK_VIRT, CTK_VIRT, 1, nullptr, s, sp));
      CTK_TRY(gemm(att + P * CTK_HID, CTK_HID, (int)V, WRef{b.wo, b.wo_p}, CTK_HID, CTK_HID, CTK_HID, tok + P * CTK_HID, CTK_HID, b.bo, CTK_ACT_NONE,
                   tok + P * CTK_HID, CTK_HID, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(mlp_block(ws, P, V, b, s, sp));
    }
    // ---- points <- virtual cross attention                          cotracker.py:515-517
    {
      const ctk_block_weights& b = w->point2virtual[i];
      CTK_TRY(ctk_layernorm(tok + P * CTK_HID, xn + P * CTK_HID, V, b.ctx_gamma, b.ctx_beta, 1e-5f, sp, s));           // norm_context(virtual)
      if (!side_q) CTK_TRY(gemm(ws.xn2, CTK_HID, (int)P, WRef{b.wq, b.wq_p}, CTK_HID, CTK_HID, CTK_HID, qkv, QL, b.bq, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));  // xn2 = norm1(points), written beside norm_context(points) above
      else CTK_TRY(side.join());  // join: q(points) is ready
      CTK_TRY(gemm(xn + P * CTK_HID, CTK_HID, (int)V, WRef{b.wkv, b.wkv_p}, CTK_HID, 2 * CTK_HID, CTK_HID, qkv + P * QL + CTK_HID, QL, b.bkv, CTK_ACT_NONE, nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(attn(qkv, QL, 1, S, qkv + P * QL + CTK_HID, qkv + P * QL + 2 * CTK_HID, QL, 1, S, att, 1, S, S, N, CTK_VIRT, 1, nullptr, s, sp,
                   nullptr, fr.point_mask));  // mask over QUERIES (cotracker.py:561-564)
      CTK_TRY(gemm(att, CTK_HID, (int)P, WRef{b.wo, b.wo_p}, CTK_HID, CTK_HID, CTK_HID, tok, CTK_HID, b.bo, CTK_ACT_NONE, tok, CTK_HID, s, nullptr, 0, 1, 0, 0, 0, sp, false));
      CTK_TRY(mlp_block(ws, 0, P, b, s, sp));
    }
  }
  return CTK_OK;
}

int check_weights(const ctk_model_weights* w) {
  if (!w) return CTK_E_NULL;
  if ((!w->in_w && !w->in_p) || !w->in_bias_t || !w->virtual_tokens || !w->head_w || !w->head_b) return CTK_E_NULL;
  const bool sp = split_mode(w);  // packed blobs are all-or-nothing: the SH activation pipeline needs every Linear split
  auto bad = [sp](const ctk_block_weights& b) { return sp != (b.wq_p && b.wkv_p && b.wo_p && b.w1_p && b.w2_p) || (!sp && (b.wq_p || b.wkv_p || b.wo_p || b.w1_p || b.w2_p)); };
  if (sp != (w->corr_fc1_p && w->corr_fc2_p)) return CTK_E_NULL;
  for (int i = 0; i < CTK_DEPTH; ++i)
    if (bad(w->time_blocks[i]) || bad(w->virtual2point[i]) || bad(w->virtual_self[i]) || bad(w->point2virtual[i])) return CTK_E_NULL;
  for (int i = 0; i < CTK_DEPTH; ++i) {
    CTK_TRY(check_block(w->time_blocks[i], false));
    CTK_TRY(check_block(w->virtual2point[i], true));
    CTK_TRY(check_block(w->virtual_self[i], false));
    CTK_TRY(check_block(w->point2virtual[i], true));
  }
  return CTK_OK;
}

int input_projection(int S, int N, const float* x, bool x_split, const ctk_model_weights* w, const UfWs& ws, hipStream_t s) {
  // tokens = input_transform(x + time_emb)   (cotracker3_online.py:247, cotracker.py:484)
  return gemm(x, CTK_X_LD, N * S, WRef{w->in_w, w->in_p}, CTK_X_LD, CTK_HID, CTK_X_LD, ws.tokens, CTK_HID, nullptr, CTK_ACT_NONE, nullptr, 0,
              s, w->in_bias_t, S, 1, 0, 0, CTK_X_DIM, x_split, false);
}

// ---- corr_embed workspace -------------------------------------------------------------------
struct CorrWs {
  float* vol;  // [4][chunk*S][2432]   (SH format in split mode: same bytes)
  float* h1;   // [4*chunk*S][384]     (SH format in split mode)
  void* fm_sh[CTK_LEVELS];  // split mode: SH copy of the window's pyramid (scaled by 2^8), [S*H*W][4][2][32] halves
  size_t bytes;
  int chunk;
  int corr_version;  // CTK_OPT_CORR_VERSION as read ONCE per entry-point call: the layout of fm_sh and the sampler kernel must agree
};

int corr_chunk_points(const ctk_window_args* a) {
  int c = a->points_per_chunk > 0 ? a->points_per_chunk : a->N;
  if (c > a->N) c = a->N;
  return c;
}

CorrWs carve_corr(const ctk_window_args* a, void* base) {
  CorrWs w;
  w.corr_version = ctk_opt(CTK_OPT_CORR_VERSION);
  w.chunk = corr_chunk_points(a);
  const size_t rows = (size_t)w.chunk * a->S;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  w.vol = reinterpret_cast<float*>(p + off);
  off += align256(rows * CTK_LEVELS * CTK_CORR_LD * sizeof(float));
  w.h1 = reinterpret_cast<float*>(p + off);
  off += align256(rows * CTK_LEVELS * CTK_HID * sizeof(float));
  for (int l = 0; l < CTK_LEVELS; ++l) {  // always carved (the size query does not know the weights' mode): ~8 MB per frame
    w.fm_sh[l] = p + off;
    off += align256((size_t)a->S * (a->H[l] > 0 ? a->H[l] : 0) * (a->W[l] > 0 ? a->W[l] : 0) * CTK_C * sizeof(float));
  }
  w.bytes = off;
  return w;
}

// split mode, once per window: SH copy of the pyramid for the correlation sampler's footprint DMA
int prepare_pyramid_sh(const ctk_window_args* a, const CorrWs& ws, hipStream_t s) {
  for (int l = 0; l < CTK_LEVELS; ++l) {
    if (!a->fmaps[l]) return CTK_E_NULL;
    if (a->H[l] <= 0 || a->W[l] <= 0) return CTK_E_SHAPE;
    CTK_TRY(ctk_launch_pyramid_split(a->fmaps[l], (long)a->S * a->H[l] * a->W[l], ws.fm_sh[l], ws.corr_version, s));
  }
  return CTK_OK;
}

// x is f32 [N*S, CTK_X_LD] or, when x_split, the same matrix in SH format.  In split mode the hidden h1 is SH
// (fc1's epilogue writes it, fc2 streams it) and so is the correlation volume (corr_sh.hip); the caller has run
// prepare_pyramid_sh for this window.
int run_corr_embed(const ctk_window_args* a, const ctk_model_weights* w, float* x, bool x_split, const CorrWs& ws, hipStream_t s) {
  const bool sp = split_mode(w);
  if (x_split && !sp) return CTK_E_SHAPE;
  if ((!w->corr_fc1_w && !w->corr_fc1_p) || !w->corr_fc1_b || (!w->corr_fc2_w && !w->corr_fc2_p) || !w->corr_fc2_b) return CTK_E_NULL;
  hipStream_t aux = static_cast<hipStream_t>(a->aux_stream);
  const bool pipelined = sp && aux != nullptr && (overlap_mode() & 1) != 0;
  for (int n0 = 0; n0 < a->N; n0 += ws.chunk) {
    const int cnt = (a->N - n0 < ws.chunk) ? a->N - n0 : ws.chunk;
    // Software pipeline over point pieces: the sampler (VALU / LDS bound, MFMA pipe ~11 % busy) of piece j+1 runs on the
    // caller's stream while corr_mlp of piece j (MFMA bound) runs on the auxiliary stream; one workgroup of each kind
    // fits on a CU (77 KiB + 64 KiB of LDS).  Each piece has its own slice of the volume / hidden buffers.
    const int pieces = pipelined ? ((cnt >= 4096) ? 4 : (cnt >= 1024 ? 2 : 1)) : 1;
    const int per = (cnt + pieces - 1) / pieces;
    JoinGuard pipe{s, aux};
    for (int j = 0; j < pieces; ++j) {
      const int p0 = j * per;
      const int pc = (cnt - p0 < per) ? cnt - p0 : per;
      if (pc <= 0) break;
      const long rows = (long)pc * a->S;
      float* vol = ws.vol + (size_t)p0 * a->S * CTK_LEVELS * CTK_CORR_LD;
      float* h1 = ws.h1 + (size_t)p0 * a->S * CTK_LEVELS * CTK_HID;
      hipStream_t gs = s;
      if (sp) CTK_TRY(ctk_launch_corr_volume_sh(a, ws.fm_sh, n0 + p0, pc, vol, rows * CTK_CORR_LD * 2, ws.corr_version, s));
      else CTK_TRY(ctk_launch_corr_volume(a, n0 + p0, pc, vol, rows * CTK_CORR_LD, CTK_CORR_LD, s));
      if (pipelined && pieces > 1) {
        CTK_TRY(pipe.fork());
        gs = aux;
      }
      // corr_mlp.fc1 + exact GELU over all 4 levels at once        cotracker3_online.py:205, blocks.py:71-72
      CTK_TRY(gemm(vol, CTK_CORR_LD, (int)(rows * CTK_LEVELS), WRef{w->corr_fc1_w, w->corr_fc1_p}, CTK_CORR_LD, CTK_HID, CTK_CORR_LD, h1, CTK_HID,
                   w->corr_fc1_b, CTK_ACT_GELU_ERF, nullptr, 0, gs, nullptr, 0, 1, 0, 0, CTK_CORR_K, sp, sp));
      // corr_mlp.fc2, one batch per level, written into x[n*S+t][l*256 ...]   (torch.cat :209)
      CTK_TRY(gemm(h1, CTK_HID, (int)rows, WRef{w->corr_fc2_w, w->corr_fc2_p}, CTK_HID, 256, CTK_HID, x + (long)(n0 + p0) * a->S * CTK_X_LD + CTK_X_CORR,
                   CTK_X_LD, w->corr_fc2_b, CTK_ACT_NONE, nullptr, 0, gs, nullptr, 0, CTK_LEVELS, rows * CTK_HID, 256, 0, sp, x_split));
    }
    if (pipelined && pieces > 1) CTK_TRY(pipe.join());  // join before the next chunk reuses the buffers / x is consumed
  }
  return CTK_OK;
}

int check_window(const ctk_window_args* a) {
  if (!a) return CTK_E_NULL;
  if (a->S <= 0 || a->N <= 0 || a->iters < 0) return CTK_E_SHAPE;
  if ((long)(a->N + CTK_VIRT) * a->S > 2000000000L / CTK_MLP * 64) return CTK_E_SHAPE;
  if (a->flags & ~CTK_WINDOW_NO_SPACE_ATTN) return CTK_E_SHAPE;  // unknown flag bits: a caller built the pre-v6 struct (no `flags`)
  return CTK_OK;
}

}  // namespace

extern "C" int ctk_abi_version(void) { return CTK_ABI_VERSION; }

extern "C" const char* ctk_error_string(int code) {
  switch (code) {
    case CTK_OK: return "ok";
    case CTK_E_NULL: return "required pointer is NULL";
    case CTK_E_SHAPE: return "unsupported shape";
    case CTK_E_ALIGN: return "pointer or leading dimension not 16-byte aligned";
    case CTK_E_WORKSPACE: return "workspace too small";
    case CTK_E_STATE: return "call not allowed in the current state";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown error";
  }
}

extern "C" int ctk_update_former_workspace_bytes(int32_t S, int32_t N, size_t* out_bytes) {
  if (!out_bytes) return CTK_E_NULL;
  if (S <= 0 || N <= 0) return CTK_E_SHAPE;
  *out_bytes = carve_uf(S, N, nullptr).bytes;
  return CTK_OK;
}

extern "C" int ctk_update_former(int32_t S, int32_t N, const float* x, const ctk_model_weights* w, float* delta,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !delta || !workspace) return CTK_E_NULL;
  if (S <= 0 || N <= 0) return CTK_E_SHAPE;
  CTK_TRY(check_weights(w));
  if (!ctk_aligned16(workspace)) return CTK_E_ALIGN;
  const UfWs ws = carve_uf(S, N, workspace);
  if (ws.bytes > workspace_bytes) return CTK_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CTK_TRY(input_projection(S, N, x, false, w, ws, s));
  CTK_TRY(run_transformer(S, N, former_of(w), ws, s));
  return ctk_launch_heads(ws.tokens, w->head_w, w->head_b, S, N, delta, nullptr, nullptr, nullptr, s);
}

// ---- general update former (CoTracker2: 6 + 6 layers, 456 -> 130, attention mask) ---------------------------
extern "C" int ctk_update_former_ex(int32_t S, int32_t N, const void* x, int32_t x_split, const ctk_former_weights* w,
                                    const uint8_t* point_mask, float* delta, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (!x || !delta || !workspace || !w) return CTK_E_NULL;
  if (S <= 0 || N <= 0 || w->depth <= 0 || w->depth > 64) return CTK_E_SHAPE;
  if (w->in_ld <= 0 || (w->in_ld % 32) || w->out_ld <= 0 || (w->out_ld % 64)) return CTK_E_SHAPE;
  if ((!w->in_w && !w->in_p) || !w->virtual_tokens || (!w->head_w && !w->head_p) || !w->head_b) return CTK_E_NULL;
  if (!w->time_blocks || !w->virtual2point || !w->virtual_self || !w->point2virtual) return CTK_E_NULL;
  const bool sp = w->in_p != nullptr;
  if (x_split && !sp) return CTK_E_SHAPE;
  for (int i = 0; i < w->depth; ++i) {
    CTK_TRY(check_block(w->time_blocks[i], false));
    CTK_TRY(check_block(w->virtual2point[i], true));
    CTK_TRY(check_block(w->virtual_self[i], false));
    CTK_TRY(check_block(w->point2virtual[i], true));
    if (sp != (w->time_blocks[i].wq_p != nullptr)) return CTK_E_NULL;
  }
  if (!ctk_aligned16(workspace)) return CTK_E_ALIGN;
  const UfWs ws = carve_uf(S, N, workspace);
  if (ws.bytes > workspace_bytes) return CTK_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // tokens = input_transform(x) (+ per-frame bias rows = W e_t + b when in_bias_t is given, else + in_b)
  CTK_TRY(gemm(static_cast<const float*>(x), w->in_ld, N * S, WRef{w->in_w, w->in_p}, w->in_ld, CTK_HID, w->in_ld, ws.tokens, CTK_HID,
               w->in_bias_t ? nullptr : w->in_b, CTK_ACT_NONE, nullptr, 0, s, w->in_bias_t, S, 1, 0, 0, w->in_dim, x_split != 0, false));
  const FormerRef fr{w->depth, w->time_blocks, w->virtual2point, w->virtual_self, w->point2virtual, w->virtual_tokens, point_mask, sp, nullptr};
  CTK_TRY(run_transformer(S, N, fr, ws, s));
  // heads: delta[n*S+t][0..out_ld) = tokens @ head_w^T + head_b   (flow_head, cotracker.py:526)
  return gemm(ws.tokens, CTK_HID, N * S, WRef{w->head_w, w->head_p}, CTK_HID, w->out_ld, CTK_HID, delta, w->out_ld, w->head_b, CTK_ACT_NONE,
              nullptr, 0, s, nullptr, 0, 1, 0, 0, 0, false, false);
}

extern "C" int ctk_corr_embed_workspace_bytes(const ctk_window_args* a, size_t* out_bytes) {
  if (!out_bytes) return CTK_E_NULL;
  CTK_TRY(check_window(a));
  *out_bytes = carve_corr(a, nullptr).bytes;
  return CTK_OK;
}

extern "C" int ctk_corr_embed(const ctk_window_args* a, const ctk_model_weights* w, float* x, void* workspace,
                              size_t workspace_bytes, void* stream) {
  CTK_TRY(check_window(a));
  if (!w || !x || !workspace) return CTK_E_NULL;
  if (!ctk_aligned16(workspace) || !ctk_aligned16(x)) return CTK_E_ALIGN;
  const CorrWs ws = carve_corr(a, workspace);
  if (ws.bytes > workspace_bytes) return CTK_E_WORKSPACE;
  if (split_mode(w)) CTK_TRY(prepare_pyramid_sh(a, ws, static_cast<hipStream_t>(stream)));
  return run_corr_embed(a, w, x, false, ws, static_cast<hipStream_t>(stream));
}

extern "C" int ctk_corr_volume_sh_workspace_bytes(const ctk_window_args* a, size_t* out_bytes) {
  return ctk_corr_embed_workspace_bytes(a, out_bytes);
}

extern "C" int ctk_corr_volume_sh(const ctk_window_args* a, void* out, void* workspace, size_t workspace_bytes, void* stream) {
  CTK_TRY(check_window(a));
  if (!out || !workspace) return CTK_E_NULL;
  if (!ctk_aligned16(workspace) || !ctk_aligned16(out)) return CTK_E_ALIGN;
  const CorrWs ws = carve_corr(a, workspace);
  if (ws.bytes > workspace_bytes) return CTK_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  CTK_TRY(prepare_pyramid_sh(a, ws, s));
  return ctk_launch_corr_volume_sh(a, ws.fm_sh, 0, a->N, out, (long)a->N * a->S * CTK_CORR_LD * 2, ws.corr_version, s);
}

// Workspace of a whole window: x | update-former buffers | correlation buffers
extern "C" int ctk_forward_window_workspace_bytes(const ctk_window_args* a, size_t* out_bytes) {
  if (!out_bytes) return CTK_E_NULL;
  CTK_TRY(check_window(a));
  size_t total = align256((size_t)a->N * a->S * CTK_X_LD * sizeof(float));
  total += carve_uf(a->S, a->N, nullptr).bytes;
  total += carve_corr(a, nullptr).bytes;
  *out_bytes = total;
  return CTK_OK;
}

extern "C" int ctk_forward_window(const ctk_window_args* a, const ctk_model_weights* w, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  CTK_TRY(check_window(a));
  CTK_TRY(check_weights(w));
  if (!workspace || !a->coords || !a->vis || !a->conf) return CTK_E_NULL;
  if (!ctk_aligned16(workspace)) return CTK_E_ALIGN;
  size_t need = 0;
  CTK_TRY(ctk_forward_window_workspace_bytes(a, &need));
  if (need > workspace_bytes) return CTK_E_WORKSPACE;
  char* base = static_cast<char*>(workspace);
  float* x = reinterpret_cast<float*>(base);
  size_t off = align256((size_t)a->N * a->S * CTK_X_LD * sizeof(float));
  const UfWs uws = carve_uf(a->S, a->N, base + off);
  off += uws.bytes;
  const CorrWs cws = carve_corr(a, base + off);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool sp = split_mode(w);  // split mode: the transformer input x is kept in SH format
  if (sp && a->iters > 0) CTK_TRY(prepare_pyramid_sh(a, cws, s));
  for (int it = 0; it < a->iters; ++it) {                       // cotracker3_online.py:187
    CTK_TRY(run_corr_embed(a, w, x, sp, cws, s));               // :190-210
    CTK_TRY(ctk_assemble_tokens(a, x, sp, s));                  // :212-245
    CTK_TRY(input_projection(a->S, a->N, x, sp, w, uws, s));    // :247 + cotracker.py:484
    FormerRef fr = former_of(w);
    fr.aux = static_cast<hipStream_t>(a->aux_stream);
    fr.space_attn = (a->flags & CTK_WINDOW_NO_SPACE_ATTN) == 0;
    CTK_TRY(run_transformer(a->S, a->N, fr, uws, s));           // :250
    CTK_TRY(ctk_launch_heads(uws.tokens, w->head_w, w->head_b, a->S, a->N, nullptr, a->coords, a->vis, a->conf, s));  // :252-259
  }
  return CTK_OK;
}

// ---- hipGraph of one window (BASELINE.json configs[3]) ------------------------------------------------------
struct ctk_window_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int64_t nodes;
};

namespace {
// Capture `enqueue(stream)` on a private stream into an instantiated graph.
template <typename F>
int capture_graph(F enqueue, ctk_window_graph** out) {
  hipStream_t cs = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  if (e != hipSuccess) return (int)e;
  // thread-local mode: allocations made by other host threads (e.g. torch's caching allocator) stay legal
  e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(cs);
    return (int)e;
  }
  const int rc = enqueue(cs);
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(cs, &graph);
  (void)hipStreamDestroy(cs);
  if (rc != CTK_OK || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc != CTK_OK ? rc : (e != hipSuccess ? (int)e : (int)hipErrorUnknown);
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return (int)e;
  }
  size_t n = 0;
  (void)hipGraphGetNodes(graph, nullptr, &n);
  ctk_window_graph* g = new (std::nothrow) ctk_window_graph{graph, exec, (int64_t)n};
  if (!g) {
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    return (int)hipErrorOutOfMemory;
  }
  *out = g;
  return CTK_OK;
}
}  // namespace

extern "C" int ctk_window_graph_create(const ctk_window_args* a, const ctk_model_weights* w, void* workspace,
                                       size_t workspace_bytes, ctk_window_graph** out) {
  if (!out) return CTK_E_NULL;
  *out = nullptr;
  if (ctk_profile_is_on()) return CTK_E_STATE;
  // validate before touching the capture machinery (ctk_forward_window repeats these checks)
  CTK_TRY(check_window(a));
  CTK_TRY(check_weights(w));
  if (!workspace || !a->coords || !a->vis || !a->conf) return CTK_E_NULL;
  size_t need = 0;
  CTK_TRY(ctk_forward_window_workspace_bytes(a, &need));
  if (need > workspace_bytes) return CTK_E_WORKSPACE;
  return capture_graph([&](hipStream_t cs) { return ctk_forward_window(a, w, workspace, workspace_bytes, cs); }, out);
}

// ---- CoTracker2 window driver (cotracker.py:86-173): one capture-safe call per window --------------------------
namespace {
struct V2Ws {
  float* pos;     // [N,456]
  float* fcorrs;  // [N,S,196]
  float* x;       // [N*S,in_ld]  (f32 or SH: same bytes)
  float* delta;   // [N*S,out_ld]
  float* normed;  // [S*N,128]
  void* former;   // update-former workspace
  size_t former_bytes;
  size_t bytes;
};

V2Ws carve_v2(const ctk_v2_window_args* a, const ctk_v2_weights* w, void* base) {
  V2Ws r;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* q = reinterpret_cast<float*>(p + off);
    off += align256(nfloat * sizeof(float));
    return q;
  };
  const size_t rows = (size_t)a->N * a->S;
  r.pos = take((size_t)a->N * w->former.in_dim);
  r.fcorrs = take(rows * CTK_LEVELS * CTK_TAPS);
  r.x = take(rows * w->former.in_ld);
  r.delta = take(rows * w->former.out_ld);
  r.normed = take(rows * CTK_C);
  r.former = p + off;
  r.former_bytes = carve_uf(a->S, a->N, nullptr).bytes;
  off += r.former_bytes;
  r.bytes = off;
  return r;
}

int check_v2(const ctk_v2_window_args* a, const ctk_v2_weights* w) {
  if (!a || !w) return CTK_E_NULL;
  if (a->S <= 0 || a->N <= 0 || a->iters < 0) return CTK_E_SHAPE;
  if (w->former.in_dim != 456 || w->former.out_dim != CTK_C + 2 || w->former.in_ld < 456 || (w->former.in_ld % 32) ||
      w->former.out_ld < CTK_C + 2 || (w->former.out_ld % 64))
    return CTK_E_SHAPE;
  if (!a->coords || !a->track_feat || !a->vis || !a->track_mask || !a->vis_out) return CTK_E_NULL;
  if (!w->pos_hwc || !w->norm_w || !w->norm_b || (!w->upd_w && !w->upd_p) || !w->upd_b || !w->vis_w || !w->vis_b) return CTK_E_NULL;
  if (w->pos_h <= 0 || w->pos_w <= 0) return CTK_E_SHAPE;
  for (int l = 0; l < CTK_LEVELS; ++l) {
    if (!a->fmaps[l]) return CTK_E_NULL;
    if (a->H[l] <= 0 || a->W[l] <= 0) return CTK_E_SHAPE;
  }
  return CTK_OK;
}
}  // namespace

extern "C" int ctk_forward_window_v2_workspace_bytes(const ctk_v2_window_args* a, const ctk_v2_weights* w, size_t* out_bytes) {
  if (!out_bytes) return CTK_E_NULL;
  CTK_TRY(check_v2(a, w));
  *out_bytes = carve_v2(a, w, nullptr).bytes;
  return CTK_OK;
}

extern "C" int ctk_forward_window_v2(const ctk_v2_window_args* a, const ctk_v2_weights* w, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  CTK_TRY(check_v2(a, w));
  if (!workspace) return CTK_E_NULL;
  if (!ctk_aligned16(workspace)) return CTK_E_ALIGN;
  const V2Ws ws = carve_v2(a, w, workspace);
  if (ws.bytes > workspace_bytes) return CTK_E_WORKSPACE;
  const ctk_former_weights* fw = &w->former;
  const int x_split = fw->in_p != nullptr;
  const int S = a->S, N = a->N;
  // sampled_pos_emb: pos_emb at the FIRST frame's coordinates of the window (cotracker.py:126-130), fixed over the iterations
  CTK_TRY(ctk_sample_features4d(w->pos_hwc, w->pos_h, w->pos_w, fw->in_dim, a->coords, N, ws.pos, stream));
  for (int it = 0; it < a->iters; ++it) {                                                             // cotracker.py:132
    CTK_TRY(ctk_corrblock_sample(a->fmaps, a->H, a->W, S, N, a->track_feat, a->coords, ws.fcorrs, stream));   // :134-137
    CTK_TRY(ctk_v2_assemble(S, N, a->coords, ws.fcorrs, a->track_feat, a->track_mask, a->vis, ws.pos, fw->in_ld, ws.x, x_split,
                            stream));                                                                  // :139-150
    CTK_TRY(ctk_update_former_ex(S, N, ws.x, x_split, fw, a->point_mask, ws.delta, ws.former, ws.former_bytes, stream));  // :152-155
    CTK_TRY(ctk_v2_apply_delta(S, N, ws.delta, fw->out_ld, a->coords, w->norm_w, w->norm_b, 1e-5f, ws.normed, stream));  // :157-159,167
    // track_feat += GELU(Linear(GroupNorm(delta_feats)))   (track_feat_updater, cotracker.py:162-170), rows t*N+n
    CTK_TRY(gemm(ws.normed, CTK_C, S * N, WRef{w->upd_w, w->upd_p}, CTK_C, CTK_C, CTK_C, a->track_feat, CTK_C, w->upd_b, CTK_ACT_GELU_ERF,
                 a->track_feat, CTK_C, static_cast<hipStream_t>(stream)));
  }
  return ctk_v2_vis_head(a->track_feat, (int64_t)S * N, w->vis_w, w->vis_b, a->vis_out, stream);   // :172
}

extern "C" int ctk_v2_window_graph_create(const ctk_v2_window_args* a, const ctk_v2_weights* w, void* workspace,
                                          size_t workspace_bytes, ctk_window_graph** out) {
  if (!out) return CTK_E_NULL;
  *out = nullptr;
  if (ctk_profile_is_on()) return CTK_E_STATE;
  CTK_TRY(check_v2(a, w));
  if (!workspace) return CTK_E_NULL;
  if (carve_v2(a, w, nullptr).bytes > workspace_bytes) return CTK_E_WORKSPACE;
  return capture_graph([&](hipStream_t cs) { return ctk_forward_window_v2(a, w, workspace, workspace_bytes, cs); }, out);
}

extern "C" int ctk_window_graph_launch(ctk_window_graph* g, void* stream) {
  if (!g) return CTK_E_NULL;
  const hipError_t e = hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream));
  return e == hipSuccess ? CTK_OK : (int)e;
}

extern "C" int ctk_window_graph_nodes(const ctk_window_graph* g, int64_t* out_nodes) {
  if (!g || !out_nodes) return CTK_E_NULL;
  *out_nodes = g->nodes;
  return CTK_OK;
}

extern "C" int ctk_window_graph_destroy(ctk_window_graph* g) {
  if (!g) return CTK_OK;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return CTK_OK;
}
