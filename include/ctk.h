/*
 * ctk.h -- C-ABI of the MI355X (gfx950) CoTracker3 iterative-update hot path.
 *
 * The reference (facebookresearch/co-tracker @ 2025-03-04) is pure Python/PyTorch and
 * has no FFI; each entry point below replaces the reference interface cited beside
 * it (paths relative to the reference root), at the operator boundary fixed in
 * SURVEY.md section 8(b).  INTEGRATION.md shows the ctypes binding a maintainer
 * would add to cotracker/models/core/cotracker/cotracker3_online.py.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory (float32 unless
 *    stated) owned by the caller; the library never allocates, frees or retains
 *    device memory.  Its only process-wide state is (i) the option table of
 *    ctk_set_option (validated relaxed atomics: a launch uses what it reads when it is
 *    enqueued; no option changes what is computed), (ii) the opt-in bench recorder
 *    (ctk_profile_enable) and (iii) per-device caches of read-only queries (CU count,
 *    fork/join event rings); kernels write no device-side globals.  Entry points may be
 *    called concurrently from several host threads on different streams.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); no host
 *    synchronisation, no host reads of device data -> safe under stream capture.
 *  - return value: 0 ok, <0 invalid argument (CTK_E_*), >0 a hipError_t.
 *  - batch size B = 1 per call (every reference config has B = 1; the Python host
 *    loops over B).  C = 128 feature channels, hidden = 384, heads = 8 x 48,
 *    mlp = 1536, 64 virtual tracks, 4 pyramid levels, 7x7 taps -- the values fixed by
 *    cotracker/models/build_cotracker.py:31-38 and cotracker3_online.py:43-84.
 *
 * Data layout (ours, not the reference's)
 *  - feature pyramid level l: NHWC  [T, H_l, W_l, 128]   (reference: [B,T,128,H,W])
 *  - support patches   level l: [N, 49, 128]             (reference: [B,49,N,128])
 *  - window state: coords [S,N,2] (level-0 feature units), vis [S,N], conf [S,N] logits
 *  - transformer input x: [N*S, CTK_X_LD] row = n*S+t, columns
 *        [0,1024) corr embeddings (level-major), 1024 vis, 1025 conf,
 *        [1026,1110) posenc(84), [1110,1120) zero padding
 *    (reference order is [vis,conf,corr,posenc], cotracker3_online.py:212-245; the
 *    host permutes input_transform.weight columns once at load time)
 *  - tokens: [(N+64)*S, 384], row = n*S+t, virtual tracks are n = N..N+63
 */
#ifndef CTK_H_
#define CTK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (what a binding written against an older header must know):
 *   v9 (round 6): + ctk_set_option / ctk_get_option (every back-end choice of the library in one validated, atomic table; the
 *       environment variables are read ONCE when the library is loaded); - the two stream-K scratch entry points of v7 (the
 *       stream-K walk of the persistent GEMMs left the library); ctk_gemm_pp_mode(m) = ctk_set_option(CTK_OPT_GEMM_PP, m) and
 *       accepts bits 0 and 5 only; the release library has no debug switches and exports no ctk_debug_* symbol.
 *   v8 (round 4): + ctk_bilinear_sampler (Op D); ctk_window_args.flags must be 0 or CTK_WINDOW_NO_SPACE_ATTN -- unknown bits are
 *       CTK_E_SHAPE in every entry point taking the struct; ctk_probe_mfma kind 2; the *_workspace_bytes queries no longer include
 *       the stream-K scratch.
 *   v7: ctk_gemm_scratch_bytes and its setter (removed in v9).   v6: ctk_window_args.flags, the encoder entry points.   v5: CoTracker2 window. */
#define CTK_ABI_VERSION 9
#define CTK_LEVELS 4
#define CTK_C 128          /* latent_dim                       cotracker3_online.py:60  */
#define CTK_TAPS 49        /* (2*corr_radius+1)^2, radius 3    build_cotracker.py:33    */
#define CTK_CORR_K 2401    /* 49*49                            cotracker3_online.py:84  */
#define CTK_CORR_LD 2432   /* 2401 padded to a multiple of 32 (zero columns)            */
#define CTK_HID 384        /* hidden_size                      cotracker3_online.py:77  */
#define CTK_HEADS 8
#define CTK_HEAD_DIM 48
#define CTK_MLP 1536
#define CTK_VIRT 64        /* num_virtual_tracks               cotracker3_online.py:49  */
#define CTK_X_DIM 1110     /* input_dim                        cotracker3_online.py:71  */
#define CTK_X_LD 1120      /* 1110 padded to a multiple of 32                           */
#define CTK_X_CORR 0
#define CTK_X_VIS 1024
#define CTK_X_CONF 1025
#define CTK_X_POSENC 1026
#define CTK_DEPTH 3        /* time_depth = space_depth = 3     cotracker3_online.py:74-75 */

enum {
  CTK_OK = 0,
  CTK_E_NULL = -1,      /* required pointer is NULL            */
  CTK_E_SHAPE = -2,     /* size out of range / not supported   */
  CTK_E_ALIGN = -3,     /* pointer or leading dimension not 16-byte aligned */
  CTK_E_WORKSPACE = -4, /* workspace too small                 */
  CTK_E_STATE = -5      /* call not allowed in the current state (e.g. graph capture while the profiler is on) */
};

enum { CTK_ACT_NONE = 0, CTK_ACT_GELU_ERF = 1, CTK_ACT_GELU_TANH = 2 };

/* Weights of one transformer block (AttnBlock blocks.py:401-438 or CrossAttnBlock
 * cotracker.py:534-577).  Linear weights are torch layout [out,in] row-major.        */
typedef struct ctk_block_weights {
  const float* wq;   const float* bq;    /* to_q      [384,384],[384]   blocks.py:375 */
  const float* wkv;  const float* bkv;   /* to_kv     [768,384],[768]   blocks.py:376 (k rows 0..383, v rows 384..767) */
  const float* wo;   const float* bo;    /* to_out    [384,384],[384]   blocks.py:377 */
  const float* w1;   const float* b1;    /* mlp.fc1   [1536,384],[1536] blocks.py:61  */
  const float* w2;   const float* b2;    /* mlp.fc2   [384,1536],[384]  blocks.py:67  */
  const float* ctx_gamma; const float* ctx_beta; /* norm_context [384] (cotracker.py:540) or NULL for AttnBlock */
  /* optional ctk_pack_weight blobs of wq/wkv/wo/w1/w2: when non-NULL that Linear runs on the
   * split-half MFMA back end, when NULL on the exact-f32 one (then the f32 pointer must be set) */
  const void* wq_p; const void* wkv_p; const void* wo_p; const void* w1_p; const void* w2_p;
} ctk_block_weights;

/* EfficientUpdateFormer (cotracker.py:387-531) + corr_mlp (cotracker3_online.py:84). */
typedef struct ctk_model_weights {
  const float* corr_fc1_w;   /* [384, CTK_CORR_LD] zero-padded columns */
  const float* corr_fc1_b;   /* [384]  */
  const float* corr_fc2_w;   /* [256,384] */
  const float* corr_fc2_b;   /* [256]  */
  const float* in_w;         /* input_transform.weight, columns permuted to the x layout, [384, CTK_X_LD] */
  const float* in_bias_t;    /* [S,384] = input_transform.bias + W @ time_emb_S[t]  (time embedding
                                 folded into the projection: W(x+e_t)+b = Wx + (W e_t + b),
                                 cotracker3_online.py:247 + cotracker.py:484) */
  const float* virtual_tokens; /* virual_tracks [64,384]        cotracker.py:416 */
  const float* head_w;       /* [4,384] = cat(flow_head.weight, vis_conf_head.weight)  cotracker.py:526-529 */
  const float* head_b;       /* [4] */
  const void* corr_fc1_p;    /* optional ctk_pack_weight blobs of corr_fc1_w / corr_fc2_w / in_w (see ctk_block_weights) */
  const void* corr_fc2_p;
  const void* in_p;
  ctk_block_weights time_blocks[CTK_DEPTH];
  ctk_block_weights virtual2point[CTK_DEPTH];
  ctk_block_weights virtual_self[CTK_DEPTH];
  ctk_block_weights point2virtual[CTK_DEPTH];
} ctk_model_weights;

/* One sliding window / offline pass.  Replaces CoTrackerThreeOnline.forward_window
 * (cotracker3_online.py:171-264) and the inline loop of cotracker3_offline.py:139-216. */
typedef struct ctk_window_args {
  int32_t S;                  /* frames in the window (16 online/sliding, T offline)  */
  int32_t N;                  /* tracked points                                       */
  int32_t iters;              /* update iterations (predictor.py:158 uses 6)          */
  int32_t H[CTK_LEVELS];      /* level sizes                                          */
  int32_t W[CTK_LEVELS];
  const float* fmaps[CTK_LEVELS];   /* NHWC [S,H_l,W_l,128], first frame of the window */
  const float* support[CTK_LEVELS]; /* [N,49,128]                                      */
  const uint8_t* point_mask;  /* [N] 1 = track already queried (attention_mask, cotracker3_online.py:484,493-496); NULL = all 1 */
  float* coords;              /* [S,N,2] in/out, level-0 feature units                 */
  float* vis;                 /* [S,N]   in/out, logits                                */
  float* conf;                /* [S,N]   in/out, logits                                */
  float scale_x, scale_y;     /* model_resolution / stride = (W/4, H/4)  cotracker3_online.py:224-232 */
  int32_t points_per_chunk;   /* correlation stage processes this many points at a time (0 = all) */
  void* aux_stream;           /* optional second HIP stream (or NULL).  When given, ctk_forward_window forks work onto it
                                 and joins it back before returning control of `stream` to later launches: the sampler of
                                 one point piece beside corr_mlp of the previous one, and the points<-virtual query
                                 projection beside the virtual-track chain.  Same results, bit for bit (the launches and
                                 their inputs are unchanged; only their stream differs).  Must not be the capture-origin
                                 of another graph; safe inside ctk_window_graph_create (it joins that capture).  */
  int32_t flags;              /* CTK_WINDOW_NO_SPACE_ATTN: EfficientUpdateFormer.forward(add_space_attn=False) -- only the time
                                 blocks run, the virtual tracks are still appended and stripped (cotracker.py:496-502,521-523).
                                 MUST be 0 otherwise: every entry point taking this struct returns CTK_E_SHAPE when an
                                 unknown bit is set (a caller that built the pre-v6 struct hands over 4 bytes of garbage) */
} ctk_window_args;
#define CTK_WINDOW_NO_SPACE_ATTN 1

int ctk_abi_version(void);
const char* ctk_error_string(int code);

/* ---- whole-window driver (Op A + C + B + state update, `iters` times) ----------- */
int ctk_forward_window_workspace_bytes(const ctk_window_args* a, size_t* out_bytes);
int ctk_forward_window(const ctk_window_args* a, const ctk_model_weights* w,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- hipGraph of a whole window (BASELINE.json configs[3]: streaming update captured once, replayed per chunk)
 * ctk_window_graph_create captures ONE ctk_forward_window(a, w, workspace) -- every launch of all `iters`
 * iterations -- on a private capture stream and instantiates it.  The executable graph bakes in the POINTERS
 * of *a, *w and workspace: the caller keeps those buffers alive and at the same addresses and refreshes their
 * CONTENTS (pyramid, support, coords/vis/conf, point_mask) before every ctk_window_graph_launch, which
 * enqueues the whole window on `stream` as one graph launch.  The handle is a host object owned by the caller
 * (destroy with ctk_window_graph_destroy); it holds no device memory.  Capture is refused (CTK_E_STATE)
 * while ctk_profile_enable(1) is active: events cannot be recorded inside a capture.                      */
typedef struct ctk_window_graph ctk_window_graph;
int ctk_window_graph_create(const ctk_window_args* a, const ctk_model_weights* w, void* workspace,
                            size_t workspace_bytes, ctk_window_graph** out);
int ctk_window_graph_launch(ctk_window_graph* g, void* stream);
int ctk_window_graph_nodes(const ctk_window_graph* g, int64_t* out_nodes); /* kernel nodes captured */
int ctk_window_graph_destroy(ctk_window_graph* g);

/* ---- Op A: corr_embed  (cotracker3_online.py:190-210; get_correlation_feat :130-143,
 *      einsum :202-204, corr_mlp :205) -> x[:, 0:1024]                                */
int ctk_corr_embed_workspace_bytes(const ctk_window_args* a, size_t* out_bytes);
int ctk_corr_embed(const ctk_window_args* a, const ctk_model_weights* w, float* x /*[N*S,CTK_X_LD]*/,
                   void* workspace, size_t workspace_bytes, void* stream);
/* 49x49 correlation volume only (before corr_mlp), for parity tests:
 * out [4][N*S][CTK_CORR_LD], row = n*S+t.                                             */
int ctk_corr_volume(const ctk_window_args* a, float* out, void* stream);
/* The same volumes from the split-half pipeline's sampler (footprint x support on f16 MFMA x3, blend after
 * the correlation), in SH format: out halves [4][N*S][2*CTK_CORR_LD].  workspace holds the SH copy of the
 * window's pyramid (ctk_corr_volume_sh_workspace_bytes).                                                  */
int ctk_corr_volume_sh_workspace_bytes(const ctk_window_args* a, size_t* out_bytes);
int ctk_corr_volume_sh(const ctk_window_args* a, void* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Op C: assemble_tokens (cotracker3_online.py:212-245, posenc :19-39) -> x[:,1024:1120] */
int ctk_assemble_tokens(const ctk_window_args* a, void* x /* f32 [N*S,CTK_X_LD], or SH when x_split */,
                        int32_t x_split, void* stream);

/* ---- Op B: EfficientUpdateFormer.forward (cotracker.py:483-531): x -> delta [N*S,4] */
int ctk_update_former_workspace_bytes(int32_t S, int32_t N, size_t* out_bytes);
int ctk_update_former(int32_t S, int32_t N, const float* x, const ctk_model_weights* w,
                      float* delta /*[N*S,4] row=n*S+t*/, void* workspace, size_t workspace_bytes, void* stream);

/* ---- general update former: CoTracker2 (cotracker.py:46-56: time_depth = space_depth = 6, input_dim 456,
 * output_dim 130, attention mask) on the same block kernels.  x [N*S, in_ld] (f32, or SH when x_split), row = n*S+t,
 * columns >= in_dim zero; delta [N*S, out_ld] f32 (columns >= out_dim are the zero-padded head rows).  point_mask [N]
 * (1 = track already queried) masks the point KEYS of virtual<-points and the point QUERIES of points<-virtual exactly
 * as CrossAttnBlock.forward does (cotracker.py:560-572); NULL = no mask.  Workspace: ctk_update_former_workspace_bytes. */
typedef struct ctk_former_weights {
  int32_t depth;               /* layers (6)                                                     */
  int32_t in_dim, in_ld;       /* 456, padded to a multiple of 32 (480)                          */
  int32_t out_dim, out_ld;     /* 130, padded to a multiple of 64 (192)                          */
  const float* in_w;           /* input_transform.weight [384, in_ld] (zero padded) or NULL when in_p */
  const void* in_p;            /* ctk_pack_weight blob: selects the split-half back end for EVERY Linear */
  const float* in_b;           /* input_transform.bias [384] (used when in_bias_t == NULL)       */
  const float* in_bias_t;      /* optional [S,384] = bias + W @ time_emb[t] (time embedding folded in) */
  const float* virtual_tokens; /* virual_tracks [64,384]                                         */
  const float* head_w;         /* flow_head.weight [out_ld,384] zero padded rows, or NULL when head_p */
  const void* head_p;
  const float* head_b;         /* [out_ld]                                                       */
  const ctk_block_weights* time_blocks;    /* [depth] */
  const ctk_block_weights* virtual2point;  /* [depth] */
  const ctk_block_weights* virtual_self;   /* [depth] */
  const ctk_block_weights* point2virtual;  /* [depth] */
} ctk_former_weights;
int ctk_update_former_ex(int32_t S, int32_t N, const void* x, int32_t x_split, const ctk_former_weights* w,
                         const uint8_t* point_mask, float* delta, void* workspace, size_t workspace_bytes, void* stream);

/* CoTracker2 iteration around CorrBlock and the former (cotracker.py:128-172).  Layouts: coords / track_mask / vis
 * [S,N,*], track_feat [S,N,128] (= CorrBlock targets), fcorrs [N,S,196] (ctk_corrblock_sample), pos [N,456].
 * ctk_v2_assemble: x[n*S+t][0..in_ld) = cat(get_2d_embedding(coords - coords[0]) (130), fcorrs (196), track_feat (128),
 *   track_mask, vis) + pos[n], zero padded to in_ld; the time embedding (:150) is folded into in_bias_t.
 * ctk_v2_apply_delta: coords[t,n] += delta[n*S+t][0:2]; normed[t*N+n][0:128] = GroupNorm(1,128)(delta[n*S+t][2:130])
 *   (:157-167; the Linear + GELU + residual of track_feat_updater is then one ctk_gemm with resid = C = track_feat).
 * ctk_v2_vis_head: vis_predictor (:172).  ctk_sample_features4d: sample_features4d (model_utils.py:258-290) of a
 *   channels-last map [H,W,C] at (x, y) -> [N,C] (4-D grid_sample semantics; used for pos_emb, :126-130).          */
int ctk_v2_assemble(int32_t S, int32_t N, const float* coords, const float* fcorrs, const float* track_feat,
                    const float* track_mask, const float* vis, const float* pos, int32_t in_ld, void* x, int32_t x_split,
                    void* stream);
int ctk_v2_apply_delta(int32_t S, int32_t N, const float* delta, int32_t out_ld, float* coords, const float* gamma,
                       const float* beta, float eps, float* normed, void* stream);
int ctk_v2_vis_head(const float* track_feat, int64_t R, const float* w, const float* b, float* out, void* stream);
int ctk_sample_features4d(const float* map, int32_t H, int32_t W, int32_t C, const float* coords, int32_t N, float* out,
                          void* stream);

/* ---- Op D: the stand-alone bilinear_sampler with the reference's FULL signature (model_utils.py:191-255) ----
 * input  [B,C,H,W] (D = 0) with coords [B,P,2] = (x, y), or [B,C,D,H,W] (D > 0) with coords [B,P,3] = (t, x, y), both in
 *        the reference's own NCHW layout; P = product of the coords' inner dims (Ho*Wo, or Do*Ho*Wo);
 * align_corners 0 / 1, padding_mode CTK_PAD_ZEROS / CTK_PAD_BORDER ("reflection": CTK_E_SHAPE);  out [B,C,P].
 * Bit-identical to torch.nn.functional.grid_sample on the CPU for finite coordinates (4-D: ATen's vectorised kernel with its
 * FMA contractions; 5-D: the scalar grid_sampler_3d -- csrc/sampler_math.h), NaN coordinates are not specified.
 * Also what sample_features4d / sample_features5d (model_utils.py:258-323) reduce to (cotracker_amd/model_utils.py).     */
#define CTK_PAD_ZEROS 0
#define CTK_PAD_BORDER 1
int ctk_bilinear_sampler(const float* input, int32_t B, int32_t C, int32_t D, int32_t H, int32_t W, const float* coords,
                         int64_t P, int32_t align_corners, int32_t padding_mode, float* out, void* stream);

/* ---- CoTracker2 window driver: CoTracker2.forward_window (cotracker.py:86-173) as ONE capture-safe call per window
 * (the CoTracker2 counterpart of ctk_forward_window): sampled positional embedding once, then `iters` x { CorrBlock
 * sample -> token assembly -> update former (attention mask) -> coords += delta, GroupNorm(feature delta) ->
 * track_feat += GELU(Linear(.)) }, then the visibility head.  coords and track_feat are updated IN PLACE.
 * ctk_v2_window_graph_create captures one such call into a hipGraph (same ownership rules as ctk_window_graph_create;
 * launch / nodes / destroy through the ctk_window_graph_* functions).                                               */
typedef struct ctk_v2_window_args {
  int32_t S, N, iters;
  int32_t H[CTK_LEVELS], W[CTK_LEVELS];
  const float* fmaps[CTK_LEVELS]; /* NHWC [S,H_l,W_l,128] CorrBlock pyramid, NOT normalised (blocks.py:300-307)  */
  float* coords;                  /* [S,N,2] in/out, level-0 feature units                                       */
  float* track_feat;              /* [S,N,128] in/out, already multiplied by the attention mask (cotracker.py:346) */
  const float* vis;               /* [S,N] visibility logits fed to the former (constant over the iterations)    */
  const float* track_mask;        /* [S,N] 0/1 as float (cotracker.py:338-344)                                   */
  const uint8_t* point_mask;      /* [N] attention mask (cotracker.py:331-333) or NULL                           */
  float* vis_out;                 /* [S,N] vis_predictor(track_feat) after the last iteration (cotracker.py:172) */
} ctk_v2_window_args;
typedef struct ctk_v2_weights {
  ctk_former_weights former;      /* 6 + 6 layers, 456 -> 130 (in_ld / out_ld padded)                            */
  const float* pos_hwc; int32_t pos_h, pos_w;  /* pos_emb as channels-last [pos_h,pos_w,456] (cotracker.py:60-66) */
  const float* norm_w; const float* norm_b;    /* GroupNorm(1,128) (cotracker.py:79)                             */
  const float* upd_w; const void* upd_p; const float* upd_b; /* track_feat_updater Linear 128->128 (f32 and/or packed) */
  const float* vis_w; const float* vis_b;      /* vis_predictor Linear 128->1: [128], [1]                        */
} ctk_v2_weights;
int ctk_forward_window_v2_workspace_bytes(const ctk_v2_window_args* a, const ctk_v2_weights* w, size_t* out_bytes);
int ctk_forward_window_v2(const ctk_v2_window_args* a, const ctk_v2_weights* w, void* workspace, size_t workspace_bytes,
                          void* stream);
int ctk_v2_window_graph_create(const ctk_v2_window_args* a, const ctk_v2_weights* w, void* workspace,
                               size_t workspace_bytes, ctk_window_graph** out);

/* ---- Op D: standalone samplers --------------------------------------------------- */
/* Integer floor indices (x0,y0) of the 7 x-taps and 7 y-taps of every (t,n,level), exactly as
 * bilinear_sampler + ATen grid_sampler_3d compute them (model_utils.py:238-255).
 * out int32 [S,N,4,2,7].  The bit-exactness contract of BASELINE.md section 2.          */
int ctk_tap_indices(const ctk_window_args* a, int32_t* out, void* stream);
/* get_correlation_feat (cotracker3_online.py:130-143) for one level: out [S,N,49,128].  */
int ctk_sample_patches(const float* fmap /*NHWC [S,H,W,128]*/, int32_t S, int32_t H, int32_t W,
                       const float* coords /*[S,N,2] level-0 units*/, int32_t N, int32_t level,
                       float* out, void* stream);
/* get_track_feat (cotracker3_online.py:113-128, sample_features5d model_utils.py:293-323):
 * trilinear support patches for one level.  fmap NHWC [T,H,W,128]; frames [N] float (frame
 * index relative to fmap[0]); coords [N,2] in this level's units; out [N,49,128].        */
int ctk_sample_support(const float* fmap, int32_t T, int32_t H, int32_t W, const float* frames,
                       const float* coords, int32_t N, float* out, void* stream);
/* CorrBlock.corr + CorrBlock.sample fused (cotracker/models/core/cotracker/blocks.py:284-362, CoTracker2's 4D
 * correlation-volume sampler; num_levels=4, radius=3, padding_mode="border" as built at cotracker.py:119-124):
 * out[n*S+s][l*49 + a*7 + b] = bilinear sample (grid_sampler_2d, align_corners) of the level-l volume
 * <targets[s,n,:], fmaps_l[s,:,y,x]> / sqrt(128) at (x, y) = coords[s,n]/2^l + (a-3, b-3).  The volume is never
 * materialised: only the <=9x9 footprint dots are formed.  fmaps[l]: NHWC [S,H[l],W[l],128] (level l>0 =
 * ctk_avg_pool2_nhwc of level l-1, NOT normalised -- blocks.py:300-307); targets [S,N,128] (= track_feat,
 * blocks.py:342); coords [S,N,2] level-0 units; out [N,S,196] (the reference's [B*N,S,LRR], blocks.py:338-339). */
int ctk_corrblock_sample(const float* const* fmaps, const int32_t* H, const int32_t* W, int32_t S, int32_t N,
                         const float* targets, const float* coords, float* out, void* stream);
/* Channel L2-normalise + NCHW->NHWC (cotracker3_online.py:384-394) and 2x2 average pooling
 * (:401-409).  in [F,128,H,W] -> out NHWC [F,H,W,128]; pool: in NHWC [F,H,W,128] -> [F,H/2,W/2,128]. */
int ctk_normalize_to_nhwc(const float* in, int32_t F, int32_t H, int32_t W, float* out, void* stream);
int ctk_avg_pool2_nhwc(const float* in, int32_t F, int32_t H, int32_t W, float* out, void* stream);

/* ---- CNN feature encoder on the split-half MFMA path (round 3) ---------------------------------------------
 * BasicEncoder (cotracker/models/core/cotracker/blocks.py:141-219, ResidualBlock :79-138), called once per forward at
 * cotracker3_online.py:373-384 / cotracker3_offline.py:60-75.  Activations are NHWC; a convolution reads SH-format input
 * ([pixel][C/32] lines of 128 bytes: 32 hi | 32 lo halves, see the SH notes below) and writes f32 [pixel][n_out].  The host
 * side (co-tracker_amd/encoder_hip.py) strings these calls together in the reference's order.
 *
 * ctk_conv2d_sh: nn.Conv2d(Cin, n_out, (KH,KW), stride, padding=pad, zeros) as an implicit GEMM (no im2col buffer).
 *   in_sh  [F][Hin][Win][Cin/32] lines; Cin % 32 == 0
 *   wp     ctk_pack_weight blob of the matrix W'[n_pad][KH*KW*Cin], W'[n][(ky*KW+kx)*Cin + c] = weight[n][c][ky][kx], rows
 *          n >= n_out zero, n_pad % 128 == 0; bias: n_pad floats (zeros beyond n_out)
 *   out    f32 [F*Hout*Wout][n_out], Hout = (Hin + 2 pad - KH)/stride + 1; zeros: >= 128 zero bytes on the device          */
int ctk_conv2d_sh(const void* in_sh, int32_t F, int32_t Hin, int32_t Win, int32_t Cin, const void* wp, const float* bias,
                  int32_t n_out, int32_t n_pad, int32_t KH, int32_t KW, int32_t stride, int32_t pad, float* out,
                  const void* zeros, void* stream);
/* Stem: x = 2*(frame/255) - 1 (cotracker3_online.py:320) and the 7x7 stride-2 pad-3 patches of the 3-channel frames
 * [F,3,H,W] as SH rows of 160 columns ((ky,kx,c) order, 147 used, rest 0): conv1 (blocks.py:150-157) = ctk_conv2d_sh on
 * a [F][Ho][Wo] grid with Cin = 160, 1x1, stride 1.  out_sh: F*Ho*Wo*160*4 bytes, Ho = (H-1)/2+1.                         */
int ctk_enc_stem_im2col(const float* frames, int32_t F, int32_t H, int32_t W, void* out_sh, void* stream);
/* nn.InstanceNorm2d (no affine, eps, biased variance; blocks.py:110-113,147-148) statistics of x f32 [F][HW][C]:
 * stats [F][C][2] = (mean, 1/sqrt(var + eps)), sums in f64 in a fixed order.  workspace: ctk_enc_inorm_workspace_bytes.    */
int ctk_enc_inorm_workspace_bytes(int32_t F, int64_t HW, int32_t C, size_t* out_bytes);
int ctk_enc_inorm_stats(const float* x, int32_t F, int64_t HW, int32_t C, float eps, float* stats, void* workspace, void* stream);
/* y = relu((x - mean) * rstd)  (blocks.py:130-131, 188-190, 216-217); with skip: out = relu(skip' + y) (blocks.py:138) where
 * skip' = skip, or (skip - mean_s) * rstd_s when skip_stats is given (the 1x1 downsample branch, blocks.py:123-126,133-136).
 * Writes SH (out_sh, the next convolution's input) and / or f32 (out_f32); C % 32 == 0.                                    */
int ctk_enc_inorm_apply(const float* x, const float* stats, const float* skip, const float* skip_stats, int32_t F, int64_t HW,
                        int32_t C, void* out_sh, float* out_f32, void* stream);
/* F.interpolate(., (Ho, Wo), bilinear, align_corners=True) of the four stage outputs (f32 NHWC [F][H_k][W_k][C_k]) and
 * torch.cat over channels (blocks.py:198-215) -> SH [F][Ho][Wo][sum C_k / 32] lines, the input of conv2.                    */
int ctk_enc_fuse(const float* const* src, const int32_t* H, const int32_t* W, const int32_t* C, int32_t F, int32_t Ho, int32_t Wo,
                 void* out_sh, void* stream);
/* fmaps / sqrt(max(sum_c fmaps^2, 1e-12)) on NHWC [P][128] (cotracker3_online.py:384-394).                                  */
int ctk_enc_l2norm(const float* x, int64_t P, float* out, void* stream);

/* ---- primitives (exported for unit tests and reuse) ------------------------------ */
/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N] + bias_rows[(m % period),N]) + resid[M,N]
 * N % 64 == 0, K % 32 == 0, lda/ldw % 4 == 0, A and W 16-byte aligned.  batch > 1 repeats with
 * element strides a_bs / c_bs (shared W).  Two back ends for the same nn.Linear contract:
 *   Wp == NULL : exact-f32 MFMA (v_mfma_f32_32x32x2_f32), W = torch layout [N,K] f32
 *   Wp != NULL : split-half MFMA (3 x v_mfma_f32_32x32x16_f16 per product, f32 accumulate, ~2^-21
 *                relative per product); Wp = blob written by ctk_pack_weight, W is ignored.      */
typedef struct ctk_gemm_args {
  const float* A; int64_t lda; int32_t M;
  const float* W; int64_t ldw; int32_t N; int32_t K;
  const void* Wp;
  float* C; int64_t ldc;
  const float* bias;
  const float* bias_rows; int32_t bias_period;
  const float* resid; int64_t ldr;
  int32_t act;
  int32_t batch; int64_t a_bs; int64_t c_bs;
  int32_t k_valid;         /* non-padding columns of K (0 = K); only used for flop accounting */
  int32_t a_split;         /* A is in SH format (see below): lda / a_bs count halves, lda % 64 == 0; needs Wp */
  int32_t c_split;         /* write C in SH format: ldc / c_bs count halves, ldc % 64 == 0, no resid; needs Wp */
} ctk_gemm_args;
int ctk_gemm(const ctk_gemm_args* g, void* stream);
/* Split a torch-layout weight [N,K] (K % 32 == 0, row stride ldw) into the packed two-half form
 * the split-half back end reads: 64-byte header {s, 1/s} (s = power of two, chosen on the device
 * from max|W|) + [N][K/32][2][32] IEEE halves (hi, lo of s*W).  Done once per weight at load.   */
/* SH ("split-half") format of an activation matrix X[M][K], K % 32 == 0: IEEE halves
 * [M][K/32][2][32] -- per row and 32-column tile one 128-byte line: 32 hi halves, 32 lo halves,
 * x = hi + lo (hi = rn16(x), lo = rn16(x - hi), |x| < 65504).  Same size as f32.  The split-half
 * GEMM streams it straight into LDS; LayerNorm / attention / GEMM epilogues / token assembly can
 * emit it.  ctk_split_rows converts f32 [M][K] (row stride ld floats) to SH.                      */
/* Numeric range of the split-half format (what "fp32-class" means here; pinned by tests/test_gpu_range.py):
 *   2^-3 <~ |x| < 65504   both halves are normal f16 numbers: x is carried with >= 21 significant bits, a product of two SH
 *                          operands (3 MFMAs, f32 accumulate) has ~2^-21 relative error;
 *   |x| <~ 2^-3            `lo` falls into the f16 subnormals (step 2^-24): the error becomes ABSOLUTE, <= 2^-25 per element;
 *                          below 6.1e-5 `hi` is subnormal too -- same absolute bound.  Harmless where the consumer is on an
 *                          absolute scale (softmax logits, residual adds on an O(1) stream), which is every consumer on this
 *                          path: LayerNorm re-normalises the residual stream before every GEMM, and packed weights are
 *                          pre-scaled by a power of two into [2^13, 2^14) so their own `lo` never underflows;
 *   |x| >= 65504           `hi` overflows to inf, `lo` becomes -inf/NaN, and the non-finite value reaches the window state
 *                          (coords / vis / conf) -- nothing on the path clamps or masks it.  The library itself does not test
 *                          for it; the CoTracker3 host models check the finished window state once per forward and re-run that
 *                          forward on the exact-f32 back end (Wp == NULL everywhere) when it is non-finite (cotracker_amd/model.py,
 *                          `range_guard`; with the streaming hipGraph the check is deferred by one call and a hit RAISES instead --
 *                          INTEGRATION.md).  The CoTracker2 host model (model_v2.py, ctk_forward_window_v2) does NOT check: it is
 *                          unguarded.  A caller that drives ctk_forward_window directly should do the check itself.  Measured
 *                          margin on synthetic weights at BASELINE configs[1] scale: the MLP hidden layer has to be scaled by
 *                          3e4 before the first fallback (profiles/r03_range_sweep_c2.json).
 * The exact-f32 back end (v_mfma_f32_32x32x2_f32, f32 operands in HBM) has the reference's range and ~1/3 of the speed.    */
int ctk_split_rows(const float* x, int64_t ld, int64_t M, int32_t K, void* out, void* stream);
int ctk_pack_weight_bytes(int32_t N, int32_t K, size_t* out_bytes);
int ctk_pack_weight(const float* W, int64_t ldw, int32_t N, int32_t K, void* packed, void* stream);

/* LayerNorm over 384 channels, rows [0,R): y = (x-mean)/sqrt(var+eps) [*gamma+beta].
 * y is f32 [R,384], or SH [R][12][2][32] halves when out_split != 0.                     */
int ctk_layernorm(const float* x, void* y, int64_t R, const float* gamma, const float* beta,
                  float eps, int32_t out_split, void* stream);

/* softmax(q k^T * 48^-0.5) v  (Attention.forward, blocks.py:379-398), 8 heads x 48.
 * row(b,i) = b*bs + i*is (rows of a matrix with leading dimension ld floats).           */
typedef struct ctk_attn_args {
  const float* q; int64_t q_ld; int64_t q_bs; int64_t q_is;
  const float* k; const float* v; int64_t kv_ld; int64_t kv_bs; int64_t kv_is;
  void* out; int64_t o_ld; int64_t o_bs; int64_t o_is;   /* f32, or SH halves (o_ld = 768) when o_split */
  int32_t nbatch; int32_t n1; int32_t n2;
  int32_t splits;          /* key-range splits (>1 needs workspace) */
  float* partial;          /* [splits, nbatch, 8, n1, 50] or NULL    */
  int32_t o_split;         /* write out in SH format                 */
  /* CoTracker2's attention_mask (CrossAttnBlock.forward, cotracker.py:560-572), one flag per point, shared by all
   * batches (frames) and heads; NULL = no mask.  key_mask[j] == 0: key j gets logit -FLT_MAX (probability 0 unless
   * every key is masked, then uniform -- exactly the reference's additive bias).  query_mask[i] == 0: every logit of
   * query i is -FLT_MAX, i.e. that query attends uniformly (the reference's quirk for not-yet-queried tracks).   */
  const uint8_t* key_mask;   /* [n2] */
  const uint8_t* query_mask; /* [n1] */
} ctk_attn_args;
int ctk_attention(const ctk_attn_args* a, void* stream);

/* ---- opt-in kernel timing (bench.py) ------------------------------------------------
 * When enabled, every kernel launch of this library is bracketed by two HIP events recorded on
 * the launch stream; ctk_profile_read synchronises them and returns one row per kernel with the
 * launch count, summed duration and the summed ALGORITHMIC flops/bytes of those launches.
 * Bench-only: the recorder is process-global and not thread safe (the one exception to the
 * "no mutable global state" rule above; it is off by default).                              */
typedef struct ctk_profile_row {
  char name[32];
  int64_t launches;
  double total_ms;
  double flops;
  double bytes;
} ctk_profile_row;
int ctk_profile_enable(int on);
/* ---- back-end options ------------------------------------------------------------------
 * Where the library holds two kernels for one operator, the choice is an OPTION: a process-wide table of validated integers
 * (relaxed atomics: ctk_set_option may be called from any thread at any time; a launch uses the value it reads when it is
 * enqueued).  The initial value of every option comes from its environment variable, read ONCE when the library is loaded --
 * nothing in the library calls getenv() later.  No option changes what is computed: two back ends of one operator agree to the
 * last-bit differences stated below.  Unknown keys / out-of-range values: CTK_E_SHAPE, nothing changes.
 *   key                                env var            default  values
 *   CTK_OPT_GEMM_PP                    CTK_GEMM_PP        33       bit 0: the big split-half Linears (N % 256 == 0 or N % 192 == 0, >= one
 *                                                                  256-row tile per CU) run on the persistent ping-pong kernels of
 *                                                                  csrc/gemm_pp.hip (0 = always gemm_f16x3.hip's kernels: same products, same
 *                                                                  K order, bit-identical but for the residual Linears' last bit);
 *                                                                  bit 5 (32): tail split -- a last round of tiles that would be <= TAIL_PCT %
 *                                                                  full goes to the 64 x 64-tile kernel
 *   CTK_OPT_GEMM_TAIL_PCT              CTK_GEMM_TAIL_PCT  25       0..100
 *   CTK_OPT_CORR_VERSION               CTK_CORR           3        split-half correlation sampler: 3 = wave-owned footprint rows straight into
 *                                                                  MFMA registers, 1 = the round-3 kernel (footprint through LDS, two barriers per
 *                                                                  frame); same arithmetic, volumes agree to 3e-6
 *   CTK_OPT_CORR_MAP                   CTK_CORR_MAP       3        workgroup -> (point, level) dealing of version 3: 3 = level-major (neighbouring
 *                                                                  points of a grid query run together), 0 = point-major, 1 / 2 / 4 = pairs / blocks
 *   CTK_OPT_ATTENTION_VALU             CTK_ATTN           0        1 = every attention shape on the VALU kernel (the fallback of the MFMA kernels)
 *   CTK_OPT_ATTENTION_TIME_PERSISTENT  CTK_ATTN_TIME      1        0 = the non-persistent time-attention kernel (bit-identical)
 *   CTK_OPT_OVERLAP                    CTK_OVERLAP        0        use of ctk_window_args.aux_stream: bit 0 = sampler of one point piece beside
 *                                                                  corr_mlp of the previous one, bit 1 = the points<-virtual query projection
 *                                                                  beside the virtual-track chain (bit-identical; measured <= 0.2 % either way) */
enum {
  CTK_OPT_GEMM_PP = 0,
  CTK_OPT_GEMM_TAIL_PCT = 1,
  CTK_OPT_CORR_VERSION = 2,
  CTK_OPT_CORR_MAP = 3,
  CTK_OPT_ATTENTION_VALU = 4,
  CTK_OPT_ATTENTION_TIME_PERSISTENT = 5,
  CTK_OPT_OVERLAP = 6,
  CTK_OPT_COUNT = 7
};
int ctk_set_option(int key, int value);
int ctk_get_option(int key, int* value);
/* = ctk_set_option(CTK_OPT_GEMM_PP, mode), ignoring an invalid mode (the pre-v9 name) */
void ctk_gemm_pp_mode(int mode);
int ctk_profile_read(ctk_profile_row* rows, int max_rows, int* nrows);
/* Register-only MFMA loop (2 workgroups x 4 waves per CU) to calibrate the sustained peak of this
 * chip under its power budget: kind 0 = v_mfma_f32_32x32x2_f32, 1 = v_mfma_f32_32x32x16_bf16 (constant operands),
 * 2 = v_mfma_f32_32x32x16_f16 on pseudo-random operands (the instruction and the toggle activity of the split-half
 * kernels: bench.py's `sustained_mfma` figure).  Any other kind: CTK_E_SHAPE.                                   */
int ctk_probe_mfma(int kind, int iters, float* scratch, double* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTK_H_ */
