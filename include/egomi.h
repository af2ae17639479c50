/* egomi.h — C-ABI of libegomi.so: the MI355X (gfx950) kernels under EgoScaler's trajectory-generator
 * hot path.
 *
 * The reference has NO native/FFI boundary on this path: its boundary is the
 * Python nn.Module API of egoscaler/models/pointllm/{builder.py,model_arch.py}, which
 * egoscaler_amd/ mirrors.  This header is the boundary UNDER that API: every entry point names the
 * reference function (file:line, relative to /root/reference/egoscaler/) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked host; caller owns all memory
 *   - no entry point allocates, synchronises or throws; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream)
 *   - return 0 (EGOMI_OK) or a negative EGOMI_E_* code; shapes are validated on the host before
 *     any launch, so a bad call never reaches the GPU
 *   - dtype arguments: EGOMI_F32 / EGOMI_BF16 (raw uint16 storage)
 *   - thread-safe for distinct streams
 */
#ifndef EGOMI_H
#define EGOMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGOMI_OK 0
#define EGOMI_E_BADARG (-1)
#define EGOMI_E_SHAPE (-2)
#define EGOMI_E_LAUNCH (-3)
#define EGOMI_E_UNSUPPORTED (-4)

#define EGOMI_F32 0
#define EGOMI_BF16 1

typedef void* egomi_stream_t;

int egomi_version(void);
const char* egomi_strerror(int code);
/* HIP's message for the most recent failed launch on this thread (diagnostics for EGOMI_E_LAUNCH) */
const char* egomi_last_launch_error(void);

/* ------------------------------------------------------------------------------------------------
 * A1  RGB-D un-projection + ordered compaction (+ optional strided subsample)
 *     replaces  data/tools/pcm_tools.py:68-96  get_points_colors
 *     (frame loop / rgbd concat: vis/interactive.py:22-32, data/train/7_get_object_trajectory.py:244-253)
 *
 * rgb   u8  [B,T,H,W,3]      depth f32 [B,T,H,W]
 * boxes i32 [n_boxes,4] = (ymin,ymax,xmin,xmax) pixels masked out in every frame (may be NULL)
 * x = (u - pp)/fx * z,  y = (v - pp)/fy * z  in float64 exactly as numpy evaluates it;
 * colour = float32(c)/255.0f; valid = all(rgb != 0) & outside boxes & (z < d_thres).
 * d_thres = NaN disables the depth test (d_thres=None in the reference).
 * Output order is the index contract: frame-major, then row-major pixel order, valid pixels only.
 *   n_out == 0 : write every valid pixel  -> out_points f64 [B,cap,3], out_colors f32 [B,cap,3],
 *                cap = T*H*W rows reserved per sample; out_count[b] = n_valid
 *   n_out  > 0 : first-N strided subsample (stride = floor(n_valid / n_out), rows j*stride),
 *                cap = n_out; a sample with n_valid < n_out sets out_count[b] = -n_valid and its
 *                rows are left untouched (the host mirror raises ValueError).  Up to 62.9 M pixels per sample in
 *                this form (the chunk prefix lives in LDS); larger frames return EGOMI_E_UNSUPPORTED
 * Two launches (validity bits + in-chunk ranks + chunk counts; then one thread per output row, or a coalesced
 * dense writer): the workspace holds 2 x 2 B per 16 pixels + 4 B per 4096-pixel chunk per sample.
 */
size_t egomi_unproject_workspace_bytes(int B, int T, int H, int W);
int egomi_unproject_gather(const uint8_t* rgb, const float* depth, const int32_t* boxes, int n_boxes,
                           int B, int T, int H, int W, double pp, double fx, double fy, float d_thres,
                           int n_out, double* out_points, float* out_colors, int32_t* out_count,
                           void* workspace, size_t workspace_bytes, egomi_stream_t stream);

/* N4  depth map -> dense cloud: everything DepthAnything.get_depth does after the network.
 *     replaces  data/third_party/Depth-Anything-V2/metric_depth/depth.py:46-60 (get_depth) and :27-31 (get_only_depth),
 *     called from data/train/7_get_object_trajectory.py:101-108
 * pred f32 [B,h0,w0] network output, rgb u8 [B,H,W,3] (may be NULL when out_points is NULL), tab_ws i32 [W+H] scratch
 * out_z f32 [B,H,W] = PIL Image.resize((W,H), NEAREST) of pred (Pillow's running-double index walk, bit-exact);
 * out_points f64 [B,H*W,3] = ((u-pp)/fx*z, (v-pp)/fy*z, z), out_colors f64 [B,H*W,3] = rgb/255.0 — both or neither;
 * the reference returns no cloud unless fx, fy, pp > 0 (depth.py:53): asking for one with such intrinsics is BADARG. */
int egomi_depth_to_cloud(const float* pred, int B, int h0, int w0, const uint8_t* rgb, int H, int W,
                         double fx, double fy, double pp, int32_t* tab_ws,
                         float* out_z, double* out_points, double* out_colors, egomi_stream_t stream);

/* A2  pc_norm: centre xyz on the centroid, divide by the largest radius, in float64; colours pass
 *     through.   replaces  models/pointllm/pointllm/data/utils.py:146-157
 * points f64 [B,N,3], colors f32 [B,N,3] -> out f32 [B,N,6] */
int egomi_pc_norm(const double* points, const float* colors, float* out, int B, int N, egomi_stream_t stream);

/* A3  farthest point sampling.   replaces  models/pointllm/pointllm/model/pointbert/misc.py:40-60
 * pts f32 [B,N,C] (xyz = first 3 channels, C>=3), start i32 [B] (the reference draws it from the
 * global torch RNG, misc.py:52) -> out_idx i32 [B,G], out_center f32 [B,G,3].
 * fp32, d = (dx*dx+dy*dy)+dz*dz, running min from 1e10, arg-max with the LOWEST index on ties:
 * indices are bit-exact with the reference.  N <= 16384. */
int egomi_fps(const float* pts, int B, int N, int C, const int32_t* start, int G,
              int32_t* out_idx, float* out_center, egomi_stream_t stream);

/* A4+A5  kNN grouping.   replaces  pointbert/dvae.py:107-140 (square_distance, knn_point) and
 *        dvae.py:150-187 (Group.forward gather, centre subtraction on xyz only)
 * pts f32 [B,N,C], center f32 [B,G,3] -> out_idx i32 [B,G,K] ordered by (distance, index)
 * ascending, out_nb [B,G,K,C] (dtype out_dtype).  dist = ((-2*dot)+|c|^2)+|p|^2 in fp32 with
 * dot = (cx*px+cy*py)+cz*pz; no [G,N] matrix touches HBM.  N <= 8192, K <= 64. */
int egomi_knn_group(const float* pts, const float* center, int B, int N, int C, int G, int K,
                    int32_t* out_idx, void* out_nb, int out_dtype, egomi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense products (MFMA).  replaces the aten linear/matmul calls inside
 *   pointbert/dvae.py:193-204 (1x1 convs, A6), pointbert/point_encoder.py:14-18,38-40 (A8),
 *   model/pointllm.py:67-81 (point_proj, A9), HF modeling_llama.py:174-176,243-281 (A11),
 *   model/pointllm.py:187,227-228 (lm_head, A12) and their autograd backward.
 *
 *   C[M,N] = act(alpha * A.B + bias) + residual  (+ C when accumulate)
 * a_layout 0: A is [M,K] row-major;  1: A is [K,M]
 * b_layout 0: B is [N,K] row-major (nn.Linear weight); 1: B is [K,N]
 * ab_dtype F32 (exact fp32 MFMA) or BF16 (fp32 accumulate); c_dtype F32 or BF16 (F32 inputs need
 * F32 output).  bias [N] has ab_dtype; residual [M,N] (ldr) has c_dtype.  act: 0 none, 1 GELU(erf),
 * 2 ReLU.  Batched: grid z in [0,batch): z0 = z / batch_inner, z1 = z % batch_inner, operand
 * pointers advance by z0*s?0 + z1*s?1 ELEMENTS (residual uses C's strides).
 */
typedef struct egomi_gemm_desc {
    const void* A; const void* B; void* C; const void* bias; const void* residual;
    int M, N, K;
    int64_t lda, ldb, ldc, ldr;
    int a_layout, b_layout;
    int ab_dtype, c_dtype;
    int batch, batch_inner;
    int64_t sA0, sA1, sB0, sB1, sC0, sC1;
    float alpha;
    int accumulate;
    int act;
    int force_generic;   /* 1: never take the tuned kernel (used by tests to cross-check it) */
    /* optional fp32 scratch.  Skinny products (M <= 512, e.g. single-token decode): split-K slabs of split_k * M * N * 4
     * bytes; split_k 0 = chosen by the library from the tile count.  Large products (256x256 kernel): the last tile rows
     * are cut into K-slices so the ragged last round fills the chip; their slabs live here too (split_k 0 = planned by
     * the library, rows*16+S = explicit).  NULL = neither. */
    void* workspace; int64_t workspace_bytes; int split_k;
    /* 1: the first 4096 bytes of `workspace` are ticket words that were ZERO when the caller first handed the buffer over and
     * are touched by nobody else; every completed launch leaves them zero again.  With this promise (and workspace_bytes >=
     * 4096 + CUs * 2 * 256 KiB) large products run the PERSISTENT form of the 256x256 kernel: one block per CU walks whole
     * tiles, the remainder of the last round is shared K-slice-wise and summed inside the launch by the last block to
     * arrive (no second kernel, nobody waits).  1: the promise is made, the library decides (today it keeps the per-tile
     * kernel: with weights cold from HBM the persistent form measured 5 % slower in the training step, csrc/gemm_fast.hip);
     * 2: the promise is made and the persistent form is taken for every shape it can run.  0: no promise. */
    int ws_tickets_zeroed;
    /* fused epilogue.  EGOMI_EPI_SWIGLU: B is [Wgate;Wup] stacked with its rows interleaved in blocks of 32 (row 64g+c =
     * Wgate[32g+c], row 64g+32+c = Wup[32g+c]); C [M,N] receives gate|up in that interleaved-32 layout (kept for backward)
     * and C2 [M, N/2] (row stride ldc2) receives silu(gate)*up = what egomi_swiglu_il_fwd(C) would write, bit for bit
     * (replaces the separate pass over C of HF LlamaMLP.forward, modeling_llama.py:174-176).  bf16 output, N % 256 == 0, no
     * bias / residual / activation / alpha / accumulate, products large enough for the 256x256 kernel
     * (egomi_gemm_kernel_id == 2); anything else returns EGOMI_E_UNSUPPORTED.
     * EGOMI_EPI_SWIGLU_BWD: the product A.B^T [M,N] is d(act), the gradient of silu(gate)*up (the data gradient of down_proj); it is never
     * stored.  C2 [M, 2N] (row stride ldc2) holds gate|up in the interleaved-32 layout (read), C [M, 2N] (row stride ldc) receives
     * d(gate|up) in the same layout = what egomi_swiglu_il_bwd would write from the stored bf16 d(act), bit for bit (replaces autograd's
     * backward of modeling_llama.py:174-176 and a round trip of d(act) through HBM).  bf16, N % 64 == 0, same restrictions as above.
     * EGOMI_EPI_SLABS: skinny split-K products (M <= 512, the single-token decode projections) leave their fp32 K-slice slabs
     * [slices][M][N] (row stride N) in `workspace` UNSUMMED and never touch C: the caller's next kernel (egomi_slabs_rmsnorm,
     * egomi_qkv_finish) sums them in slice order while doing its own work, which saves the combine pass and a round trip of the
     * product through HBM.  Plain product only (no bias / residual / activation / alpha / accumulate; add the residual in the
     * consumer), N % 4 == 0.  egomi_gemm_slab_count(desc) tells how many slices the library will write for this descriptor
     * (0: it would not split this product — use EGOMI_EPI_NONE); a request it cannot honour returns EGOMI_E_UNSUPPORTED. */
    int epilogue; void* C2; int64_t ldc2;
} egomi_gemm_desc;
#define EGOMI_EPI_NONE 0
#define EGOMI_EPI_SWIGLU 1
#define EGOMI_EPI_SLABS 2
#define EGOMI_EPI_SWIGLU_BWD 3
int egomi_gemm(const egomi_gemm_desc* desc, egomi_stream_t stream);
/* which kernel egomi_gemm would run for this descriptor: 2 = the 8-phase bf16 NT kernel (256x256 tiles, per-tile or persistent, or the
 * 352x256 form below), 1 = 128x128 / 256x128 bf16 NT kernel, 0 = generic */
int egomi_gemm_kernel_id(const egomi_gemm_desc* desc);
/* 352x256 form of the 8-phase kernel (gemm_nt_bf16_tall_kernel): whole rounds where 256x256 tiles leave a ragged last one (M = 5536, N = 4096:
 * 16 x 16 = 256 tiles instead of 22 x 16 = 352).  mode 0 = never, 1 = by the library's round model (default) — the whole product in this form, or
 * its first Na columns in this form and the rest on 256x256 tiles where neither fits alone (gate|up, the down_proj data gradient) —, 2 = the whole
 * product wherever the form applies (M % 8 == 0, N % 8 == 0, M >= 352, K >= 2048; plain, EGOMI_EPI_SLABS, EGOMI_EPI_SWIGLU, EGOMI_EPI_SWIGLU_BWD),
 * -1 = back to EGOMI_GEMM_TALL / the default.  A product that takes this form whole has no K-sliced tail rows (egomi_gemm_tail_plan: slices = 0).
 * Whole tiles are bit-identical to the 256x256 form's (same K order per element), fused epilogues included.
 * Process-wide, not thread-safe: measurement and tests only.  No reference counterpart (torch.nn.Linear's GEMM is the vendor library's). */
int egomi_gemm_set_tall(int mode);
int egomi_gemm_slab_count(const egomi_gemm_desc* desc);   /* EGOMI_EPI_SLABS: slices egomi_gemm will leave for this descriptor, 0 = none */
/* EGOMI_EPI_SLABS on a LARGE product (egomi_gemm_kernel_id == 2): whole 256-row tiles get the normal epilogue (residual
 * included); the K-sliced tail rows [row0, M) are left as `slices` fp32 slabs [slices][M - row0][N] at the start of the slab
 * area (workspace + 4096 when ws_tickets_zeroed) for egomi_rmsnorm_fwd_tail / egomi_rmsnorm_bwd_tail to sum.  bf16 output, no
 * bias / activation / alpha / accumulate.  slices = 0: the plan has no tail rows, nothing is pending. */
int egomi_gemm_tail_plan(const egomi_gemm_desc* desc, int* row0, int* slices);
/* Query only, for k-major products (egomi_gemm_kernel_id == 3: the weight / data gradients of nn.Linear's backward, train.py:183): the
 * rows [row0, M) egomi_gemm will compute as `slices` K-slices and sum in its own combine pass (ragged last round of 256x256 tiles);
 * slices = 0: none.  EGOMI_E_UNSUPPORTED when the descriptor does not take that kernel. */
int egomi_gemm_tn_tail_plan(const egomi_gemm_desc* desc, int* row0, int* slices);
/* Measurement hooks (bench.py `roofline`; no reference counterpart).  egomi_gemm_time_next(start, stop): the NEXT egomi_gemm call
 * of this thread, if it takes the 256x256 kernel (egomi_gemm_kernel_id == 2), records `start` right before and `stop` right after
 * THAT kernel on the launch stream — the slab-combine pass of K-sliced tail rows is a separate kernel and lies outside the
 * bracket; any other path leaves both events untouched.  The request is consumed by that call either way.  Events come from
 * egomi_event_create (timing-enabled HIP events); egomi_event_elapsed_ms waits for `stop`. */
int egomi_event_create(void** event);
int egomi_event_destroy(void* event);
int egomi_event_elapsed_ms(void* start, void* stop, float* ms);
int egomi_gemm_time_next(void* start, void* stop);

/* ------------------------------------------------------------------------------------------------
 * Fused attention (bf16, head_dim 128; forward also head_dim 64): softmax(scale * Q.K^T + mask).V without materialising the
 * [S,S] scores.  replaces HF eager_attention_forward, transformers/models/llama/modeling_llama.py:
 * 191-214, and its autograd backward (A11).
 * q/k/v: rows = b*S + s, row stride ld_qkv elements, head h at column h*head_dim (q, k, v may be
 * three column slices of one [B*S, 3*H*hd] buffer).  o: [B*S, H*hd] row stride ld_o.
 * mask: key j visible to query i iff (!causal || j <= i) && (key_mask == NULL || key_mask[b*S+j]).
 * lse [B,H,S] fp32 = log sum exp of the scaled, masked scores (saved for backward).
 * backward: delta [B,H,S] fp32 = rowsum(dout * o) (egomi_attn_bwd's dQ kernel computes it into `delta`, its dK/dV kernel reads it), then
 * dq/dk/dv written with row stride ld_dqkv (deterministic: no atomics).
 */
typedef struct egomi_attn_desc {
    const void* q; const void* k; const void* v; void* o; float* lse;
    const void* dout; float* delta; void* dq; void* dk; void* dv;
    const uint8_t* key_mask;
    int B, H, S, head_dim;
    int64_t ld_qkv, ld_o, ld_dqkv;
    float scale;
    int causal;
    int dtype;
    /* optional, backward only: q and k were rotated (HF apply_rotary_pos_emb, modeling_llama.py:139-167) before the
     * attention; when both tables are given (fp32 [>= S, head_dim/2], row = position s), egomi_attn_bwd writes dq and dk
     * already rotated back (what egomi_rope(..., inverse=1) on the bf16 dq/dk would give, bit for bit), so the caller
     * drops that pass.  NULL = plain dq/dk. */
    const float* rope_cos; const float* rope_sin;
} egomi_attn_desc;
int egomi_attn_fwd(const egomi_attn_desc* desc, egomi_stream_t stream);
int egomi_attn_bwd(const egomi_attn_desc* desc, egomi_stream_t stream);
/* A/B switch (like the EGOMI_* environment switches): 1 = the first form of the forward kernel, 2 = the round-3 instruction stream
 * (bit-identical to 1: tests/test_gpu_kernels.py), 3 = the round-4 restructure (32-key tiles in a 4-stage ring, QK of tile t+1 under the
 * exponentials of tile t, lazy running max, end-aligned query blocks; same tolerances vs fp32, not the same bits as 1 / 2), 4 = form 3 in
 * persistent blocks (default at head_dim 128 and S <= 1024: two resident blocks per CU walk the work items as one continuous K/V stream,
 * the next item's Q fragments and key mask prefetched; bit-identical to 3).
 * Process-wide, not thread-safe: measurement and tests only. */
int egomi_attn_set_fwd_form(int form);
int egomi_attn_set_fwd_blocks(int blocks);  /* form 4: cap the number of persistent blocks (0 = two per CU): tests reach several items per block at small shapes */
int egomi_attn_set_fwd_group(int group);   /* block order of form 3: 0 = rank-major, G = a (b,h) pair's blocks in groups of G ranks on one XCD */
int egomi_attn_set_bwd_form(int form);   /* likewise for the two backward kernels */

/* Consumers of EGOMI_EPI_SLABS products (single-token decode; replace splitk combine + the next elementwise kernels of
 * HF LlamaDecoderLayer.forward, modeling_llama.py:243-281, with identical rounding).  slabs: fp32 [slices][rows][cols].
 * egomi_slabs_rmsnorm: x_out = round(sum slabs + residual), h_out = rmsnorm(x_out) * w          (cols % 8 == 0, <= 8192)
 * egomi_qkv_finish   : q|k|v = round(sum slabs) [B, 3*H*hd]; RoPE(pos) on q, k; q -> qkv (row stride ld), k, v -> caches
 *                      [B,H,Smax,hd] at position pos */
int egomi_slabs_rmsnorm(const float* slabs, int slices, int rows, int cols, const void* residual, int64_t ldr, const void* w, float eps,
                        void* x_out, int64_t ldx, void* h_out, int64_t ldh, int dtype, egomi_stream_t stream);
int egomi_qkv_finish(const float* slabs, int slices, void* qkv, int64_t ld, const float* cos_tab, const float* sin_tab, int pos,
                     void* kcache, void* vcache, int B, int H, int hd, int Smax, int dtype, egomi_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * A13  cached decoding.  replaces the past_key_values branch of HF LlamaAttention.forward
 * (modeling_llama.py:243-281) and the greedy step of generate (models/pointllm/model_arch.py:94-108,
 * pointllm/model/pointllm.py:255-275).  KV cache layout [B, H, Smax, hd] per layer (contiguous
 * per-(batch, head) streams).  No entry point reads a length from device memory: every step's
 * constants are arguments, so a whole multi-step decode can be captured into one hipGraph.
 *   kv_append : k, v rows [B*S, H*hd] (row stride ld) -> cache[b, h, pos0+s, :]
 *   attn_decode: q [B, H*hd] against keys [0, T_len) -> out [B, H*hd]; key_mask [B, >=T_len] u8 or NULL
 *   argmax_rows: ids[b] = argmax logits[b, :] (lowest index on ties); seq[b*ld_seq + pos] = ids[b]
 */
int egomi_kv_append(const void* k, const void* v, int64_t ld, void* kcache, void* vcache, int B, int S, int H, int hd, int Smax,
                    int pos0, int dtype, egomi_stream_t stream);
int egomi_attn_decode(const void* q, int64_t ld_q, const void* kcache, const void* vcache, const uint8_t* key_mask, int64_t ld_mask,
                      void* out, int64_t ld_o, int B, int H, int hd, int Smax, int T_len, float scale, int dtype,
                      egomi_stream_t stream);
int egomi_argmax_rows(const void* logits, int64_t ld, int B, int V, int64_t* ids, int64_t* seq, int64_t ld_seq, int pos, int dtype,
                      egomi_stream_t stream);
/* sample_rows: everything HF GenerationMixin.generate does between two forward passes under the arguments the reference passes
 * (models/pointllm/model_arch.py:82-108: do_sample=True, top_k=50, top_p=0.95, temperature, repetition_penalty,
 * output_scores=True, return_dict_in_generate=True; train.py:223-228 validates this way), for a whole batch in one launch:
 *   scores[b, :] = TopP(TopK(Temperature(RepetitionPenalty(logits[b, :]))))   fp32, -inf where a warper removed the token
 *                  (transformers/generation/logits_process.py: RepetitionPenaltyLogitsProcessor on the tokens seq[b, rep_from:pos],
 *                  TemperatureLogitsWarper, TopKLogitsWarper — ties with the k-th value kept —, TopPLogitsWarper — ascending stable
 *                  order, prefix with cumulative probability <= 1 - top_p removed, min_tokens_to_keep = 1)
 *   ids[b]       = do_sample ? a draw from softmax(scores[b, :]) (Gumbel-max on Philox4x32-10 bits keyed by rng[0] = seed and
 *                  rng[1] + draw = draw counter, both read from DEVICE memory at run time so that a replayed hipGraph draws fresh
 *                  numbers; replaces torch.multinomial, whose stream cannot be reproduced) : argmax (lowest index on ties)
 *   done[b]      (may be NULL) HF's unfinished_sequences: a finished row emits pad_id, a row finishes when it emits eos_id (< 0: never)
 *   seq[b*ld_seq + pos] = ids[b]  (may be NULL)
 * top_k = 0 / top_p = 1 / temperature = 1 / repetition_penalty = 1 switch the respective step off.  `pos`, `draw` are launch
 * constants (no length is read from device memory), so T steps can be captured into one hipGraph. */
int egomi_sample_rows(const void* logits, int64_t ld, int B, int V, float* scores, int64_t ld_scores, int64_t* seq, int64_t ld_seq,
                      int pos, int rep_from, int64_t* ids, int* done, float repetition_penalty, float temperature, int top_k,
                      float top_p, int do_sample, const uint64_t* rng, int draw, int64_t eos_id, int64_t pad_id, int dtype,
                      egomi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * A14  trajectory <-> token ids for whole batches, displacement metrics (integer contracts bit-exact).
 * replaces models/pointllm/utils/utils.py:13-21 (discretize_action / token_to_action), :47-104
 * (str_to_float, rt2 6-DoF: split on <tsep>, first run of six <p*> tokens per segment, unmatched
 * segments repeat the previous step), the sequence layout of models/pointllm/dataset.py:150-194 and
 * models/utils/metrics.py:7-55 (ADE/FDE, documented [T,D] form, float64).
 * bins: float64 [num_bins] = numpy.linspace(-1, 1, num_bins), computed by the caller (device memory).
 *   tokenize  : traj f32 [B,Tmax,6], steps i32 [B] (NULL = Tmax) -> ids i64 [B,L] =
 *               <ts> (p*6 <tsep>) x steps <te> <eos> pad..., mask u8 [B,L]; err[b]=1 if L is too short
 *               (sequence truncated to whole steps).  bin = clamp(digitize(v)-1, 0, num_bins-1).
 *   detokenize: ids i64 [B,L] (cut at the first eos) -> out f32 [B,Tmax,6] bin values, n_steps i32 [B]
 *   metrics   : gen/gt f32 [B,Tmax,D] with lengths (NULL = Tmax) -> ade, fde float64 [B]
 */
int egomi_traj_tokenize(const float* traj, const int32_t* steps, int B, int Tmax, const double* bins, int num_bins, int64_t p0,
                        int64_t ts, int64_t tsep, int64_t te, int64_t eos, int64_t pad, int L, int64_t* ids, uint8_t* mask,
                        int32_t* err, egomi_stream_t stream);
int egomi_traj_detokenize(const int64_t* ids, int B, int L, const double* bins, int num_bins, int64_t p0, int64_t tsep, int64_t eos,
                          int Tmax, float* out, int32_t* n_steps, egomi_stream_t stream);
int egomi_traj_metrics(const float* gen, const int32_t* n_gen, const float* gt, const int32_t* n_gt, int B, int Tmax, int D, double* ade,
                       double* fde, egomi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Trainable point backbone (--unfreeze_pc_encoder, models/pointllm/model_arch.py:33-36): the backward
 * and train-mode pieces of pointbert/dvae.py:189-221 (Conv1d / BatchNorm1d / ReLU / max) and
 * pointbert/point_encoder.py:58-76,142 (LayerNorm, DropPath residual).
 *   layernorm_bwd : dx = dx_add + LN'(dy); dw, db (fp32 [cols], caller-zeroed, may be NULL) accumulate
 *   bn_train_fwd  : y = relu?(BatchNorm1d(x)) with BATCH statistics over the R rows (biased variance),
 *                   running stats updated with `momentum` (unbiased variance), stats = fp32 [4*C]
 *                   scratch/saved block {sum, sumsq, mean, rstd}
 *   bn_train_bwd  : dx, dgamma, dbeta (fp32 [C], written by the call) from dy, x, y and the saved stats
 *   group_argmax  : x [BG,M,C] -> max over M and its first arg-max;  group_max_bwd scatters dout back
 *   smallk_wgrad  : dW[n,k] (fp32, caller-zeroed/accumulating) += sum_r dy[r,n]*x[r,k], K <= 8
 *   group_sum     : out[bg,c] = sum_m x[(bg*M+m)*ldx + c]  (backward of the expanded group-global feature)
 *   rowscale_add  : out[r,:] = resid[r,:] + scale[r / rows_per_sample] * branch[r,:]   (DropPath)
 * Column reductions (dw / db, the batch statistics, dgamma / dbeta, dW) are two-stage and ORDERED — one partial row per block in the caller's
 * fp32 scratch `partials`, added in block order by a second kernel; no atomics, the same bits every run (the batch statistics feed the
 * forward output, so this makes the train-mode backbone's forward replayable too).  `partial_floats` must be at least
 *   layernorm_bwd (dw or db given): min(ceil(rows / 4), 512) * 2 * cols        bn_train_fwd / bn_train_bwd: ceil(R / 256) * 2 * C
 *   smallk_wgrad: ceil(R / 512) * N * K
 * (EGOMI_E_BADARG otherwise); cols <= 2048 for layernorm_bwd.
 */
int egomi_layernorm_bwd(const void* dy, const void* x, const void* w, void* dx, const void* dx_add, float* dw, float* db,
                        int rows, int cols, float eps, float* partials, int64_t partial_floats, int dtype, egomi_stream_t stream);
int egomi_bn_train_fwd(const void* x, int64_t R, int C, const void* gamma, const void* beta, float eps, int relu, void* y,
                       float* stats, void* running_mean, void* running_var, float momentum, float* partials, int64_t partial_floats, int dtype,
                       egomi_stream_t stream);
int egomi_bn_train_bwd(const void* dy, const void* x, const void* y, int64_t R, int C, const float* stats, const void* gamma, int relu,
                       float* dgamma, float* dbeta, void* dx, float* partials, int64_t partial_floats, int dtype, egomi_stream_t stream);
int egomi_group_argmax(const void* x, int BG, int M, int C, void* out, int32_t* idx, int dtype, egomi_stream_t stream);
int egomi_group_max_bwd(const void* dout, const int32_t* idx, int BG, int M, int C, void* dx, int64_t ldx, int accumulate, int dtype,
                        egomi_stream_t stream);
int egomi_smallk_wgrad(const void* dy, const void* x, int x_dtype, int64_t R, int N, int K, float* dW, float* partials, int64_t partial_floats, int dtype,
                       egomi_stream_t stream);
int egomi_group_sum(const void* x, int BG, int M, int C, int64_t ldx, void* out, int dtype, egomi_stream_t stream);
int egomi_rowscale_add(const void* resid, const void* branch, const float* scale, int64_t rows, int cols, int rows_per_sample, void* out,
                       int dtype, egomi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row / elementwise kernels (HBM-bound).  `dtype` is the activation/parameter dtype T.
 */
/* LayerNorm forward with optional fused pre-add: s = x (+ add); y = (s-mean)*rstd*w + b; sum_out = s
 * replaces nn.LayerNorm in pointbert/point_encoder.py:60,63,142 and the `x + pos` of :95-98 */
int egomi_layernorm_fwd(const void* x, const void* add, const void* w, const void* b, void* sum_out, void* y,
                        int rows, int cols, float eps, int dtype, egomi_stream_t stream);

/* RMSNorm.  replaces HF LlamaRMSNorm, transformers/models/llama/modeling_llama.py:53-67 (A11).
 * fwd: y = w * T(x * rsqrt(mean(x^2)+eps)); rstd [rows] saved when non-NULL.  cols % 8 == 0.
 * bwd: dx = dx_add + rstd*(w*dy - x_hat*mean(w*dy*x_hat)); dw [cols] fp32 += sum_rows dy*x_hat
 *      (dx_add, dw may be NULL).  cols <= 8192. */
int egomi_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int cols, float eps, int dtype,
                      egomi_stream_t stream);
int egomi_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, const void* dx_add,
                      float* dw, int rows, int cols, int dtype, egomi_stream_t stream);
/* Tail forms: rows >= row0 of the input (x resp. dy) are still the `slices` fp32 K-slice slabs [slices][rows - row0][cols] an
 * EGOMI_EPI_SLABS product left (egomi_gemm_tail_plan); they are summed in slice order (+ residual rows for the forward form),
 * rounded, WRITTEN to x / dy, and normalised in the same pass — bit-identical to the combine pass followed by
 * egomi_rmsnorm_fwd / egomi_rmsnorm_bwd.  row0 == rows: no pending rows (then exactly the plain kernels' arithmetic). */
int egomi_rmsnorm_fwd_tail(void* x, const void* w, void* y, float* rstd, int rows, int cols, float eps, int row0, const float* slabs, int slices,
                           const void* residual, int64_t ldr, int dtype, egomi_stream_t stream);
int egomi_rmsnorm_bwd_tail(void* dy, const void* x, const void* w, const float* rstd, void* dx, const void* dx_add, float* dw, int rows,
                           int cols, int row0, const float* slabs, int slices, int dtype, egomi_stream_t stream);

/* RoPE in place on x [rows, H, hd] (row stride ld elements); position of row r = pos_offset + r % S;
 * cos/sin tables fp32 [>= pos_offset+S, hd/2].  inverse=1 applies the transposed rotation (backward).
 * replaces HF rotate_half/apply_rotary_pos_emb, modeling_llama.py:130-160 (tables: :112-127) */
int egomi_rope(void* x, const float* cos_tab, const float* sin_tab, int64_t rows, int S, int pos_offset, int H, int hd,
               int64_t ld, int inverse, int dtype, egomi_stream_t stream);
/* Tail form for the stacked q|k|v product [rows, 3*H*hd]: rows >= row0 are still `slices` K-slice slabs (EGOMI_EPI_SLABS,
 * egomi_gemm_tail_plan); they are summed, rounded and written (q, k rotated; v as is); rows < row0 get egomi_rope's rotation of
 * q and k in place.  Bit-identical to the combine pass followed by egomi_rope on the first 2*H heads. */
int egomi_rope_qkv_tail(void* qkv, const float* cos_tab, const float* sin_tab, int64_t rows, int S, int pos_offset, int H, int hd, int64_t ld,
                        int row0, const float* slabs, int slices, int dtype, egomi_stream_t stream);

/* SwiGLU: out = silu(gate)*up.  replaces HF LlamaMLP.forward, modeling_llama.py:174-176.
 * gate/up [rows, cols] with row stride ld_in; out / dgate / dup row stride ld_out; dact row stride ld_act. */
int egomi_swiglu_fwd(const void* gate, const void* up, void* out, int64_t rows, int cols, int64_t ld_in, int64_t ld_out,
                     int dtype, egomi_stream_t stream);
int egomi_swiglu_bwd(const void* dact, const void* gate, const void* up, void* dgate, void* dup, int64_t rows, int cols,
                     int64_t ld_in, int64_t ld_act, int64_t ld_out, int dtype, egomi_stream_t stream);
/* The same two ops on the "interleaved-32" gate|up layout: gu / dgu are ONE [rows, 2*cols] array in which hidden unit c
 * sits at column 64*(c/32) + c%32 (gate) and 32 columns further (up).  It is what x . [Wgate;Wup]^T yields when the rows of
 * the stacked weight are interleaved in blocks of 32, which puts gate and up of the same units into the same GEMM wave
 * (egomi_gemm epilogue EGOMI_EPI_SWIGLU).  cols % 32 == 0. */
int egomi_swiglu_il_fwd(const void* gu, void* out, int64_t rows, int cols, int64_t ld_gu, int64_t ld_out, int dtype, egomi_stream_t stream);
int egomi_swiglu_il_bwd(const void* dact, const void* gu, void* dgu, int64_t rows, int cols, int64_t ld_gu, int64_t ld_act, int64_t ld_dgu,
                        int dtype, egomi_stream_t stream);

/* exact (erf) GELU and its derivative.  replaces nn.GELU in point_proj, model/pointllm.py:72-76 */
int egomi_gelu_fwd(const void* x, void* y, int64_t n, int dtype, egomi_stream_t stream);
int egomi_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, egomi_stream_t stream);

/* Row softmax of fp32 scores [Z*Sq, Sk] (row stride ld_s) -> P (dtype, row stride ld_p).
 * key j is visible to query i iff (!causal || j <= q_offset+i) && (key_mask == NULL || key_mask[b*Sk+j]),
 * b = z / heads.  replaces pointbert/point_encoder.py:48-50 and HF eager_attention_forward,
 * modeling_llama.py:204-210.  Sk <= 2048.   bwd: dS = P*(dP - sum_j P*dP). */
int egomi_softmax_fwd(const float* scores, int64_t ld_s, const uint8_t* key_mask, int Z, int heads, int Sq, int Sk, int causal,
                      int q_offset, void* P, int64_t ld_p, int dtype, egomi_stream_t stream);
int egomi_softmax_bwd(const void* P, int64_t ld_p, const float* dP, int64_t ld_dp, void* dS, int64_t ld_ds, int64_t rows, int Sk,
                      int dtype, egomi_stream_t stream);

/* A10  point-token splice.  replaces model/pointllm.py:131-171 (mm_use_point_start_end=True).
 * splice_scan: per sample start_pos = position of the <point_start> whose span receives the features (-1: text-only sample or
 * error), cloud_idx = which of the n_clouds clouds it receives (the reference's running cur_point_idx, :135,143,156), and
 * err: 0 ok | 1 start/end count mismatch (:146) | 2 a <point_end> not at start+P+1 (:150) | 4 cloud index past the clouds given
 * (:143 IndexError).  A sample with several segments is treated as the reference treats it: only its LAST segment is spliced,
 * with the cloud the sample started at, and the cloud index advances once per segment.  scratch: B int32.
 * embed_splice_fwd: out[b,s] = feats[cloud_idx[b], s-start-1] if start < s <= start+P else W[ids[b,s]]
 * (embedding lookup :107 + torch.cat splice :155).  bwd: dW (fp32 [V,d]) += rows outside the span,
 * dfeats[cloud_idx[b]] = rows inside it (rows of clouds no sample received are not written: zero them first).
 * feats/start_pos/dW/dfeats may be NULL; cloud_idx NULL = cloud b for sample b. */
int egomi_splice_scan(const int64_t* ids, int B, int S, int64_t patch_id, int64_t start_id, int64_t end_id, int P, int n_clouds,
                      int32_t* start_pos, int32_t* err, int32_t* cloud_idx, int32_t* scratch, egomi_stream_t stream);
int egomi_embed_splice_fwd(const int64_t* ids, const void* W, const void* feats, const int32_t* start_pos, const int32_t* cloud_idx,
                           int B, int S, int d, int P, int V, void* out, int dtype, egomi_stream_t stream);
int egomi_embed_splice_bwd(const void* dout, const int64_t* ids, const int32_t* start_pos, const int32_t* cloud_idx, int B, int S, int d,
                           int P, int V, float* dW, void* dfeats, int dtype, egomi_stream_t stream);

/* A12  cross-entropy with ignore_index, mean over kept rows.  replaces F.cross_entropy at
 * models/pointllm/train.py:174-181.  count must be zeroed, then ce_count, then ce_fwd_bwd:
 * row_loss[R] (fp32 scratch of the caller) = -log p[target] per row (0 for ignored rows), loss_sum (fp32, zeroed by caller) += their sum
 * in a fixed order (no atomics: the same bits every run); dlogits (may alias logits, may be NULL) = (softmax - onehot) * grad_scale / count. */
int egomi_ce_count(const int64_t* targets, int64_t n, int64_t ignore, int32_t* count, egomi_stream_t stream);
int egomi_ce_fwd_bwd(const void* logits, int64_t ld, const int64_t* targets, int R, int V, int64_t ignore, const int32_t* count,
                     float* loss_sum, float* row_loss, void* dlogits, int64_t ldd, float grad_scale, int dtype, egomi_stream_t stream);

/* AdamW step (torch.optim.AdamW semantics; reference optimizer models/pointllm/train.py:107-111).
 * fp32 master/moments/grad; model_copy (copy_dtype, may be NULL) receives the updated value. */
int egomi_adamw(float* master, void* model_copy, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                float beta2, float eps, float weight_decay, int step, float grad_scale, int copy_dtype, egomi_stream_t stream);
/* The same step with the gradient given in bf16: the rank-summed wire buffer of the data-parallel exchange read in place (what DeepSpeed's
 * bf16 engine does after its reduce: fp32 master / moments updated from the reduced low-precision gradient, train.py:92-104,184). */
int egomi_adamw_g16(float* master, void* model_copy, const void* grad_bf16, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, int copy_dtype, egomi_stream_t stream);

/* layout helpers: out[c, r] = in[r, c] for r < R, zero for R <= r < ldo; cast; add */
int egomi_transpose(const void* in, int R, int C, int64_t ldi, void* out, int64_t ldo, int dtype, egomi_stream_t stream);
int egomi_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, egomi_stream_t stream);
int egomi_add(const void* a, const void* b, void* out, int64_t n, int dtype, egomi_stream_t stream);
/* Gradient exchange of the data-parallel step (replaces DeepSpeed's bucketed gradient reduction behind
 * model_engine.backward/step, train.py:92-125,183-184): the reduce half of a direct reduce-scatter.  in [W, c] holds the
 * chunk each of the W ranks sent to this one (all-to-all over xGMI, bf16 on the wire); out [c] = their sum accumulated in
 * fp32 in rank order.  c % 8 == 0, pointers 16-B aligned.  (in,out) dtypes: (bf16,bf16) (bf16,f32) (f32,f32). */
int egomi_rank_sum(const void* in, int in_dtype, int W, int64_t c, void* out, int out_dtype, egomi_stream_t stream);
/* out[c] (fp32, caller-initialised) += sum_r x[r,c]: bias gradients (backward of nn.Linear bias, model/pointllm.py:67-81) */
int egomi_colsum(const void* x, int64_t R, int C, int64_t ld, float* out, int dtype, egomi_stream_t stream);

/* A6 helpers.  group_max: x [BG, M, C] -> out [BG, C] (concat=0) or [BG*M, 2C] = [group max | x]
 * (concat=1).  replaces torch.max / cat / expand at pointbert/dvae.py:216-219.
 * linear_smallk: y = act(x.w^T + b) for K <= 8 (x may be fp32): pos_embed.0 (point_encoder.py:128)
 * and the first 1x1 conv (dvae.py:194). */
int egomi_group_max(const void* x, int BG, int M, int C, void* out, int concat, int dtype, egomi_stream_t stream);
int egomi_linear_smallk(const void* x, int x_dtype, const void* w, const void* b, void* y, int64_t R, int N, int K, int act,
                        int dtype, egomi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EGOMI_H */
