/* egovlp_hip.h -- C ABI of libegovlp_hip.so: the MI355X (gfx950) kernels behind the EgoVLP
 * EgoClip pre-training hot path.
 *
 * The reference (showlab/EgoVLP) has NO native boundary: its hot path is reached through Python
 * `nn.Module`s and every device kernel is implicit ATen/cuBLAS/cuDNN.  This header is the
 * boundary a maintainer binds instead: each entry point below replaces the implicit kernels issued
 * by the cited reference lines.  The binding is ctypes (egovlp_amd/_lib.py).
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers (HBM) unless noted; buffers are
 *    BORROWED for the duration of the enqueue (the caller -- PyTorch's caching allocator -- owns them);
 *  - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (graph-capturable:
 *    no allocation, no synchronisation, no host readback inside);
 *  - return value: 0 = ok, 1 = invalid argument (nothing enqueued), >=2 = 2 + hipError_t of the launch;
 *  - re-entrant and stateless (called from the autograd engine thread and DDP hooks as well);
 *  - bf16 tensors are raw uint16 bit patterns.  "split planes" (x_hi, x_lo): x_hi = bf16(x),
 *    x_lo = bf16(x - x_hi); x_lo may be NULL wherever `passes == 1` / documented optional;
 *  - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef EGOVLP_HIP_H
#define EGOVLP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t egv_bf16;

/* Version of this ABI (struct layouts, argument lists, mode / format codes).  egv_version() returns the number the LIBRARY was built
 * with and egv_abi_check() compares a caller's view of it -- version and the sizes of the ABI structs -- with the library's: a
 * binding generated from another header fails loudly at load time instead of passing garbage in trailing fields (round 5 grew
 * egv_block_geom and the mode bits without either).                                                                              */
#define EGV_ABI_VERSION 6

enum { EGV_ACT_NONE = 0, EGV_ACT_GELU = 1, EGV_ACT_GELU_BWD = 2, EGV_ACT_RELU_BWD = 3 };

/* ---- GEMM --------------------------------------------------------------------------------------
 * C[M,N] = alpha * A[M,K] . B[N,K]^T, then (in this order) + bias[n], activation, + residual[m,n].
 * Replaces nn.Linear / Conv2d-as-GEMM forward, dgrad and wgrad: model/video_transformer.py:41-50
 * (Mlp fc1/fc2), :70,76 (patch-embed conv), :88-89,103,135 (qkv/proj); HF DistilBERT q/k/v/out_lin,
 * ffn.lin1/lin2; model/model.py:72-79 (projections).  passes = 1: bf16 operands (hi planes only);
 * passes = 3: split-bf16 operands, fp32-grade product (three bf16 MFMA products); passes = 2: "f16x2" operands (see
 * egv_f16x2_encode: a_hi / a_lo = the two fp16 planes of a first-operand encoding, b_hi / b_lo of a second-operand encoding;
 * big-tile NT kernel only, un-split) -- the same fp32-grade product from TWO fp16 MFMA products; passes = 4: ONE fp16 MFMA product of
 * plain fp16 planes (a_hi = fp16(A), b_hi = fp16(B) = plane 1 of a second-operand f16x2 encoding; a_lo / b_lo ignored; big-tile NT
 * kernel only, un-split; epilogues: bias + residual -> fp32 / split planes, or EGV_ACT_GELU -> out_fmt 1 / 2) -- 2^-11 per operand:
 * for the Linears whose share of the 1e-3 parity budget allows it -- and, with trans == 1, the weight gradient of the
 * FP16 BACKWARD (dW = dY^T X on fp16(S dY) and the forward's fp16 activation plane, split-K allowed; alpha rescales the product --
 * 1 / (1 - 2^-6) when X is plane 1 of an f16x2 first-operand encoding -- but not the column sums).  Requirements: K % 32 == 0, N % 4 == 0,
 * lda % 8 == ldb % 8 == 0, 16-byte aligned base pointers.
 *  act = EGV_ACT_GELU      : v = gelu(v); if aux_out != NULL the pre-activation is stored there first
 *  act = EGV_ACT_GELU_BWD  : v *= gelu'(aux_in[m,n])           (fc2 dgrad -> dZ)
 *  act = EGV_ACT_RELU_BWD  : v  = aux_in[m,n] > 0 ? v : 0
 * Outputs: any subset of out_f32 / (out_hi[, out_lo]).  Split-K (ksplit > 1, used by wgrad where
 * K = #tokens): raw partial sums go to partial[ksplit][M][N] and a second kernel reduces them into
 * out_f32 (ldo must equal N; accumulate == 1 adds to the existing contents; accumulate == 2, trans == 1 only: the slabs are left
 * un-reduced for egv_splitk_reduce_multi); no epilogue then.   */
typedef struct egv_gemm_desc {
  const egv_bf16* a_hi; const egv_bf16* a_lo; int64_t lda;
  const egv_bf16* b_hi; const egv_bf16* b_lo; int64_t ldb;
  int32_t M, N, K, passes;
  float alpha;
  int32_t act;
  const float* bias;
  const float* residual; int64_t ldr;
  const float* aux_in; float* aux_out; int64_t ldaux;
  float* out_f32; int64_t ldo;
  egv_bf16* out_hi; egv_bf16* out_lo; int64_t ldoh;
  int32_t ksplit, accumulate;
  float* partial;
  int32_t trans;    /* (see below) */
  int32_t aux_bf16; /* != 0: aux_in / aux_out are bf16 [M, ldaux] instead of fp32 (the GELU pre-activation saved by fc1 for
                       fc2's dgrad: half the bytes when backward runs single-pass bf16 anyway).  Big-tile kernel only.
                       2: the bf16 buffer holds gelu'(pre-activation) itself -- EGV_ACT_GELU stores the derivative (it has
                       Phi and phi in registers) and EGV_ACT_GELU_BWD multiplies by the stored value, no second erf.
                       3: as 2 with the derivative stored as FP16 (passes == 2 / 4 launches only): the fp16 backward, where bf16's
                       2^-9 on gelu' would cap the accuracy of dZ.
                       trans = 0: A[M,K], B[N,K] (contraction index contiguous).  1 ("TN", wgrad): A is stored [K, lda] with
                       its M rows as COLUMNS and B is stored [K, ldb] with N columns, C[m,n] = sum_k A[k,m] B[k,n] --
                       no transposed copy of either operand is ever made (CDNA4 transpose-read from LDS).  Requires
                       M >= 256, N >= 256, both multiples of 8; K is arbitrary (rows past K are zero-filled).        */
  float* colsum;    /* trans == 1 only, optional: colsum[m] = sum_k A[k,m] (the bias gradient), from the same pass;
                       with ksplit > 1, partial must hold ksplit*M*N + ksplit*M floats.                              */
  int32_t grid_cap; /* persistent workgroups of the big-tile kernel for THIS launch: 0 = one per CU (256); data-parallel
                       callers pass e.g. 248 so that the RCCL kernels of the overlapped gradient all-reduce find free CUs.
                       Multiples of 8 in [8, 256]; anything else is an invalid argument.  (Per call, not per process: the
                       library keeps no state between calls.)                                                         */
  int32_t out_fmt;  /* format of (out_hi, out_lo): 0 = split-bf16 planes; 1 = the f16x2 operand format below, first-operand role
                       (two fp16 planes); 2 = ONE plane of plain fp16 in out_hi (out_lo unused): the operand of a passes == 4
                       consumer.  1 / 2: EGV_ACT_GELU of a passes == 2 or 4 product only (fc1 -> fc2 of the forward); 2 also with
                       EGV_ACT_GELU_BWD of a passes == 4 product: dZ of the fp16 backward as ONE plane of UN-CLAMPED fp16 (a scaled
                       gradient beyond fp16's range becomes inf, which the loss-scale logic answers with a skipped step).            */
  egv_bf16* out_bf; /* out_fmt != 0, optional: bf16(value) as a further plane [M, ldoh] -- the single-pass operand the backward GEMMs
                       (wgrad) read, since an fp16 plane cannot share an MFMA with bf16 gradients.                               */
} egv_gemm_desc;
int egv_gemm_nt(const egv_gemm_desc* d, void* stream);
/* The split-K reduction of SEVERAL trans == 1 launches (weight gradients) in ONE launch: a descriptor with ksplit > 1 and
 * accumulate == 2 leaves its slabs in `partial` (ksplit x M x N products, then ksplit x M column sums) and writes nothing else;
 * this call sums them: out[i][0 .. mn[i]) = sum_z partial[i][z], colsum[i][0 .. m[i]) likewise (colsum / its entries may be NULL).
 * HOST arrays of `count` <= 8 device pointers / sizes; mn[i] = M x N of launch i (a multiple of 4), ksplit[i] >= 2.  egv_block_bwd
 * finishes the six weight gradients of a SpaceTimeBlock this way when they share one stream.                                      */
int egv_splitk_reduce_multi(int32_t count, const float* const* partial, float* const* out, const int64_t* mn, const int32_t* ksplit,
                            float* const* colsum, const int32_t* m, void* stream);
/* ---- format kernels (HBM-bound) -----------------------------------------------------------------
 * fp32 [rows, cols] -> split planes, optionally also the TRANSPOSED planes t_*[cols, ldt] (ldt >= rows,
 * columns rows..ldt-1 are zero-filled so a following GEMM can contract over a K padded to 32) and the
 * column sums colsum[cols] (= bias gradient; overwritten).  Any output may be NULL.               */
int egv_split_f32(const float* x, int64_t ldx, int32_t rows, int32_t cols,
                  egv_bf16* hi, egv_bf16* lo, int64_t ldo,
                  egv_bf16* t_hi, egv_bf16* t_lo, int64_t ldt, float* colsum, void* stream);
/* The same for `count` tensors in one launch (host arrays of device pointers / sizes; no column sums): the once-per-
 * optimizer-step refresh of every weight's operand planes W[N,K] and W^T[K,N] (~100 tensors per step).
 * t_cols[i] = columns of the transposed planes tensor i owns (rows[i] <= t_cols[i] <= ldt[i]; rows[i] .. t_cols[i]-1 are
 * zero-filled) -- several tensors may share one pair of transposed planes side by side (DistilBERT's fused q/k/v weight). */
int egv_split_f32_multi(int32_t count, const float* const* x, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                        egv_bf16* const* hi, egv_bf16* const* lo, const int64_t* ldo, egv_bf16* const* t_hi,
                        egv_bf16* const* t_lo, const int64_t* ldt, const int32_t* t_cols, void* stream);
/* The same with one more optional output per tensor: t16[i] (NULL entries allowed, t16 itself may be NULL) = the TRANSPOSED matrix
 * as ONE plane of plain fp16 [cols, ldt] (saturating), same geometry rules as t_hi -- W^T for the dgrad GEMMs of the fp16 backward
 * (egv_gemm_nt passes == 4 with b_hi = this plane).                                                                              */
int egv_split_f32_multi_t16(int32_t count, const float* const* x, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                            egv_bf16* const* hi, egv_bf16* const* lo, const int64_t* ldo, egv_bf16* const* t_hi,
                            egv_bf16* const* t_lo, const int64_t* ldt, const int32_t* t_cols, uint16_t* const* t16, void* stream);
/* split planes [rows, cols] -> transposed planes [cols, ldt] (+ zero pad, + colsum of hi+lo).        */
int egv_transpose_planes(const egv_bf16* hi, const egv_bf16* lo, int64_t ldx, int32_t rows, int32_t cols,
                         egv_bf16* t_hi, egv_bf16* t_lo, int64_t ldt, float* colsum, void* stream);

/* ---- the f16x2 operand format (forward GEMMs of the video tower; csrc/f16x2.h) ----------------------------------------------
 * A fp32-grade product from TWO fp16 MFMA products instead of three bf16 ones.  With e = 2^-6:
 *   first operand  (activations):  a1 = fp16((1 - e) a),  a2 = fp16(a - a1)
 *   second operand (weights):      b1 = fp16(b),          b2 = fp16(b1 + (b - b1) / e)
 *   A . B^T ~= A1 B1^T + A2 B2^T :  a2 b2 = (e a - rho)(b1 + (b - b1) / e) returns the e a b1 that a1 left out, carries b's residual
 * and cancels a1's rounding error rho; the roundings that are not compensated are attenuated by e or 2^-12 / e (~2^-17 per product:
 * 5.3e-6 on random operands where split-bf16 x3 gives 4.4e-6; embeddings 3.2e-5 from the fp32 oracle where it gives 2.7e-5).
 * An operand [rows, cols] (cols % 8 == 0) is two fp16 planes [rows, ld] (ld % 8 == 0) -- the geometry of split-bf16 planes.
 * bf (optional): bf16(x) [rows, ld], the operand of single-pass backward GEMMs.  role: 0 = first operand, 1 = second operand;
 * 2 = ONE plane of plain, UN-CLAMPED fp16 in p1 (p2 / bf unused): a scaled gradient entering the fp16 backward as fp32.
 * Replaces nothing in the reference (fp32 there); producers: this converter (weights, tests), egv_layernorm_fwd_f16x2, the
 * EGV_ACT_GELU epilogue with out_fmt = 1.  Range: fp16 (saturating at 65504; below ~4e-3 a2 loses relative, not absolute, accuracy):
 * forward operands only.                                                                                                          */
int egv_f16x2_encode(const float* x, int64_t ldx, int32_t rows, int32_t cols, uint16_t* p1, uint16_t* p2, egv_bf16* bf,
                     int64_t ldo, int32_t role, void* stream);
/* `count` tensors in one launch (HOST arrays of device pointers / sizes): the once-per-optimizer-step refresh of the weights.     */
int egv_f16x2_encode_multi(int32_t count, const float* const* x, const int64_t* ldx, const int32_t* rows, const int32_t* cols,
                           uint16_t* const* p1, uint16_t* const* p2, const int64_t* ldo, int32_t role, void* stream);
/* nn.LayerNorm (model/video_transformer.py:146,156,159 -> the qkv / fc1 Linears) with the output written in the f16x2 format,
 * first-operand role (+ optional bf16 plane); y2 == NULL: ONE plane of plain fp16 in y1 instead (the operand of a passes == 4
 * product); cols % 8 == 0, cols <= 1024; mean / rstd [rows] saved for egv_layernorm_bwd.                                          */
int egv_layernorm_fwd_f16x2(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int32_t rows,
                            int32_t cols, uint16_t* y1, uint16_t* y2, egv_bf16* ybf, int64_t ldy, float* mean, float* rstd,
                            void* stream);

/* ---- one call per SpaceTimeBlock ----------------------------------------------------------------------------------
 * SpaceTimeBlock.forward (model/video_transformer.py:163-177: t = timeattn(norm3(x)); tr = x + t; s = attn(norm1(tr)); sr = x + s;
 * out = sr + mlp(norm2(sr))) and its backward, enqueued from C: the same kernels with the same arguments in the same order as the
 * per-kernel entry points above (LayerNorm -> qkv GEMM -> divided attention -> proj GEMM, twice, then fc1 / GELU / fc2), with every
 * intermediate in ONE caller-provided arena per direction -- what costs the host ~50 tensor allocations and ~30 calls per block
 * otherwise.  Token-major [M = B (1 + T n), D] fp32 residual stream; D = 64 H.
 * fwd_passes / bwd_passes: 3 = split-bf16 three-product, 1 = single-pass bf16 (bwd_passes <= fwd_passes); train != 0 keeps what
 * the backward needs (z_bf16 != 0: fc1 saves gelu'(z) as bf16 -- single-pass backward -- instead of the fp32 pre-activation).
 * The forward arena must stay untouched until egv_block_bwd has run on it.                                                     */
typedef struct egv_block_geom {
  int32_t B, T, n, H, D, Hd;
  int32_t fwd_passes, bwd_passes, train, z_bf16;
  float eps;
  int32_t grid_cap;     /* as egv_gemm_desc.grid_cap */
  int32_t f16_single;   /* fwd_passes == 2 only: which Linears of THIS block run ONE fp16 product (egv_gemm_nt passes == 4) instead of
                           the two of the f16x2 format -- bit 0: fc1 (norm2 then writes one plain fp16 plane), bit 1: fc2 (the GELU
                           epilogue of fc1 then writes h as one plain fp16 plane), bit 2: both qkv Linears (norm3 / norm1 write one
                           plain fp16 plane), bit 3: both proj Linears (the attention kernels write fp16(value) as their second
                           output plane; w_hi[1] / w_hi[3] are then the weights' f16x2 encodings like those of the other single-
                           product Linears).  0 elsewhere.  Which blocks may is the caller's precision policy.        */
} egv_block_geom;
typedef struct egv_block_params {                 /* weight index: 0 timeattn.qkv, 1 timeattn.proj, 2 attn.qkv, 3 attn.proj, 4 fc1, 5 fc2 */
  const float *n3w, *n3b, *n1w, *n1b, *n2w, *n2b; /* LayerNorm affine (norm3 = temporal, norm1 = spatial, norm2 = MLP)              */
  const float* bias[6];
  const egv_bf16 *w_hi[6], *w_lo[6]; int64_t ldw[6];     /* W[N,K] planes (forward)                                              */
  const egv_bf16 *wt_hi[6], *wt_lo[6]; int64_t ldwt[6];  /* W^T[K,N] planes (dgrad; may be NULL for egv_block_fwd)               */
} egv_block_params;
int64_t egv_block_fwd_arena_bytes(const egv_block_geom* g);
/* byte offsets into the forward arena of: n3_hi, timeattn-out hi, n1_hi, attn-out hi, n2_hi, h_hi, qkv_t hi, qkv_s hi, tr, sr, z   */
int egv_block_fwd_offsets(const egv_block_geom* g, int64_t* off11);
int egv_block_fwd(const egv_block_geom* g, const egv_block_params* p, const float* x, float* out, void* arena, void* stream);
/* Backward.  g_out: dL/d out fp32 [M, D]; g_hi / g_lo: the same as planes if the caller has them (else NULL: split here).
 * Outputs: d_x fp32 [M, D] and its planes dx_hi[, dx_lo]; `grads`: ONE fp32 buffer holding dW x 6, db x 6 and the three
 * LayerNorms' (dgamma, dbeta) at the offsets of egv_block_grad_layout (order: weights 0..5, biases 0..5, norm3 g/b, norm1 g/b,
 * norm2 g/b).  Weight-gradient GEMM i runs on side_stream[i] behind side_event[i] recorded on `stream` (NULL: on `stream`) with
 * wgrad_ksplit[i] k-slices (slabs in the backward arena); the caller joins the side streams.                                     */
typedef struct egv_block_bwd_io {
  const float* g_out; const egv_bf16 *g_hi, *g_lo;
  const float* x; const void* fwd_arena; void* bwd_arena;
  float* d_x; egv_bf16 *dx_hi, *dx_lo;
  float* grads;
  void* side_stream[6]; void* side_event[6];
  int32_t wgrad_ksplit[6];
} egv_block_bwd_io;
int64_t egv_block_bwd_arena_bytes(const egv_block_geom* g, const int32_t* wgrad_ksplit6);
int egv_block_grad_layout(const egv_block_geom* g, int64_t* offsets18, int64_t* total_floats);
int egv_block_bwd(const egv_block_geom* g, const egv_block_params* p, const egv_block_bwd_io* io, void* stream);

/* ---- one call per DistilBERT TransformerBlock -----------------------------------------------------------------
 * HF modeling_distilbert.py TransformerBlock.forward (:227-259; built by the reference at model/model.py:31-36 and called at :122):
 * sa = LN(out_lin(MHA(x)) + x); out = LN(lin2(gelu(lin1(sa))) + sa), and its backward, enqueued from C on ONE stream: the same
 * kernels with the same arguments in the same order as the per-kernel entry points (egv_split_f32, egv_gemm_nt, egv_text_attn_*,
 * egv_layernorm_*, egv_dropout).  The split-K factors of the M = B*L GEMMs come from the caller (>= 1 each): nt_ksplit_fwd for the
 * q/k/v (fused), out_lin, lin1, lin2 products, nt_ksplit_bwd for dZ, d_sa, d_ctx, d_x in that order, wgrad_ksplit per weight.
 * Weight index: 0 = q/k/v fused [3D, D] (rows q | k | v), 1 = out_lin, 2 = lin1, 3 = lin2.  attn_p / ffn_p: the dropout
 * probabilities (0 outside train()) with their seeds, seed_dev as in egv_text_attn_fwd.                                          */
typedef struct egv_text_geom {
  int32_t B, L, H, D, Hd;
  int32_t fwd_passes, bwd_passes, train;
  float eps, attn_p, ffn_p;
  int32_t grid_cap;
  uint64_t attn_seed, ffn_seed;
  const uint64_t* seed_dev;
  int32_t nt_ksplit_fwd[4], nt_ksplit_bwd[4], wgrad_ksplit[4];
} egv_text_geom;
typedef struct egv_text_params {
  const float *ln1w, *ln1b, *ln2w, *ln2b;          /* sa_layer_norm, output_layer_norm                                     */
  const float* bias[4];
  const egv_bf16 *w_hi[4], *w_lo[4]; int64_t ldw[4];     /* W[N,K] planes (forward)                                       */
  const egv_bf16 *wt_hi[4], *wt_lo[4]; int64_t ldwt[4];  /* W^T[K,N] planes (dgrad; may be NULL for the forward call)     */
} egv_text_params;
int64_t egv_text_layer_fwd_arena_bytes(const egv_text_geom* g);
int egv_text_layer_fwd(const egv_text_geom* g, const egv_text_params* p, const float* x /* [B*L, D] */, const int64_t* mask /* [B, L] */,
                       float* out, void* arena, void* stream);
int64_t egv_text_layer_bwd_arena_bytes(const egv_text_geom* g);
/* offsets (floats) into `grads` of: dW x 4, db x 4, sa_layer_norm dgamma, dbeta, output_layer_norm dgamma, dbeta               */
int egv_text_layer_grad_layout(const egv_text_geom* g, int64_t* offsets12, int64_t* total_floats);
int egv_text_layer_bwd(const egv_text_geom* g, const egv_text_params* p, const float* g_out, const int64_t* mask, const void* fwd_arena,
                       void* bwd_arena, float* d_x, float* grads, void* stream);

/* ---- LayerNorm ----------------------------------------------------------------------------------
 * nn.LayerNorm over the last dim (video eps 1e-6: model/video_transformer.py:146,156,159,228,253;
 * DistilBERT eps 1e-12).  Optional fused pre-add: the normalised input is x + x_add (DistilBERT's
 * post-LN `LN(sublayer + x)`), and that sum is stored to sum_out if non-NULL.  Outputs: split planes
 * and/or fp32; mean/rstd [rows] are saved for backward.                                            */
int egv_layernorm_fwd(const float* x, const float* x_add, int64_t ldx, const float* gamma, const float* beta,
                      float eps, int32_t rows, int32_t cols, float* sum_out,
                      egv_bf16* y_hi, egv_bf16* y_lo, float* y_f32, int64_t ldy,
                      float* mean, float* rstd, void* stream);
/* dx[r,:] = (add1 + add2)[r,:] + LN'(dy; x, gamma, mean, rstd)[r,:];  dgamma/dbeta [cols] overwritten.
 * dy is given EITHER as fp32 (dy) OR as split-bf16 planes (dy_hi[, dy_lo]; dy == NULL) -- the dgrad GEMM that produces
 * it can write planes directly; lddy is the leading dimension of whichever is used.
 * dx_hi/dx_lo (optional, contiguous [rows, cols]): the same dx as split-bf16 planes, i.e. already in the operand
 * format of the dgrad / wgrad GEMMs that consume it.  `work`: 2 * cols * egv_layernorm_bwd_parts(rows) floats.     */
int egv_layernorm_bwd_parts(int32_t rows);
int egv_layernorm_bwd(const float* dy, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy,
                      const float* x, int64_t ldx, const float* gamma,
                      const float* mean, const float* rstd, int32_t rows, int32_t cols,
                      const float* add1, const float* add2, float* dx, int64_t lddx,
                      egv_bf16* dx_hi, egv_bf16* dx_lo, float* dgamma, float* dbeta, float* work, void* stream);
/* The same with the plane formats as an argument.  dx_fmt bit 0: 0 = dx planes are split-bf16 (dx_hi[, dx_lo]); 1 = ONE plane of
 * UN-CLAMPED fp16 in dx_hi (dx_lo must be NULL): the operand of the next dgrad / wgrad GEMMs of the fp16 backward.  dx_fmt bit 1 (+ 2):
 * dy is ONE plane of un-clamped fp16 in dy_hi (dy and dy_lo NULL) -- what the dgrad GEMM in front writes in the fp16 backward
 * (egv_gemm_desc.out_fmt 4): half the bytes of an fp32 dy on both sides, one more fp16 rounding of a scaled gradient.                */
int egv_layernorm_bwd_fmt(const float* dy, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy,
                          const float* x, int64_t ldx, const float* gamma,
                          const float* mean, const float* rstd, int32_t rows, int32_t cols,
                          const float* add1, const float* add2, float* dx, int64_t lddx,
                          egv_bf16* dx_hi, egv_bf16* dx_lo, int32_t dx_fmt, float* dgamma, float* dbeta, float* work, void* stream);

/* The two stages of the backward as separate entry points: egv_layernorm_bwd_partial = everything but the reduction of the per-block
 * partial sums (dx [+ planes] are final; dgamma / dbeta are zeroed, `work` holds the partials), egv_layernorm_bwd_reduce = ONE launch
 * that finishes up to four such backwards of the same (rows, cols) (HOST arrays of `count` device pointers) -- egv_block_bwd reduces
 * the three LayerNorms of a SpaceTimeBlock with one launch instead of three.                                                       */
int egv_layernorm_bwd_partial(const float* dy, const egv_bf16* dy_hi, const egv_bf16* dy_lo, int64_t lddy,
                              const float* x, int64_t ldx, const float* gamma,
                              const float* mean, const float* rstd, int32_t rows, int32_t cols,
                              const float* add1, const float* add2, float* dx, int64_t lddx,
                              egv_bf16* dx_hi, egv_bf16* dx_lo, int32_t dx_fmt, float* dgamma, float* dbeta, float* work, void* stream);
int egv_layernorm_bwd_reduce(int32_t count, const float* const* work, int32_t rows, int32_t cols, float* const* dgamma,
                             float* const* dbeta, void* stream);

/* ---- video tokens -------------------------------------------------------------------------------
 * Patch gather for the 16x16/s16 conv (model/video_transformer.py:70-77): video [B*T,C,H,W] fp32 ->
 * A[(bt*gh + py)*gw + px][c*P*P + i*P + j] split planes, K = C*P*P (a multiple of 32 for P=16/C=3).   */
int egv_patch_gather(const float* video, int32_t BT, int32_t C, int32_t H, int32_t W, int32_t P,
                     egv_bf16* a_hi, egv_bf16* a_lo, int64_t lda, void* stream);
/* The same gather straight from decoded uint8 frames [B*T,C,H,W]: ToTensor's x/255 and
 * Normalize's (x - mean[c]) / std[c] (data_loader/transforms.py:38-39; base/base_dataset.py `frames.float() / 255`) are
 * applied in the kernel, fp32, same operation order = bit-identical planes; `mean` / `std` are HOST arrays of C <= 4 floats. */
int egv_patch_gather_u8(const uint8_t* video, int32_t BT, int32_t C, int32_t H, int32_t W, int32_t P,
                        const float* mean, const float* std, egv_bf16* a_hi, egv_bf16* a_lo, int64_t lda, void* stream);
/* The same gather with the TRAIN transform of the loader fused in (data_loader/transforms.py:14-19): per clip a crop box
 * (top, left, h, w) and a flip flag -- boxes[B][5] int32 on the DEVICE, the host's random draws -- select the region of the
 * decoded uint8 clip [B*T, C, Hs, Ws] that is resized (bilinear, align_corners = False semantics, on x / 255) to R x R,
 * mirrored when flip != 0, normalised and written as patch planes (R % P == 0, R % 4 == 0).  The box must lie inside the frame. */
int egv_patch_gather_u8_aug(const uint8_t* video, int32_t BT, int32_t T, int32_t C, int32_t Hs, int32_t Ws, int32_t R,
                            int32_t P, const int32_t* boxes, const float* mean, const float* std, egv_bf16* a_hi,
                            egv_bf16* a_lo, int64_t lda, void* stream);
/* x[b,0,:] = cls + pos[0]; x[b,1+f*n+i,:] = pe[(b*T+f)*n+i,:] + pos[1+i] + temporal[f]
 * (model/video_transformer.py:305-320; pos tiling by the MODEL's num_frames, sliced to T).          */
int egv_assemble_tokens(const float* pe, const float* cls, const float* pos, const float* temporal,
                        int32_t B, int32_t T, int32_t n, int32_t D, float* x, void* stream);
/* backward of the above: d_pe (gather), d_cls, d_pos [n+1,D], d_temporal [T_model,D] (rows >= T zeroed). */
int egv_assemble_tokens_bwd(const float* dx, int32_t B, int32_t T, int32_t n, int32_t D, int32_t T_model,
                            float* d_pe, float* d_cls, float* d_pos, float* d_temporal, void* stream);

/* ---- divided space-time attention ------------------------------------------------------------------
 * VarAttention core (model/video_transformer.py:104-133) on the fused qkv buffer [B, S, 3, H, 64] given as split-bf16
 * planes (exactly what the qkv GEMM epilogue writes; qkv_lo is ignored / may be NULL when passes == 1).
 * S = 1 + T*n, token order 1 + f*n + i.  q is scaled by 64^-0.5 inside.  mode 0 = space (group = (b,f,h): n queries x
 * (CLS + n) keys, bf16 MFMA, scores never leave LDS / registers), mode 1 = time (group = (b,i,h): T queries x (CLS + T)
 * keys, bf16 MFMA as well).  The CLS query row (attends to all S keys, :112) rides in every group as an extra query; its
 * partials are merged by a small combine kernel.  Output: split planes [B, S, H*64]; lse [B, H, S] (log-sum-exp of each
 * query row, saved for backward).  `work`: egv_divided_attn_fwd_work_floats(...) floats.
 * mode bits 1-2 (mode = 2 fmt + (0 space | 1 time); fmt != 0 with passes == 3 only) = the format of the output planes:
 *   0  split-bf16 (out_hi = bf16(v), out_lo = bf16(v - out_hi)): a three-product proj;
 *   1  out_hi = bf16(v) (what a bf16 backward reads), out_lo = fp16(v): the operand of a proj Linear that runs ONE fp16 product;
 *   2  the f16x2 operand format, first-operand role (out_hi = a1, out_lo = a2): a TWO-product proj whose backward is fp16;
 *   3  out_hi = fp16(v), out_lo unused (may be NULL): a one-product proj whose backward is fp16 as well.                 */
int egv_divided_attn_fwd(const egv_bf16* qkv_hi, const egv_bf16* qkv_lo, int32_t B, int32_t T, int32_t n, int32_t H,
                         int32_t mode, int32_t passes, egv_bf16* out_hi, egv_bf16* out_lo, float* lse, float* work,
                         void* stream);
int64_t egv_divided_attn_fwd_work_floats(int32_t B, int32_t T, int32_t n, int32_t H, int32_t mode);
/* Backward: out_* = the forward output planes, dout_* = gradient w.r.t. them (planes, e.g. from the proj dgrad
 * epilogue).  dqkv [B,S,3,H,64] is written ONCE as split planes -- every dK / dV row already contains the CLS query's
 * contribution -- i.e. directly in the operand format of the qkv dgrad / wgrad GEMMs.  Only the CLS token's own
 * gradients are accumulated in fp32 (atomics) and converted by a finish kernel.
 * `work`: egv_divided_attn_bwd_work_floats(...) floats.
 * mode: bit 0 = time; bits 1-2 = the format the forward wrote out_* in (as above: delta = rowsum(dO o O) decodes it; formats 1-3
 * take out_lo = NULL); bit 3 (passes == 1 only) = dqkv_hi receives ONE plane of UN-CLAMPED fp16 instead of bf16 (the fp16 backward;
 * q / k / v / dO are still read as bf16 planes).                                                                        */
int egv_divided_attn_bwd(const egv_bf16* qkv_hi, const egv_bf16* qkv_lo, const egv_bf16* out_hi, const egv_bf16* out_lo,
                         const egv_bf16* dout_hi, const egv_bf16* dout_lo, const float* lse, int32_t B, int32_t T,
                         int32_t n, int32_t H, int32_t mode, int32_t passes, egv_bf16* dqkv_hi, egv_bf16* dqkv_lo,
                         float* work, void* stream);
int64_t egv_divided_attn_bwd_work_floats(int32_t B, int32_t T, int32_t n, int32_t H);

/* ---- DistilBERT pieces ----------------------------------------------------------------------------
 * Embeddings (modeling_distilbert.py:82-118): e[b,l,:] = word[ids[b,l]] + pos[l] (fp32 sum; LN is a
 * separate egv_layernorm_fwd).  Backward scatters d_e into d_word (atomic adds; d_word/d_pos must be
 * zero-initialised by the caller) and reduces d_pos.                                               */
int egv_embed_fwd(const int64_t* ids, const float* word, const float* pos, int32_t B, int32_t L, int32_t D,
                  float* e, void* stream);
int egv_embed_bwd(const int64_t* ids, const float* d_e, int32_t B, int32_t L, int32_t D, int64_t pad_id /* -1: none */,
                  float* d_word, float* d_pos, void* stream);
/* Masked multi-head attention (modeling_distilbert.py:122-203) on separate q,k,v [B, L, H*64] fp32;
 * mask [B, L] int64 (0 = padded key -> -inf).  Output split planes [B, L, H*64]; probs are recomputed
 * in backward from lse [B,H,L].                                                                     */
int egv_text_attn_fwd(const float* q, const float* k, const float* v, int64_t ldqkv, const int64_t* mask, int32_t B,
                      int32_t L, int32_t H, int32_t passes, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                      egv_bf16* out_hi, egv_bf16* out_lo, float* lse, void* stream);
/* q, k, v (and dq, dk, dv) rows are ldqkv (lddqkv) floats apart: H*64 for separate tensors, 3*H*64 when they are the
 * three column blocks of one fused [B*L, 3*H*64] projection output (one GEMM instead of three).
 * dropout_p > 0: HF's attention dropout (softmax -> dropout -> . V, modeling_distilbert.py eager attention) with the
 * counter-based mask keep(seed, ((b*H + h)*L + query)*L + key); the backward regenerates it from the same (dropout_p, seed).
 * seed_dev (optional, DEVICE, one uint64): XOR-ed into `seed` by the kernel when it runs -- the part of the seed that changes
 * from replay to replay of a step captured into a HIP graph (launch arguments are frozen at capture).                      */
int egv_text_attn_bwd(const float* q, const float* k, const float* v, int64_t ldqkv, const int64_t* mask,
                      const float* d_out, const float* lse, int32_t B, int32_t L, int32_t H, int32_t passes,
                      float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* dq, float* dk, float* dv,
                      int64_t lddqkv, float* delta_work /* B*H*L floats */, void* stream);
/* Zero-fill (hipMemsetAsync on `stream`) of a buffer the caller has just allocated.                                    */
int egv_zero(void* p, int64_t bytes, void* stream);
/* Elementwise dropout of DistilBERT (embedding output, FFN output): out[i] = x[i] * M'(i) + (add ? add[i] : 0) with
 * M'(i) = keep(seed, i) ? 1 / (1 - p) : 0.  The same call with x = dy is its backward.  16-byte aligned pointers.      */
int egv_dropout(const float* x, const float* add, float* out, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev,
                void* stream);

/* ---- contrastive head ---------------------------------------------------------------------------------
 * sim_matrix x3 + EgoNCE/NormSoftmaxLoss forward AND backward in one launch
 * (model/model.py:189-197; model/loss.py:13-25,34-53; trainer/trainer_egoclip.py:130-137).
 * text, video [n, D] fp32 (the all-gathered global batch); noun [n, dn], verb [n, dv] multi-hot fp32
 * (NULL for NormSoftmaxLoss: mask = I).  Writes loss[1], sim [n,n] (optional), and the gradients of
 * the loss w.r.t. text and video [n, D] (optional).  n <= 1024.  `use_noun/use_verb` mirror EgoNCE's ctor. */
int egv_egonce_fwd_bwd(const float* text, const float* video, const float* noun, const float* verb,
                       int32_t n, int32_t D, int32_t dn, int32_t dv, float temperature, float eps,
                       int32_t use_noun, int32_t use_verb,
                       float* loss, float* sim, float* d_text, float* d_video, float* work, void* stream);
int64_t egv_egonce_work_floats(int32_t n, int32_t D);
/* The same head in the reference's own decomposition (kept for API compatibility with code that calls
 * model.model.sim_matrix and model.loss.EgoNCE(x, mask_v, mask_n) separately):
 *  sim_matrix forward (any n, m, D): an/bn = normalised rows (saved for backward), norms [n+m], out [n,m];
 *  sim_matrix backward: g [n,m] -> da [n,D], db [m,D] (either may be NULL);
 *  EgoNCE / NormSoftmaxLoss on a given similarity matrix x [n,n] (sim_v/sim_n NULL => mask = I): loss[1], dx. */
int egv_sim_matrix_fwd(const float* a, const float* b, int32_t n, int32_t m, int32_t D, float eps,
                       float* an, float* bn, float* norms, float* out, void* stream);
int egv_sim_matrix_bwd(const float* g, const float* an, const float* bn, const float* norms, int32_t n, int32_t m,
                       int32_t D, float eps, float* da, float* db, void* stream);
int egv_egonce_from_sim(const float* x, const float* sim_v, const float* sim_n, int32_t n, float temperature,
                        int32_t use_noun, int32_t use_verb, float* loss, float* dx,
                        float* work /* n*n + 6n floats */, void* stream);

/* Max-margin ranking losses of the fine-tuning heads (model/loss.py:55-133) on a square similarity matrix x [n, n]:
 *   loss = mean_{kept (i,j)} relu(w_i m - x_ii + x_ij) + relu(w_i m - x_ii + x_ji),   w = NULL: MaxMarginRankingLoss (w_i = 1),
 *   w = weight [n]: AdaptiveMaxMarginRankingLoss; fix_norm != 0 drops the diagonal pairs (mean over 2 n (n - 1) terms).
 * dx (optional) receives d loss / d x.  n <= 4096.                                                                       */
int egv_maxmargin_fwd_bwd(const float* x, const float* weight, int32_t n, float margin, int32_t fix_norm,
                          float* loss, float* dx, void* stream);

/* Softmax cross-entropy of the classification fine-tunes (OSCC / PNR heads): model/loss.py:135-141 (nn.CrossEntropyLoss with its
 * defaults) on the [rows, cols] scores of FrozenInTime(video_only=True) (trainer/trainer_oscc.py:335-338).
 *   loss = mean over rows with target != ignore_index of (logsumexp(x_r) - x_r[target_r]);  NaN when no row is valid (as torch);
 *   dlogits (optional, [rows, ldd]) = (softmax(x_r) - onehot(target_r)) / #valid rows, 0 for ignored rows.
 *   A target outside [0, cols) that is not ignore_index is an error (torch: device assert): loss and dlogits come back NaN.
 * Deterministic (fixed summation order).  rows <= 2^20, cols <= 65536.                                                     */
int egv_cross_entropy_fwd_bwd(const float* logits, int64_t ld, const int64_t* target, int32_t rows, int32_t cols,
                              int64_t ignore_index, float* loss, float* dlogits, int64_t ldd, void* stream);

/* Dual-softmax re-scaling of a retrieval similarity matrix x [n texts, m videos] (run/test_epic.py:137-143, --dual_softmax):
 *   y = softmax(x / temp, dim 1) * x;  out = softmax(y, dim 0).  work: n*m floats.  temp = 500 in the reference.        */
int egv_dual_softmax(const float* x, int32_t n, int32_t m, float temp, float* work, float* out, void* stream);

/* ---- gradient exchange (data parallel) ------------------------------------------------------------------
 * Replaces the fp32 bucket copies of DistributedDataParallel (base/base_trainer.py:258): `count` fp32 gradient tensors
 * (HOST arrays of device pointers / sizes) are scaled by `scale` (= 1 / world size), rounded to bf16 (RNE) and written to
 * flat[offsets[i] .. offsets[i] + numel[i]) -- the buffer RCCL all-reduces -- and read back into the fp32 tensors
 * afterwards.  offsets are in elements; multiples of 8 keep every access 16 bytes wide.                               */
int egv_grad_pack_bf16(int32_t count, const float* const* grads, const int64_t* numel, egv_bf16* flat,
                       const int64_t* offsets, float scale, void* stream);
int egv_grad_unpack_bf16(int32_t count, float* const* grads, const int64_t* numel, const egv_bf16* flat,
                         const int64_t* offsets, void* stream);
/* The local reduction of the DIRECT gradient exchange (all-to-all of bucket slices over all xGMI links, then all-gather; SURVEY
 * 5 / 8(e): xGMI is point to point, a ring keeps 5 of the 7 links idle): recv holds `world` slices of slice_elems bf16 each (this
 * rank's slice of every peer's bucket); out[i] = bf16(sum_p float(recv[p][i])) -- fp32 accumulation, ONE rounding.
 * slice_elems % 8 == 0, 16-byte aligned pointers, world <= 64.                                                            */
int egv_slice_sum_bf16(const egv_bf16* recv, int32_t world, int64_t slice_elems, egv_bf16* out, void* stream);

/* ---- optimizer ----------------------------------------------------------------------------------------
 * transformers==4.2.1 AdamW (run/train_egoclip.py:73, configs/pt/egoclip.json:49-54) over a list of
 * tensors given as HOST arrays of device pointers (copied into kernel arguments in chunks), fused with
 * the refresh of the split-bf16 weight planes the GEMMs read (w_hi/w_lo may be NULL per tensor).
 * hyper_dev (optional, DEVICE, 4 floats {lr, step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t), grad_scale, skip}): when given,
 * the kernel takes the step-dependent scalars from there instead of lr / step / grad_scale, and does NOTHING when skip != 0 --
 * the block egv_loss_scale_update writes on the same stream (dynamic loss scale of the fp16 backward: whether the step is applied,
 * the 1 / S that un-scales the gradients and the bias correction at the number of APPLIED steps are decided on the device).   */
int egv_adamw_multi(int32_t count, float* const* p, const float* const* g, float* const* m, float* const* v,
                    egv_bf16* const* w_hi, egv_bf16* const* w_lo, const int64_t* numel,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                    int32_t correct_bias, float grad_scale, const float* hyper_dev, void* stream);

/* ---- dynamic loss scale (the fp16 backward) ---------------------------------------------------------------
 * The reference back-propagates in fp32 (trainer/trainer_egoclip.py:139-141); here the backward GEMMs of the video blocks can run on
 * fp16 operands, whose range needs the gradients scaled: the loss is multiplied by S before backward() and AdamW divides by it.
 * S lives in DEVICE memory and follows torch.cuda.amp.GradScaler's rule without any host synchronisation:
 *   state (8 x 32 bits): [0] float S; [1] int32 good steps since S last changed; [2] int32 found_inf; [3] int32 steps skipped so far;
 *                        [4..7] float {lr, step_size, 1 / S, skip} -- the hyper_dev block of egv_adamw_multi.
 * egv_grad_nonfinite_multi: state[2] |= 1 if any of the `count` gradient tensors (HOST arrays of device pointers / sizes) holds an
 * inf or NaN (an overflowed fp16 plane poisons every gradient behind it: gradient planes are written un-clamped).
 * egv_loss_scale_update (one thread): advance != 0: found_inf ? (skip = 1, skipped += 1, S = max(S * backoff_factor, 1), good = 0)
 * : (skip = 0, good += 1; good == growth_interval ? S = min(S * growth_factor, max_scale), good = 0), found_inf = 0, 1 / S of the
 * scale the step RAN with -> state[6]; then {lr, step_size at t = step - skipped, state[6], state[7]} -> hyper_out (NULL: state + 4;
 * advance == 0 fills a further parameter group's block from the decision already taken).                                          */
int egv_grad_nonfinite_multi(int32_t count, const float* const* grads, const int64_t* numel, int32_t* state, void* stream);
int egv_loss_scale_update(int32_t* state, float* hyper_out, float lr, float beta1, float beta2, int32_t step, int32_t correct_bias,
                          float growth_factor, float backoff_factor, int32_t growth_interval, float max_scale, int32_t advance,
                          void* stream);

/* ---- misc -----------------------------------------------------------------------------------------------
 * gather rows: out[r,:] = x[idx_stride * r * ld ...] helper for CLS-row extraction is done with strides in
 * egv_layernorm_fwd (ldx = S*D, rows = B).  relu on fp32 -> split planes: */
int egv_relu_split(const float* x, int64_t ldx, int32_t rows, int32_t cols, egv_bf16* hi, egv_bf16* lo,
                   int64_t ldo, void* stream);
/* scatter rows of a small matrix into a zeroed big one: dst[r * ld_dst + c] = src[r, c] (CLS-row grads). */
int egv_version(void);
/* 0 = the caller's header agrees with the library: abi_version == EGV_ABI_VERSION and the four struct sizes (sizeof egv_gemm_desc,
 * egv_block_geom, egv_block_params, egv_block_bwd_io) match; 1 otherwise.                                                          */
int egv_abi_check(int32_t abi_version, int64_t sizeof_gemm_desc, int64_t sizeof_block_geom, int64_t sizeof_block_params,
                  int64_t sizeof_block_bwd_io);
/* diagnostics, not on the product path: `iters` rounds of 40 independent MFMA 16x16x32 bf16 per wave, `waves` (1..8) waves
 * per workgroup, one workgroup per CU, no memory traffic -- the chip's sustained MFMA rate (tools/mfma_peak.py). */
int egv_diag_mfma_peak(int32_t iters, int32_t waves, float* out, void* stream);
/* diagnostics, not on the product path: a copy of `bytes` (multiple of 1024) from src to dst on the GEMM's own memory paths,
 * to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on a known byte count (tools/traffic_calib.py).
 * mode bits: 1 = loads by LDS-DMA (global_load_lds dwordx4, the operand path of the big GEMM; else plain 16-byte loads),
 * 2 = non-temporal stores (else write-back), 4 = read only, 8 = write only.  Valid: 0, 1, 2, 3, 4, 5, 8, 10. */
int egv_diag_traffic_calib(int32_t mode, const void* src, void* dst, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
