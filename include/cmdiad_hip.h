/*
 * cmdiad_hip.h -- C ABI of libcmdiad_hip.so: the MI355X (gfx950) kernels behind CMDIAD's hot path.
 *
 * Conventions (all entry points):
 *   - plain C: device pointers + sizes + a stream handle (hipStream_t passed as void*); no torch types.
 *   - the caller owns every buffer (inputs, outputs, workspace); the library never allocates,
 *     frees, retains a pointer past the call, or synchronises the device: all work is enqueued on
 *     `stream` and is stream-ordered.  Workspace sizes come from the *_workspace_bytes() queries.
 *   - return value: CMDIAD_OK (0) or a negative cmdiad_status; cmdiad_last_error() returns the
 *     message of the calling thread's last failure.  Nothing throws across the ABI.
 *   - one host thread per device, re-entrant across streams.
 *   - bf16 tensors are passed as uint16_t* (raw bfloat16 bits), row-major unless stated.
 *
 * Each function cites the reference interface (evenrose/CMDIAD file:line) it replaces.
 */
#ifndef CMDIAD_HIP_H
#define CMDIAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cmdiad_stream_t; /* hipStream_t */

typedef enum {
    CMDIAD_OK = 0,
    CMDIAD_ERR_ARG = -1,       /* null pointer / bad size / unsupported shape */
    CMDIAD_ERR_WORKSPACE = -2, /* workspace missing or too small */
    CMDIAD_ERR_LAUNCH = -3     /* HIP reported a launch error */
} cmdiad_status;

const char* cmdiad_last_error(void);
int cmdiad_abi_version(void); /* 2: cmdiad_reweight_scan's limits and workspace changed (see there); cmdiad_gemm_args gained m_count
                                 3: LayerNorm fold -- cmdiad_gemm_args gained row_scale / ln_xb / ln_part / add2, cmdiad_gemm_qkv gained
                                    row_scale, cmdiad_block_weights the folded weights, cmdiad_transformer_block_fwd its flags;
                                    cmdiad_ln_stats_finalize is new
                                 4: cmdiad_l2_min_keys_segments, cmdiad_gemm_streamk_*, cmdiad_coreset_greedy_f32, cmdiad_coreset_prepare / _round / _decode, cmdiad_im2col3x3_bf16 are new
                                 5: cmdiad_reweight_scan_pair is new; the value bits of cmdiad_l2_min_keys' keys are the squared distance with
                                    its four low mantissa bits cleared (see there)
                                 6: cmdiad_l2_min_keys / _counted / _segments gained keys2 (the runner-up per query, NULL = off);
                                    cmdiad_l2_rescore2 and cmdiad_l2_choose are new; cmdiad_normalize_cast turns a non-finite row into a
                                    row that cannot win */
/* 1 when the library is the test-only build that also contains the superseded kernel formulations (A/B references). */
int cmdiad_has_ab_variants(void);

/* ---------------------------------------------------------------------------------------------
 * Point-cloud front end
 * ------------------------------------------------------------------------------------------- */

/* Farthest point sampling + centre gather.
 * Replaces pointnet2_ops.furthest_point_sample(xyz[B,N,3] f32, npoint) -> int32[B,npoint] and
 * pointnet2_ops.gather_operation as called at models/models.py:76-77 (fps()).
 * xyz [B,N,3] f32; n_valid [B] int32 device array or NULL (= N points in every cloud; points at
 * index >= n_valid[b] are padding); idx_out [B,G] int32; center_out [B,G,3] f32 or NULL.
 * Bit-exact with oracle/cmdiad_oracle.c:orc_fps. */
size_t cmdiad_fps_workspace_bytes(int B, int N);
int cmdiad_fps(const float* xyz, const int32_t* n_valid, int B, int N, int G, int32_t* idx_out,
               float* center_out, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);

/* k-nearest-neighbour grouping: K nearest cloud points of every centre, ascending (d2, index),
 * plus the gathered, centre-subtracted neighbourhood.
 * Replaces knn_cuda.KNN(k, transpose_mode=True)(ref[B,N,3], query[B,G,3]) -> idx int64[B,G,k]
 * (models/models.py:86,100) and the gather/subtract at models/models.py:105-112.
 * idx_out [B,G,K] int64 (NULL allowed), neigh_out [B,G,K,3] f32 (NULL allowed).  K <= 128 and K <= N
 * (every cloud, i.e. K <= n_valid[b], must hold at least K points: the reference library has no defined result otherwise).
 * Bit-exact with oracle/cmdiad_oracle.c:orc_knn_group. */
int cmdiad_knn_group(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int G,
                     int K, int64_t* idx_out, float* neigh_out, cmdiad_stream_t stream);
/* The same result (bit for bit) by a NEIGHBOURHOOD search (ABI 6): the cloud is binned into a 64 x 64 grid on its two axes of
 * largest extent (counting sort, one workgroup per cloud), a wave scans the rings of cells around its centre and stops when the
 * K-th best distance is certified (no point outside the scanned square can be nearer) -- ~500 distance evaluations per centre
 * instead of N.  workspace: cmdiad_knn_workspace_bytes(B, N), 16-byte aligned.  Clouds of fewer than 2 048 points (and
 * CMDIAD_KNN_GRID=0) go through cmdiad_knn_group's streaming kernel. */
size_t cmdiad_knn_workspace_bytes(int B, int N);
int cmdiad_knn_group_ws(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int G, int K,
                        int64_t* idx_out, float* neigh_out, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);

/* 3-nearest-centre search + inverse-distance weights for every cloud point.
 * Replaces the selection half of models/pointnet2_utils.py:45-75 (interpolating_points):
 * d = -2*a.b + |a|^2 + |b|^2 (:19-22), three smallest (:66-67), w = 1/(d+1e-8) normalised (:69-71).
 * xyz [B,N,3], center [B,S,3] -> idx3 [B,N,3] int32, w3 [B,N,3] f32.  S <= 4096. */
int cmdiad_interp3nn(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int S,
                     int32_t* idx3, float* w3, cmdiad_stream_t stream);
/* The same idx3 / w3 (bit for bit) by a NEIGHBOURHOOD search (ABI 6): the centres are binned into a 16 x 16 grid on their two axes
 * of largest extent, a point scans the rings of cells around it until its third best formula value is certified against
 * everything unscanned (with a bound on the formula's rounding) -- ~40 evaluations per point instead of S.
 * workspace: cmdiad_interp3nn_workspace_bytes(B, S), 16-byte aligned.  S < 64 (and CMDIAD_INTERP_GRID=0) use cmdiad_interp3nn. */
size_t cmdiad_interp3nn_workspace_bytes(int B, int S);
int cmdiad_interp3nn_ws(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int S, int32_t* idx3, float* w3,
                        void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);

/* Organised -> unorganised point cloud without zeros (feature_extractors/multiple_features.py:10-25):
 * keeps, in raster order, the pixels whose x, y and z are all non-zero.
 * organized_pc [B,3,HW] f32 -> xyz [B,Nmax,3] f32 (first n_valid[b] rows valid), nz [B,Nmax] int32 pixel
 * indices (NULL allowed), pix2pt [B,HW] int32 = point index of a pixel or -1 (NULL allowed),
 * n_valid [B] int32 (NULL allowed; counts are clamped to Nmax). */
int cmdiad_unorganize(const float* organized_pc, int B, int HW, int Nmax, float* xyz, int32_t* nz,
                      int32_t* pix2pt, int32_t* n_valid, cmdiad_stream_t stream);

/* Weighted gather half of interpolating_points (pointnet2_utils.py:72):
 *   out[b][n][:] = (F[i0]*w0 + F[i1]*w1) + F[i2]*w2,  feat [B,S,D] f32 (centre-major), out [B,N,D] f32.
 * Only needed to materialise the reference's [1,D,N] `interpolated_feature_maps` for external callers;
 * the product path uses cmdiad_xyz_patch_fused and never builds it. */
int cmdiad_interp_gather(const float* feat, const int32_t* idx3, const float* w3, const int32_t* n_valid,
                         int B, int N, int S, int D, float* out, cmdiad_stream_t stream);

/* get_xyz_patch (feature_extractors/features.py:169-184) fused with the gather above: scatter into the
 * size x size map at nonzero pixels, AvgPool2d(3, stride 1), AdaptiveAvgPool2d((P,P)), reshape to
 * [P*P, D]; optionally the scalar normalisation (x - mean) * inv_std of multiple_features.py:976 and a
 * bf16 copy for the distance GEMM.  pix2pt [B, size*size] from cmdiad_unorganize.
 * patch_f32 [B,P*P,D] (NULL allowed), patch_bf16 [B,P*P,D] (NULL allowed). */
int cmdiad_xyz_patch_fused(const float* feat, const int32_t* idx3, const float* w3, const int32_t* pix2pt,
                           int B, int N, int S, int D, int size, int P, float mean, float inv_std,
                           float* patch_f32, uint16_t* patch_bf16, cmdiad_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Transformer / MLP building blocks (ViT-B/8: timm VisionTransformer reached at
 * models/models.py:41-52; Point-MAE: models/models.py:116-243; hallucination MLP:
 * utils/utils.py:86-115)
 * ------------------------------------------------------------------------------------------- */

/* out[M,N] = epilogue(A[M,K] . W[N,K]^T): bf16 MFMA, fp32 accumulate.  K % 64 == 0.
 * Epilogue, in order: + bias[N]; + group_bias[m / group_rows][N]; activation; + residual[M,N] (f32);
 * stores to out_f32 and/or out_bf16 (either may be NULL; out_f32 may alias residual). */
enum { CMDIAD_ACT_NONE = 0, CMDIAD_ACT_GELU = 1, CMDIAD_ACT_RELU = 2,
       CMDIAD_ACT_RELU_POST = 3 /* ReLU AFTER the residual; cmdiad_conv2d_nhwc_bf16 only */ };
typedef struct {
    const uint16_t* A; int lda;      /* [M,K] bf16 */
    const uint16_t* W; int ldw;      /* [N,K] bf16 (nn.Linear weight layout) */
    int M, N, K;
    const float* bias;               /* [N] or NULL */
    const float* group_bias;         /* [ceil(M/group_rows), N] or NULL */
    int group_rows;
    int act;
    const float* residual; int ldr;  /* [M,N] f32 or NULL */
    float* out_f32; int ldo32;
    uint16_t* out_bf16; int ldo16;
    uint16_t* out_pre_bf16;          /* [M,N] bf16 copy BEFORE the activation (training keeps z), or NULL */
    const uint16_t* dact_of;         /* [M,N] bf16 z: acc *= GELU'(z) before everything else, or NULL */
    int split_k;                     /* > 1: K is split; slab s of out_f32 ([split_k][M][ldo32]) gets partial s */
    const int* m_count;              /* ABI 2: device-resident live row count or NULL: only rows < min(M, *m_count) are computed
                                        and stored (the compacted row set of cmdiad_rows_dedup_plan: the launch is sized for M) */
    /* ABI 3, LayerNorm folded into the products on either side of it (models/models.py:177-180; all NULL = off):
     * consumer: out = act(row_scale[m] * acc + bias): W holds gamma o W centred over k, bias = b + W beta, A the RAW rows as
     * bf16 (ln_xb of the producer), row_scale = 1 / sigma per row from cmdiad_ln_stats_finalize; it must be readable up to
     * M rounded up to 256 rows (the values past M are never used). */
    const float* row_scale;
    /* producer, only with the in-place residual form (bias + residual -> out_f32, no activation, N % 64 == 0):
     * out_f32 = acc + bias + residual (+ add2); ln_xb [M, ld_xb] = the same rows as bf16; ln_part [N/64][M][2] f32 = (sum,
     * squared deviation from the chunk mean) of every 64-column chunk of every row. */
    uint16_t* ln_xb; int ld_xb;
    float* ln_part;
    const float* add2; int ld_add2;  /* second f32 addend [M,N] (the next block's positional embedding, models.py:240), or NULL */
} cmdiad_gemm_args;
int cmdiad_gemm_bf16(const cmdiad_gemm_args* args, cmdiad_stream_t stream);

/* Stream-K form of the in-place residual products of a transformer block (x = x + fc2(...), models/models.py:126-132,177-180):
 * out_f32 = A . W^T + bias + residual with N % 256 == 0, for shapes whose 256 x 256 tile count is between one and two per CU
 * (ViT-B/8's N = 768 products at batch 32: 297 tiles on 256 CUs).  The (tile, k-tile) list is cut into one contiguous range per
 * CU; a tile shared by two blocks is finished IN ORDER (the second block starts from the first one's parked accumulators), so the
 * result is bit-identical to cmdiad_gemm_bf16's.  Takes the cmdiad_gemm_args of the residual form only (A, W, bias, residual,
 * out_f32; everything else unset).  workspace: cmdiad_gemm_streamk_workspace_bytes() bytes, ZERO-INITIALISED once by the caller
 * and then left to the library (slots + hand-over counters; the kernel leaves the counters at zero again).  ONE stream-K launch in
 * flight PER DEVICE, not only per workspace: block b spins on block b - 1's counter, which assumes its predecessor is resident or
 * will be dispatched -- two overlapping launches (two streams) can fill every CU with waiting blocks of both.  The caller orders
 * them (cmdiad_amd.ops.gemm_streamk: an event between consecutive launches on different streams).
 * cmdiad_gemm_streamk_eligible(M, N, K) -> 1 when the shape qualifies. */
size_t cmdiad_gemm_streamk_workspace_bytes(void);
int cmdiad_gemm_streamk_eligible(int M, int N, int K);
int cmdiad_gemm_streamk_bf16(const cmdiad_gemm_args* a, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);

/* LayerNorm statistics from the producer's chunk partials (nn.LayerNorm inside Block, models/models.py:170,175): part
 * [chunks][M][2] as written through cmdiad_gemm_args.ln_part -> rstd[m] = 1 / sqrt(var_m + eps) over the chunks * 64 columns
 * (biased variance, as torch) and, optionally, mean_out[m].  Chunks are merged in order (Chan's formula): bit-reproducible. */
int cmdiad_ln_stats_finalize(const float* part, int M, int chunks, float eps, float* rstd, float* mean_out,
                             cmdiad_stream_t stream);

/* Weight-gradient product of the trainer (hallucination_network_pretrain.py:114-131, loss.backward() through
 * utils/utils.py:94-100): C[N1,N2] = sum_m P[m,n1] * Q[m,n2] with BOTH operands row-major bf16 (P = dZ [M,N1], Q = the
 * layer input [M,N2]) -- no transposed copies.  M % 64 == 0, N1 % 8 == 0, N2 % 8 == 0.  split_k > 1: grid.y slices of M, slab s
 * of out_f32 ([split_k][N1][ldo]) receives partial sum s (cmdiad_reduce_slabs adds them); every slab is fully written.
 * colsum_out (NULL allowed) [split_k][N1] f32: partial column sums of P, i.e. the bias gradient sum_m dZ[m, n1], from the
 * tiles already staged for the product. */
int cmdiad_gemm_tn_bf16(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, int M, int N1, int N2, int split_k,
                        float* out_f32, int ldo, float* colsum_out, cmdiad_stream_t stream);

/* QKV projection with head-split stores for the attention kernel (models/models.py:150-151):
 * A [B*T, C] bf16, W [3C, C], bias [3C] or NULL.  head_dim = 64, C = H*64.
 * q_out, k_out [B,H,Tp,64] bf16 (q pre-multiplied by head_dim^-0.5 (models.py:153) times log2(e)),
 * vt_out [B,H,64,Tp] bf16 (V transposed); Tp = T rounded up to 64; padding is never written.
 * row_scale [B*T] f32 or NULL (ABI 3): qkv = row_scale[m] * (A . W^T) + bias, the consumer side of the LayerNorm fold
 * (cmdiad_gemm_args.row_scale). */
int cmdiad_gemm_qkv(const uint16_t* A, const uint16_t* W, const float* bias, const float* row_scale, int B, int T, int C,
                    uint16_t* q_out, uint16_t* k_out, uint16_t* vt_out, cmdiad_stream_t stream);

/* softmax(q k^T) v per (image, head) (models/models.py:153-157), flash-style, bf16 MFMA; q as written by
 * cmdiad_gemm_qkv (pre-scaled by head_dim^-0.5 * log2(e): the kernel evaluates exp2 of the raw q.k).
 * out [B*T, C] bf16 with heads concatenated along C (the `.transpose(1,2).reshape(B,N,C)` of :157). */
int cmdiad_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, int B, int H, int T,
                     uint16_t* out, cmdiad_stream_t stream);

/* y = LayerNorm(x (+ add)) * gamma + beta -> bf16; when `add` is given (Point-MAE re-adds the
 * positional embedding in front of every block, models/models.py:240) x is updated in place to
 * x + add first.  x [M,C] f32, add [M,C] f32 or NULL, C % 64 == 0, C <= 1024.
 * out_bf16 [M, C] (NULL allowed); out_f32 (NULL allowed) with leading dimension ldo32. */
int cmdiad_layernorm(float* x, const float* add, const float* gamma, const float* beta, float eps, int M,
                     int C, uint16_t* out_bf16, float* out_f32, int ldo32, float* mean_out, float* rstd_out,
                     cmdiad_stream_t stream);

/* Second half of the Point-MAE encoder in one kernel (models/models.py:204-215): h3 = ReLU(W3b . h2 + gb[group]) is produced
 * and consumed in LDS, tok[g] = max over the group's rows of (W4 . h3 + b4).  h2 [groups*Mg, 256] bf16 and gb [groups, 512] f32
 * (= W3a . groupmax(h2) + b3) as produced by cmdiad_encoder_stage1 + cmdiad_gemm_bf16; W3b [512,256], W4 [384,512] bf16;
 * tok_out [groups, 384] f32.  Bit-identical to cmdiad_gemm_bf16(ReLU, group_bias) followed by cmdiad_gemm_groupmax. */
int cmdiad_encoder_tail(const uint16_t* h2, const float* gb, const uint16_t* W3b, const uint16_t* W4, const float* b4,
                        int groups, int Mg, float* tok_out, cmdiad_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution heads of the distillation networks: models/hallucination_network.py:72-143
 * (HallucinationCrossModalityConv), 146-182 (HallucinationRGBFeatureToXYZInputMLP), 185-220
 * (HallucinationFeatureToInputConv), models/hrnet.py:8-43,146-290 (HRNet stem + Bottlenecks).
 * Replaces the torch.nn.Conv2d / BatchNorm2d / interpolate calls those modules make (cuDNN in the reference).
 * ------------------------------------------------------------------------------------------- */

/* 3x3 (stride 1 or 2, padding 1) or 1x1 convolution as an implicit GEMM: x [B,H,W,C] bf16 NHWC (a token matrix
 * [B, H*W, C] is exactly that), W [N][ksize*ksize][C] bf16 (tap-major: the torch weight [N,C,kh,kw] permuted to
 * [N,kh,kw,C]; BatchNorm folded by the caller), C % 64 == 0, N % 4 == 0.  Output rows m = (b, yo, xo) with
 * Ho = (H + 2 pad - ksize) / stride + 1.  Epilogue: + bias[N]; ReLU (act = RELU); + residual[M,N] f32; ReLU
 * (act = RELU_POST); stores to out_f32 and/or out_bf16 with leading dimensions ldo32 / ldo16 (columns >= N untouched). */
typedef struct {
    const uint16_t* x; int B, H, Wd, C;
    const uint16_t* W; int N;
    int ksize, stride;
    const float* bias;               /* [N] or NULL */
    int act;                         /* CMDIAD_ACT_NONE / RELU / RELU_POST */
    const float* residual; int ldr;  /* [M,N] f32 or NULL */
    float* out_f32; int ldo32;
    uint16_t* out_bf16; int ldo16;
} cmdiad_conv_args;
int cmdiad_conv2d_nhwc_bf16(const cmdiad_conv_args* args, cmdiad_stream_t stream);

/* Stem convolution (hrnet.py:150-151,252-254): 3x3, padding 1, stride 1 or 2 on an f32 NCHW image with Cin <= 4 planes,
 * folded BatchNorm bias, ReLU, bf16 NHWC output [B,Ho,Wo,Cout].  w [Cout][Cin][3][3] f32 (torch layout), Cout % 8 == 0. */
int cmdiad_conv_stem(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout,
                     int stride, uint16_t* out, cmdiad_stream_t stream);

/* torch.nn.functional.interpolate(mode='bicubic', align_corners=False) (hallucination_network.py:171,204).
 * in f32 NHWC [B,h,w,ldi] of which C channels are used; exactly one output: out_bf16_nhwc [B,H,W,ldo] (the next
 * convolution's operand; columns >= C untouched) or out_f32_nchw [B,C,H,W]. */
int cmdiad_upsample_bicubic(const float* in, int B, int h, int w, int C, int ldi, int H, int W,
                            uint16_t* out_bf16_nhwc, int ldo, float* out_f32_nchw, cmdiad_stream_t stream);

/* One whole pre-LN transformer block on the fp32 residual stream x [B*T, C], in place (models/models.py:177-180 Block.forward,
 * 148-160 Attention, 126-132 Mlp; timm's ViT block reached at models.py:48 has the same algebra):
 *   x += proj(softmax(q k^T) v) with q,k,v = qkv(LN1(x (+ pos)));   x += fc2(GELU(fc1(LN2(x)))).
 * pos [B*T, C] f32 or NULL is added to x first (Point-MAE re-adds the positional embedding in front of every block,
 * models.py:240).  Weights: bf16 [out, in] matrices in nn.Linear layout, f32 biases / LayerNorm parameters (qkv_b may be
 * NULL).  C = 64 H.  q, k [B,H,Tp,64] and vt [B,H,64,Tp] bf16 scratch as in cmdiad_gemm_qkv (padding rows zeroed once by
 * the caller); workspace >= cmdiad_transformer_block_workspace_bytes(B*T, C, hidden).  Sequencing only: LayerNorm ->
 * cmdiad_gemm_qkv -> cmdiad_attention -> cmdiad_gemm_bf16 x3, all on `stream`.
 * ABI 3: with the folded weights present (qkv_wf ... fc1_bf: gamma o W centred over k as bf16, b + W beta; see
 * cmdiad_gemm_args.row_scale) the second LayerNorm is folded into proj (producer) and fc1 (consumer), and with
 *   CMDIAD_BLOCK_PREP_NEXT  fc2 also emits the bf16 rows + statistics of its output (+ pos: the NEXT block's input) into the
 *                           workspace, which the next call on the same workspace consumes when it is given
 *   CMDIAD_BLOCK_LN1_READY  (the first LayerNorm is folded into qkv; pos is NOT added again).
 * A block whose output is read before the next block (Point-MAE's fetch layers) is called without PREP_NEXT. */
typedef struct {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const uint16_t* qkv_w; const float* qkv_b;
    const uint16_t* proj_w; const float* proj_b;
    const uint16_t* fc1_w; const float* fc1_b;
    const uint16_t* fc2_w; const float* fc2_b;
    const uint16_t* qkv_wf; const float* qkv_bf;   /* ABI 3: LayerNorm-folded qkv / fc1 (all four or none) */
    const uint16_t* fc1_wf; const float* fc1_bf;
} cmdiad_block_weights;
enum { CMDIAD_BLOCK_LN1_READY = 1, CMDIAD_BLOCK_PREP_NEXT = 2 };
size_t cmdiad_transformer_block_workspace_bytes(int M, int C, int hidden);
int cmdiad_transformer_block_fwd(float* x, const float* pos, const cmdiad_block_weights* w, int B, int T, int C, int H,
                                 int hidden, float eps, int flags, uint16_t* q, uint16_t* k, uint16_t* vt, void* workspace,
                                 size_t workspace_bytes, cmdiad_stream_t stream);

/* Point-MAE Encoder (models/models.py:200-215), eval-mode BatchNorm folded into the convolutions:
 *   h1 = relu(W1' x + b1')            3 -> 128   (computed on the fly while staging the GEMM tile)
 *   h2 = W2 h1 + b2                   128 -> 256 ; g = max over the group's points
 * neigh [groups*Mg, 3] f32; w1 [128,4] f32 = {w_x, w_y, w_z, b} per output channel (BN folded).
 * Writes h2 [groups*Mg, 256] bf16, gmax [groups, 256] f32 and (optional) its bf16 copy.
 * Mg (points per group) must be 128, 64 or 32. */
int cmdiad_encoder_stage1(const float* neigh, const float* w1, const uint16_t* W2, const float* b2,
                          int groups, int Mg, uint16_t* h2_out, float* gmax_out, uint16_t* gmax_bf16_out,
                          cmdiad_stream_t stream);
/* out[g][n] = max over the Mg rows of group g of (A . W^T + bias)[row][n]  (torch.max(...,dim=2),
 * models/models.py:214).  A [groups*Mg, K] bf16, W [N,K] bf16, out [groups, N] f32 and/or bf16. */
int cmdiad_gemm_groupmax(const uint16_t* A, const uint16_t* W, const float* bias, int groups, int Mg, int N,
                         int K, float* out_f32, uint16_t* out_bf16, cmdiad_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Patch-library nearest-neighbour scoring
 * ------------------------------------------------------------------------------------------- */

/* Per-query nearest bank row under L2 (torch.cdist + torch.min(dim=1), features.py:190,227) without
 * materialising the QxNb matrix: 16-bit MFMA distance GEMM whose accumulators start at -(|q|^2 + |b|^2) / 2, so that a
 * finished accumulator is -d2 / 2, with a running (min, argmin) epilogue.  keys[q] = min over rows of
 * ((fp32 bits of max(d2', 0)) << 32 | row_offset+row), d2' = d2 with the four low mantissa bits of -d2 / 2 cleared (the
 * epilogue carries the element's place there; 2^-19 of the value, three orders below the operand rounding):
 * integer order == (distance, index) order, so bank shards combine with an integer MIN (RCCL
 * all-reduce over xGMI) and ties resolve to the lowest global row, as torch.min does.  The key of a (query, row) pair does
 * not depend on the launch geometry, the tile shape or the shard the row is in.
 * q [Q,D] bf16, q_sqnorm [Q] f32, bank [Nb,D] bf16, bank_sqnorm [Nb] f32 (squared norms of the
 * bf16-rounded rows, from cmdiad_normalize_cast), keys [Q] u64: the caller initialises keys to
 * UINT64_MAX (or any value above every key, e.g. INT64_MAX); the kernel combines with atomic min.  D % 64 == 0.
 * keys2 [Q] u64 or NULL (ABI 6), initialised like keys: the RUNNER-UP of every query, so that the exact fp32 re-score
 * (cmdiad_l2_rescore2) can repair a near-tie that the 16-bit operands resolved the wrong way -- torch.min on the fp32 cdist
 * (features.py:227) is index work, and with it min_idx equals the fp32 argmin except where a THIRD row lies inside the operand
 * noise too.  Definition (library rows only -- no tile shape, launch geometry or shard in it): the library is cut into groups of
 * 16 rows, group(row) = (row >> 6, (row >> 2) & 3) on GLOBAL rows (row_offset % 64 == 0 is required with keys2); keys2[q] is the
 * smallest key among the group minima other than keys[q]'s own group, i.e. the nearest row outside the winner's group.  Shards
 * combine as: K1 = MIN over shards of keys; K2 = MIN over shards of (keys == K1 ? keys2 : keys).
 * Contract: operands and norms finite (cmdiad_normalize_cast guarantees it: a row with a non-finite element is stored as zeros
 * with squared norm +inf -- as a library row it can never win, as a query it finds nothing and its keys stay untouched). */
enum { CMDIAD_DT_BF16 = 0, CMDIAD_DT_F16 = 1 };   /* 16-bit operand type of the distance GEMM */
int cmdiad_l2_min_keys(const uint16_t* q, const float* q_sqnorm, const uint16_t* bank, const float* bank_sqnorm,
                       int Q, int Nb, int D, uint32_t row_offset, unsigned long long* keys, unsigned long long* keys2, int dtype,
                       cmdiad_stream_t stream);

/* The same search over a query set whose live row count is known only on the device (*q_count <= Q_max, e.g. the compacted set of
 * cmdiad_rows_dedup_plan): the launch is sized for Q_max, rows at and beyond *q_count are neither read nor written. */
int cmdiad_l2_min_keys_counted(const uint16_t* q, const float* q_sqnorm, const int* q_count, int Q_max, const uint16_t* bank,
                               const float* bank_sqnorm, int Nb, int D, uint32_t row_offset, unsigned long long* keys,
                               unsigned long long* keys2, int dtype, cmdiad_stream_t stream);

/* The same search over n_seg query SEGMENTS laid out at a fixed stride -- the row-sharded search of SURVEY 8(e): the gathered,
 * separately compacted query sets of the W ranks of a node against THIS rank's library shard (what features.py:186-190,227 do on
 * one device, per shard).  Segment w = rows [w * seg_stride, w * seg_stride + min(seg_counts[w], seg_stride)) of q / q_sqnorm /
 * keys ([n_seg * seg_stride] each); seg_counts [n_seg] int32 lives on the device, so no host read sizes the launch: ONE launch
 * walks the live query tiles of all segments as one tile list (XCD-aware mapping over the live blocks), instead of one launch per
 * segment.  1 <= n_seg <= 64. */
int cmdiad_l2_min_keys_segments(const uint16_t* q, const float* q_sqnorm, const int* seg_counts, int n_seg, int seg_stride,
                                const uint16_t* bank, const float* bank_sqnorm, int Nb, int D, uint32_t row_offset,
                                unsigned long long* keys, unsigned long long* keys2, int dtype, cmdiad_stream_t stream);

/* Exact removal of repeated query rows in front of the search.  Every patch of the 56 x 56 grid without a foreground pixel under
 * it is the same vector ((0 - mean) / std in every column; features.py:169-184, multiple_features.py:976-977) and the reference's
 * torch.cdist (features.py:186-190) searches the library again for each of them (likewise the hallucinated features of those
 * patches, multiple_features.py:351).  The MOST REPEATED row of the batch is found by a row hash; a row repeats its first
 * occurrence iff hash and squared-norm bits are equal AND all D elements compare equal -- then everything cmdiad_l2_min_keys reads
 * for it is identical and so is its key (a hash collision costs a comparison, never an answer).  slot[Q] (row of the compacted set
 * that answers for q), rows[Q] (compacted -> original, first *count entries), count[1], q_compact [Q,D] / q_sqnorm_compact [Q]
 * (first *count rows written), all on the device; workspace: cmdiad_rows_dedup_workspace_bytes(Q).  Order-preserving; a batch
 * without a repeated row compacts to itself. */
size_t cmdiad_rows_dedup_workspace_bytes(int Q);
int cmdiad_rows_dedup_plan(const uint16_t* q, const float* q_sqnorm, int Q, int D, void* workspace, int* slot, int* rows, int* count,
                           uint16_t* q_compact, float* q_sqnorm_compact, cmdiad_stream_t stream);
/* keys[q] = keys_compact[slot[q]] */
int cmdiad_keys_expand(const unsigned long long* keys_compact, const int* slot, int Q, unsigned long long* keys,
                       cmdiad_stream_t stream);
/* out[q, :] = rows_compact[slot[q], :] (f32, D % 4 == 0): the per-row results of a computation that ran on the compacted rows
 * only (the hallucination MLP of the MTFI path, hallucination_network.py:34-45), back on every original row. */
int cmdiad_rows_expand_f32(const float* rows_compact, const int* slot, int Q, int D, float* out, cmdiad_stream_t stream);

/* Exact fp32 re-score of the winners: min_val[q] = || q_f32[q] - bank_f32[idx - row_offset] ||_2,
 * min_idx[q] = idx (global row).  Queries whose winner lies outside [row_offset, row_offset+Nb) are
 * left untouched (another shard owns them). */
int cmdiad_l2_rescore(const float* q, const float* bank, const unsigned long long* keys, int Q, int Nb,
                      int D, uint32_t row_offset, float* min_val, int64_t* min_idx, cmdiad_stream_t stream);
/* The same over BOTH candidates of every query (keys = best, keys2 = runner-up of cmdiad_l2_min_keys): squared fp32 distances in
 * one summation order; the smaller one wins, of equal ones the lower row -- torch.min's answer on the fp32 distance matrix
 * (features.py:227) wherever the true nearest row is one of the two.
 *   min_val / min_idx (both or neither): the decision, written for queries whose best candidate lies in [row_offset, row_offset+Nb)
 *     (a runner-up outside that range is ignored: use d2_pair + cmdiad_l2_choose when the fp32 rows are sharded);
 *   d2_pair [2][Q] f32 or NULL: the squared distances of the candidates THIS shard owns, others untouched -- zero-filled by the
 *     caller and summed over the shards, every entry has exactly one contributor; cmdiad_l2_choose then takes the decision. */
int cmdiad_l2_rescore2(const float* q, const float* bank, const unsigned long long* keys, const unsigned long long* keys2, int Q,
                       int Nb, int D, uint32_t row_offset, float* d2_pair, float* min_val, int64_t* min_idx, cmdiad_stream_t stream);
int cmdiad_l2_choose(const unsigned long long* keys, const unsigned long long* keys2, const float* d2_pair, int Q, float* min_val,
                     int64_t* min_idx, cmdiad_stream_t stream);

/* Batch-statistics BatchNorm support (the reference runs Point-MAE's BatchNorm1d layers, models/models.py:189,195,
 * in training mode because the extractor is never put in .eval()).  Double-precision moment sums, ACCUMULATED into
 * caller-zeroed buffers:  col_moments: x [rows,C] f32 (row pitch ld) -> sum[C], sumsq[C];
 * moments3: xyz [rows,3] f32 -> out9 = (x, y, z, xx, xy, xz, yy, yz, zz) sums. */
int cmdiad_col_moments(const float* x, size_t rows, int C, int ld, double* sum, double* sumsq, cmdiad_stream_t stream);
int cmdiad_moments3(const float* xyz, size_t rows, double* out9, cmdiad_stream_t stream);

/* Library copy laid out for the fp32 matrix cores ("block16"): groups of 16 rows,
 * out[(((g * D/16 + t) * 4 + kq) * 16 + j) * 4 + e] = bank[16 g + j][16 t + 4 kq + e] (zero beyond Nb),
 * cmdiad_bank_block16_floats(Nb, D) = ceil(Nb/16) * 16 * D floats.  Built once per library; the re-weighting scan
 * streams it with fully coalesced 1 KiB loads straight into MFMA B operands. */
size_t cmdiad_bank_block16_floats(int Nb, int D);
int cmdiad_bank_block16(const float* bank, int Nb, int D, float* out, cmdiad_stream_t stream);

/* Re-weighting scan (features.py:235-254: torch.cdist(m_star, bank) + topk(3, largest=False)): for each of R <= 32
 * probe rows (m_star) the 3 nearest library rows as packed keys (EXACT fp32 squared distance bits << 32 | global row),
 * merged into top3 [R,3] u64, which the caller initialises to UINT64_MAX or 0x7FFF...F (shards call it in turn or
 * all-gather their top3).  One pass over bank_block16 for all probes (v_mfma_f32_16x16x4_f32 cross term, 8 approximate
 * candidates per probe), then the candidates are re-evaluated exactly on the row-major library.  ABI version 2: R <= 32 per
 * call (version 1 took 64) and a workspace of 8 candidates per probe and block (cmdiad_reweight_workspace_bytes).
 * probes [R,D] f32, bank [Nb,D] f32 row-major, bank_block16 from cmdiad_bank_block16; D % 128 == 0. */
size_t cmdiad_reweight_workspace_bytes(int R, int Nb);
int cmdiad_reweight_scan(const float* probes, const float* bank, const float* bank_block16, int R, int Nb, int D,
                         uint32_t row_offset, unsigned long long* top3, void* workspace, size_t workspace_bytes,
                         cmdiad_stream_t stream);
/* The same for the TWO libraries of a scored batch in one launch pair (features.py:235-254 runs per library: xyz and rgb / fusion,
 * multiple_features.py:976-1003): problem 0 and problem 1 each as in cmdiad_reweight_scan (own probes, library, row offset and
 * top3); the scan kernel's workgroups are shared between the libraries in proportion to their rows, so the small library's fixed
 * cost (probes into LDS, candidate merges, the exact re-evaluation launch) runs beside the large library's stream instead of after
 * it.  Results identical to two separate calls.  Both Nb > 0 (an empty shard takes the single-library call for the other). */
size_t cmdiad_reweight_pair_workspace_bytes(int Nb0, int Nb1);
int cmdiad_reweight_scan_pair(const float* probes0, const float* bank0, const float* bank0_block16, int R0, int Nb0,
                              uint32_t row_offset0, unsigned long long* top3_0, const float* probes1, const float* bank1,
                              const float* bank1_block16, int R1, int Nb1, uint32_t row_offset1, unsigned long long* top3_1,
                              int D, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);

/* calculate_dist (features.py:186-190) as a MATERIALISED matrix, exact fp32 (sum of squared differences):
 * out[q][n] = || Q[q] - bank[n] ||_2, [Q, Nb] row-major.  API compatibility of DistHandle.materialize() only: the
 * scoring path never builds the Q x Nb matrix (960 MB per image at the reference's sizes). */
int cmdiad_l2_dist_matrix(const float* q, const float* bank, int Q, int Nb, int D, float* out, cmdiad_stream_t stream);

/* compute_single_s_s_map head / tail (features.py:227-290), batched over B images.
 * head: s_idx[b] = argmax_q min_val[b][q] (first occurrence), s_star[b] = the max; copies
 *       m_test[b] = patch[b][s_idx] and m_star[b] = bank[min_idx[b][s_idx] - row_offset] (only when that
 *       row lies in this shard) into [B,D] buffers (m_star feeds cmdiad_reweight_scan).
 * tail: knn_d[b][k-1] = || m_test[b] - bank[top3[b][k]] || for k = 1, 2 (rows owned by this shard only).
 * final: s[b] = (1 - exp(s_star / sqrt(D)) / sum_k exp(knn_d[b][k] / sqrt(D))) x s_star. */
int cmdiad_score_head(const float* min_val, const int64_t* min_idx, const float* patch, const float* bank,
                      int B, int Q, int D, int Nb, uint32_t row_offset, float* s_star, int32_t* s_idx,
                      float* m_test, float* m_star, cmdiad_stream_t stream);
int cmdiad_score_tail(const float* s_star, const float* m_test, const unsigned long long* top3,
                      const float* bank, int B, int D, int Nb, uint32_t row_offset, float* knn_d,
                      cmdiad_stream_t stream);
int cmdiad_score_final(const float* s_star, const float* knn_d, int B, int D, float* s_out,
                       cmdiad_stream_t stream);

/* Greedy k-centre coreset selection (features.py:372-425, get_coreset_idx_randomp after the random
 * projection; coreset_dtype 'FP16', dist_method_coreset 'l2').  z32 [n,d] f32 projected library (d even),
 * first_idx = 0 as the reference; idx_out [n_select] int64 in selection order. */
size_t cmdiad_coreset_workspace_bytes(int n, int d, int n_select);
int cmdiad_coreset_greedy(const float* z32, int n, int d, int n_select, int first_idx, int64_t* idx_out,
                          void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);
/* coreset_dtype 'TF32' (features.py:390-391; main.py:151): the same selection on the UNROUNDED fp32 rows -- allow_tf32 only touches
 * matrix products and the loop has none, so the reference's branch is an fp32 scan: dist = sqrt(sum (z_i - z_last)^2) in fp32, fp32
 * running minimum, first arg-max.  d <= 1024 (any parity). */
/* Row-sharded selection (fit-time sharding): every rank holds the projected library and scans its own row range; between
 * rounds the caller all-reduces (MAX) the 8-byte packed winner keys of the ranks.  prepare: the first pass (workspace:
 * cmdiad_coreset_workspace_bytes(n, d, 1)); round: rows [row_lo, row_hi) (row_lo % 4 == 0; row_hi % 4 == 0 or == n) against the pivot
 * named by *pivot_key (NULL: row first_idx), winner into *best_out with atomic max (the caller zeroes it first); decode: keys[r] of
 * rounds 0 .. n_select - 2 -> idx_out[0] = first_idx, idx_out[r + 1] = row of keys[r].  Picks identical to cmdiad_coreset_greedy. */
int cmdiad_coreset_prepare(const float* z32, int n, int d, int first_idx, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream);
int cmdiad_coreset_round(void* workspace, int n, int d, int row_lo, int row_hi, const unsigned long long* pivot_key, int first_idx,
                         unsigned long long* best_out, cmdiad_stream_t stream);
int cmdiad_coreset_decode(const unsigned long long* keys, int n_select, int first_idx, int64_t* idx_out, cmdiad_stream_t stream);
size_t cmdiad_coreset_f32_workspace_bytes(int n, int d, int n_select);
int cmdiad_coreset_greedy_f32(const float* z32, int n, int d, int n_select, int first_idx, int64_t* idx_out, void* workspace,
                              size_t workspace_bytes, cmdiad_stream_t stream);

/* Sparse random projection in front of the coreset selection (features.py:360-371: SparseRandomProjection.fit_transform): out [n,
 * n_comp] = X [n,d] . components^T with `components` [n_comp,d] in CSR (int32 indptr / sorted indices, f32 data: the fitted
 * transformer's components_, ABI 3).  Per output element the products are added one by one in the order of the row's non-zeros,
 * each product and each sum rounded to float32 -- the arithmetic of the scipy routine sklearn calls, so out is bit-identical to the
 * host transform.  d <= 4096. */
int cmdiad_sparse_project_f32(const float* X, size_t n, int d, const int* indptr, const int* indices, const float* data, int n_comp,
                              float* out, cmdiad_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Small fused element-wise / layout kernels
 * ------------------------------------------------------------------------------------------- */

/* out_bf16 = bf16 or (out_dtype = CMDIAD_DT_F16) saturating fp16 of ((x - mean) * inv_std); optional out_f32 copy of the normalised values and
 * optional per-row squared norm of the bf16-rounded row (a11 + the |b|^2 term of the distance GEMM). */
int cmdiad_normalize_cast(const float* x, size_t rows, int D, float mean, float inv_std, uint16_t* out_bf16,
                          float* out_f32, float* row_sqnorm, int out_dtype, cmdiad_stream_t stream);
/* The same with the input rows in groups of group_rows, each preceded by group_skip rows that are NOT taken (ABI 6): output row r
 * reads input row r + (r / group_rows + 1) * group_skip.  The rgb patch features are rows 1..784 of every image's 785 ViT tokens
 * (features.py:160-162 drops the cls token, multiple_features.py:976-977 normalises): group_rows 784, group_skip 1 reads the
 * token tensor in place and writes the patch rows compactly.  A non-finite row: see cmdiad_l2_min_keys. */
int cmdiad_normalize_cast_rows(const float* x, size_t rows, int D, int group_rows, int group_skip, float mean, float inv_std,
                               uint16_t* out_bf16, float* out_f32, float* row_sqnorm, int out_dtype, cmdiad_stream_t stream);

/* ViT patch embedding as a GEMM operand: rgb [B,3,S,S] f32 -> patches [B*(S/8)^2, 192] bf16 with
 * k = (c, dy, dx) matching conv weight [768,3,8,8] flattened (timm PatchEmbed, models/models.py:41). */
int cmdiad_im2col_patch8(const float* rgb, int B, int S, uint16_t* patches, cmdiad_stream_t stream);
/* 3 x 3 im2col with padding 1 and stride 1 | 2 (hrnet.py:152, the trunk's stem convolution as a GEMM in the hand-written training
 * step): img [B,C,H,W] f32 -> cols [B*Ho*Wo, ld] bf16, column c * 9 + ky * 3 + kx (the order of the flattened Conv2d weight and of
 * torch.nn.functional.unfold), columns 9 C .. ld - 1 zero.  ld % 8 == 0, ld >= 9 C. */
int cmdiad_im2col3x3_bf16(const float* img, int B, int C, int H, int W, int stride, int ld, uint16_t* cols, cmdiad_stream_t stream);

/* tokens[b][0] = cls + pos[0]; tokens[b][1+i] = patch_out[b][i] + pos[1+i]  (timm _pos_embed,
 * models/models.py:42).  patch_out [B*P, C] f32, tokens [B*(P+1), C] f32. */
int cmdiad_vit_assemble(const float* patch_out, const float* cls, const float* pos, int B, int P, int C,
                        float* tokens, cmdiad_stream_t stream);

/* out[m][:] = act(W x[m] + b) for 3-d inputs: Point-MAE pos_embed first layer nn.Linear(3,128)+GELU
 * (models/models.py:268-272).  x [M,3] f32, wb [N,4] f32 = {w_x,w_y,w_z,bias}, out [M,N] bf16, N % 8 == 0. */
int cmdiad_linear3(const float* x, const float* wb, size_t M, int N, int act, uint16_t* out,
                   cmdiad_stream_t stream);

/* Bilinear up-sampling h x h -> H x H, align_corners=False (features.py:294). in [B,h,h], out [B,H,H]. */
int cmdiad_bilinear_up(const float* in, int B, int h, int H, float* out, cmdiad_stream_t stream);

/* ---- rest of the pointnet2_ops surface (the reference imports the package at models/models.py:5 but
 * only calls furthest_point_sample and gather_operation) ---- */

/* pointnet2_utils.ball_query(radius, nsample, xyz[B,N,3], new_xyz[B,M,3]) -> idx int32 [B,M,nsample]: the first nsample
 * points in index order with squared distance < radius^2, the first hit pre-filling every slot, zeros when nothing is in
 * range.  n_valid as in cmdiad_fps (NULL = N).  Bit-exact with oracle orc_ball_query. */
int cmdiad_ball_query(const float* xyz, const int32_t* n_valid, const float* new_xyz, int B, int N, int M, float radius,
                      int nsample, int32_t* idx_out, cmdiad_stream_t stream);

/* out[b,c,j] = feat[b,c,idx[b,j]]: pointnet2_utils.gather_operation (feat [B,C,N], idx [B,M] -> [B,C,M], models/models.py:77)
 * and grouping_operation (idx [B,M,nsample] flattened to J = M*nsample -> [B,C,M,nsample]).  idx int32, feat / out f32. */
int cmdiad_gather_points(const float* feat, const int32_t* idx, int B, int C, int N, int J, float* out, cmdiad_stream_t stream);

/* ---- on-device tail of the scorer ---- */

/* KNNGaussianBlur (utils/utils.py:71-83) for n_maps maps [n_maps,H,W] f32: each map is divided by its maximum, quantised
 * like torchvision's ToPILImage (mul(255).byte()), blurred exactly as Pillow's ImageFilter.GaussianBlur(radius) does on an
 * 8-bit image (three extended-box passes per axis, 2^24 fixed point; bit-exact with oracle orc_pil_gaussian_blur_u8),
 * and mapped back (/255 * max).  H, W >= 2*floor(box radius)+2 and 2 padded 8-bit copies must fit the LDS (<= 256 x 256). */
int cmdiad_blur8_maps(const float* maps, int n_maps, int H, int W, float radius, float* out, cmdiad_stream_t stream);
size_t cmdiad_blur8_lds_bytes(int H, int W);

/* Per-pixel late fusion (multiple_features.py:985-992: lambda-weighted map pairs -> seg_fuser.score_samples):
 * out[b,p] = ((sum_k double(float(lambdas[k] * maps[b,k,p])) * coef[k]) - offset) + offset, i.e. sklearn's
 * SGDOneClassSVM.score_samples for a model fitted on the host.  maps [B,K,HW] f32 (device), lambdas [K] and coef [K]
 * host arrays, 1 <= K <= 4, out [B,HW] f64 (device). */
int cmdiad_ocsvm_score_maps(const float* maps, int B, int K, int HW, const float* lambdas, const double* coef, double offset,
                            double* out, cmdiad_stream_t stream);

/* One-class SVM FIT on the device: scikit-learn's SGDOneClassSVM.fit for float32 inputs (feature_extractors/features.py:352-358,
 * detect_fuser.fit / seg_fuser.fit; algorithm of the reference's dependency: sklearn _sgd_fast._plain_sgd32 with hinge loss, L2
 * penalty alpha = nu / 2, 'optimal' schedule, shuffle = True, tol / n_iter_no_change stopping) with the same update order:
 * coef, offset and n_iter are those of scikit-learn (tests/test_gpu_ocsvm.py).  X [n,F] f32 on the device, 1 <= F <= 4;
 * seed = the 32-bit seed scikit-learn draws from random_state; pow2 [32][32] (host): column b of M^(2^e) of the xorshift32 step
 * (cmdiad_amd/ocsvm.py computes it); coef_out [F], offset_out, n_iter_out on the host.  Synchronous (one host decision per epoch). */
size_t cmdiad_ocsvm_fit_workspace_bytes(int n, int F);
int cmdiad_ocsvm_fit(const float* X, int n, int F, double nu, int max_iter, double tol, int n_iter_no_change, uint32_t seed,
                     const uint32_t* pow2, float* coef_out, double* offset_out, int* n_iter_out, void* workspace,
                     size_t workspace_bytes, cmdiad_stream_t stream);

/* ---- training side of the FtoF distillation network (models/hallucination_network.py:47-69,
 * hallucination_network_pretrain.py:102-159) ---- */

/* Loss head: y = GELU(z3); mode 0 'l2' sum_rows ||y-t||_2, 1 'cos_dist' sum_rows (1 - cos(y,t)), 2 'smooth_l1'
 * sum of elements (beta 1); row_loss [M] gets the per-row terms (reduce with cmdiad_sum_vector, scale 1/B);
 * dz3 [M,D] bf16 (NULL allowed) = inv_b * dLoss/dz3 (through the output GELU); y_out [M,D] f32 optional.
 * ABI 3: mode + CMDIAD_LOSS_OUT_NONE = no output activation (y = z3), mode + CMDIAD_LOSS_OUT_SIGMOID = sigmoid of the output AND
 * of the target -- the convolutional head's two losses (hallucination_network.py:136-147). */
enum { CMDIAD_LOSS_OUT_NONE = 256, CMDIAD_LOSS_OUT_SIGMOID = 512 };
int cmdiad_loss_head(const float* z3, const float* target, int M, int D, int mode, float inv_b,
                     float* row_loss, uint16_t* dz3, float* y_out, cmdiad_stream_t stream);

/* ---- training side of the convolutional FtoF head (models/hallucination_network.py:72-147: conv3x3 -> BatchNorm2d -> ReLU x3,
 * conv3x3 per direction; hallucination_network_pretrain.py:106-147 trains it in train() mode = BatchNorm on batch statistics).
 * Activations are NHWC flattened to [M = B*H*W, C].  Convolution forward and data gradient: cmdiad_conv2d_nhwc_bf16 (the latter on
 * flipped, transposed weights); weight gradient: cmdiad_gemm_tn_bf16 per filter tap over cmdiad_pad_nhwc_bf16 copies. ---- */

/* The constants of one batch-statistics BatchNorm from the float64 column sums of cmdiad_col_moments (sum, sumsq [C]) over `rows`
 * rows: mean64 / var64 [C] (biased variance; the caller's running-statistics update), scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale, mean, rstd as f32 [C]. */
int cmdiad_bn_affine(const double* sum, const double* sumsq, const float* gamma, const float* beta, size_t rows, double eps, int C,
                     double* mean64, double* var64, float* scale, float* shift, float* mean, float* rstd, cmdiad_stream_t stream);
/* y = z * scale[c] + shift[c] (+ residual [M,C] f32) (ReLU when relu != 0) as bf16 and / or f32 (NULL = not wanted): BatchNorm
 * (scale = gamma / sqrt(var + eps), shift = beta - mean * scale, batch statistics from cmdiad_col_moments) + ReLU,
 * nn.Sequential(..., BatchNorm2d, ReLU, ...) of hallucination_network.py:80-90; with the residual: bn3 + identity + ReLU of
 * Bottleneck.forward (hrnet.py:38-42).  C % 8 == 0. */
int cmdiad_bn_relu_fwd(const float* z, const float* scale, const float* shift, const float* residual, int relu, size_t M, int C,
                       uint16_t* y, float* y_f32, cmdiad_stream_t stream);
/* Backward of the same (autograd of hallucination_network_pretrain.py:146): with g = dy where z * scale + shift > 0 else 0
 * (masked != 0: the layer's own ReLU) or g = dy (masked == 0: no ReLU of its own -- the caller has applied the mask of the
 * ReLU after the residual sum) and xhat = (z - mean) * rstd:  part_dbeta / part_dgamma [chunks][C] = column sums of g /
 * g * xhat over `chunks` row ranges; cmdiad_bn_partials_sum adds them (fixed order) into dbeta / dgamma [C]; then
 * dz = scale * (g - dbeta / M - xhat * dgamma / M) as bf16 (per-channel vectors 16-byte aligned, C % 4 == 0). */
int cmdiad_bn_relu_bwd_reduce(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                              const float* rstd, int masked, size_t M, int C, int chunks, float* part_dbeta, float* part_dgamma,
                              cmdiad_stream_t stream);
int cmdiad_bn_partials_sum(const float* part_dbeta, const float* part_dgamma, int chunks, int C, float* dbeta, float* dgamma,
                           cmdiad_stream_t stream);
int cmdiad_bn_relu_bwd_apply(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                             const float* rstd, const float* dbeta, const float* dgamma, int masked, size_t M, int C, uint16_t* dz,
                             cmdiad_stream_t stream);
/* x [B,H,W,C] bf16 -> the interior of out [B,H+2,W+2,C] (border untouched: the caller zero-fills out once).  In that layout the
 * (ky, kx) tap of a 3x3 convolution with padding 1 is the row offset (ky-1)*(W+2) + (kx-1): the weight gradient of a tap is
 * one [M',N]^T [M',C] product of the padded output gradient with the shifted padded input.  C % 8 == 0. */
int cmdiad_pad_nhwc_bf16(const uint16_t* x, int B, int H, int W, int C, uint16_t* out, cmdiad_stream_t stream);
/* Backward pieces of the feature-to-input convolutional head (hallucination_network.py:185-220; trained by
 * hallucination_network_pretrain.py:106-147).  relu_bwd: dz = dx where the saved ReLU OUTPUT y (bf16) is positive, else 0, as bf16
 * and / or f32 (n elements, n % 4 == 0).  upsample_bicubic_bwd: adjoint of F.interpolate(mode='bicubic', align_corners=False) (:204):
 * grad_out f32 NHWC [B,H,W,C] -> grad_in f32 [B,h,w,C]; tmp f32 [B,H,w,C]; gathering passes, no atomics.  C % 4 == 0. */
int cmdiad_relu_bwd_bf16(const float* dx, const uint16_t* y, size_t n, uint16_t* dz, float* dz_f32, cmdiad_stream_t stream);
int cmdiad_upsample_bicubic_bwd(const float* grad_out, int B, int H, int W, int C, int h, int w, float* tmp, float* grad_in,
                                cmdiad_stream_t stream);


/* out[i] = scale * sum_s slabs[s*stride + i] in fixed order (split-K partials, column partials). n, stride % 4 == 0 */
int cmdiad_reduce_slabs(const float* slabs, int S, size_t n, size_t stride, float scale, float* out,
                        cmdiad_stream_t stream);
int cmdiad_sum_vector(const float* x, size_t n, float scale, float* out, cmdiad_stream_t stream);
/* partial[chunk][n] = sum over the chunk's rows of x[r][n]  (bias gradients). */
int cmdiad_colsum_bf16(const uint16_t* x, int M, int N, int chunks, float* partial, cmdiad_stream_t stream);
/* LayerNorm weight / bias gradient partials from dh [M,C], the LN input x and its saved row statistics. */
int cmdiad_ln_param_grad(const float* dh, const float* x, const float* mean, const float* rstd, int M, int C,
                         int chunks, float* partial_g, float* partial_b, cmdiad_stream_t stream);
/* torch.optim.Adam update (no weight decay, amsgrad off; hallucination_network_pretrain.py:261) on g*grad_scale;
 * optionally refreshes the bf16 GEMM operand copy of the parameter. */
int cmdiad_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                     float eps, int step, float grad_scale, uint16_t* p_bf16, cmdiad_stream_t stream);

/* Generic fp32 -> bf16 cast (n % 4 == 0) and bf16 2-D transpose out[c][r] = in[r][c]. */
int cmdiad_cast_bf16(const float* x, size_t n, uint16_t* out, cmdiad_stream_t stream);
int cmdiad_transpose_bf16(const uint16_t* in, int rows, int cols, uint16_t* out, cmdiad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CMDIAD_HIP_H */
