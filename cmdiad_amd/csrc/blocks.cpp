// Coarse entry points of the C ABI: whole pre-LN transformer blocks sequenced on the host side of the library, so that a
// caller pays one FFI call per block instead of seven (SURVEY 8b lists `vit_block_fwd` among the proposed exports; at
// B = 1 the Python / ctypes launch path, not the GPU, bounds the drop-in).  Pure launch sequencing over the entry points
// of gemm.hip / attention.hip / misc.hip: no kernels here, nothing allocated, stream-ordered like everything else.
#include "common.h"

extern "C" size_t cmdiad_transformer_block_workspace_bytes(int M, int C, int hidden)
{
    if (M <= 0 || C <= 0 || hidden <= 0) return 0;
    const size_t rows256 = ((size_t)M + 255) / 256 * 256;
    return (size_t)M * (size_t)(2 * C + hidden) * sizeof(uint16_t)   // LN output / raw bf16 rows, attention output, MLP hidden (bf16)
         + (size_t)(C / 64) * M * 2 * sizeof(float)                  // LayerNorm fold: chunk partials [C/64][M][2]
         + rows256 * sizeof(float);                                  // ... and 1 / sigma per row (readable up to M rounded up to 256)
}

// models/models.py:177-180 (Block.forward; timm's ViT block has the same algebra, models.py:48):
//   x += proj(attn(LN1(x (+pos))));  x += fc2(GELU(fc1(LN2(x))))      on the fp32 residual stream x [B*T, C], in place.
// With folded weights the LayerNorms disappear into the products around them (gemm.hip "LayerNorm fold"): proj emits the raw
// bf16 rows + chunk statistics of its output, a one-thread-per-row kernel turns them into 1 / sigma, fc1 applies it per row;
// with PREP_NEXT fc2 does the same for the next block's first LayerNorm (adding that block's pos first), which the next call
// is told about with LN1_READY.
extern "C" int cmdiad_transformer_block_fwd(float* x, const float* pos, const cmdiad_block_weights* w, int B, int T, int C, int H,
                                            int hidden, float eps, int flags, uint16_t* q, uint16_t* k, uint16_t* vt, void* workspace,
                                            size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && w && q && k && vt && workspace, CMDIAD_ERR_ARG, "cmdiad_transformer_block_fwd: null pointer");
    CMDIAD_REQUIRE(B > 0 && T > 0 && C > 0 && H > 0 && C == H * 64 && hidden > 0, CMDIAD_ERR_ARG,
                   "cmdiad_transformer_block_fwd: need C == 64*H (B=%d T=%d C=%d H=%d hidden=%d)", B, T, C, H, hidden);
    const int M = B * T;
    CMDIAD_REQUIRE(workspace_bytes >= cmdiad_transformer_block_workspace_bytes(M, C, hidden), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_transformer_block_fwd: workspace %zu < %zu bytes", workspace_bytes,
                   cmdiad_transformer_block_workspace_bytes(M, C, hidden));
    const bool folded = w->qkv_wf != nullptr;
    CMDIAD_REQUIRE(folded ? (w->qkv_bf && w->fc1_wf && w->fc1_bf) : (!w->qkv_bf && !w->fc1_wf && !w->fc1_bf), CMDIAD_ERR_ARG,
                   "cmdiad_transformer_block_fwd: the folded weights come as all four or none");
    CMDIAD_REQUIRE((flags & ~(CMDIAD_BLOCK_LN1_READY | CMDIAD_BLOCK_PREP_NEXT)) == 0 && (folded || flags == 0), CMDIAD_ERR_ARG,
                   "cmdiad_transformer_block_fwd: flags %d need the folded weights", flags);
    uint16_t* h = (uint16_t*)workspace;            // [M, C]: LayerNorm output, or the raw rows as bf16 (folded)
    uint16_t* a = h + (size_t)M * C;               // [M, C]
    uint16_t* m = a + (size_t)M * C;               // [M, hidden]
    float* part = (float*)(m + (size_t)M * hidden);          // [C/64][M][2]
    float* rstd = part + (size_t)(C / 64) * M * 2;           // [M rounded up to 256]
    int rc;
    if (flags & CMDIAD_BLOCK_LN1_READY) {
        if ((rc = cmdiad_gemm_qkv(h, w->qkv_wf, w->qkv_bf, rstd, B, T, C, q, k, vt, stream))) return rc;
    } else {
        if ((rc = cmdiad_layernorm(x, pos, w->ln1_w, w->ln1_b, eps, M, C, h, nullptr, 0, nullptr, nullptr, stream))) return rc;
        if ((rc = cmdiad_gemm_qkv(h, w->qkv_w, w->qkv_b, nullptr, B, T, C, q, k, vt, stream))) return rc;
    }
    if ((rc = cmdiad_attention(q, k, vt, B, H, T, a, stream))) return rc;
    cmdiad_gemm_args g{};
    g.A = a; g.lda = C; g.W = w->proj_w; g.ldw = C; g.M = M; g.N = C; g.K = C; g.bias = w->proj_b;
    g.residual = x; g.ldr = C; g.out_f32 = x; g.ldo32 = C; g.act = CMDIAD_ACT_NONE; g.group_rows = 1; g.split_k = 1;
    if (folded) { g.ln_xb = h; g.ld_xb = C; g.ln_part = part; }
    if ((rc = cmdiad_gemm_bf16(&g, stream))) return rc;
    cmdiad_gemm_args f1{};
    f1.A = h; f1.lda = C; f1.ldw = C; f1.M = M; f1.N = hidden; f1.K = C;
    f1.act = CMDIAD_ACT_GELU; f1.out_bf16 = m; f1.ldo16 = hidden; f1.group_rows = 1; f1.split_k = 1;
    if (folded) {
        if ((rc = cmdiad_ln_stats_finalize(part, M, C / 64, eps, rstd, nullptr, stream))) return rc;
        f1.W = w->fc1_wf; f1.bias = w->fc1_bf; f1.row_scale = rstd;
    } else {
        if ((rc = cmdiad_layernorm(x, nullptr, w->ln2_w, w->ln2_b, eps, M, C, h, nullptr, 0, nullptr, nullptr, stream))) return rc;
        f1.W = w->fc1_w; f1.bias = w->fc1_b;
    }
    if ((rc = cmdiad_gemm_bf16(&f1, stream))) return rc;
    cmdiad_gemm_args f2{};
    f2.A = m; f2.lda = hidden; f2.W = w->fc2_w; f2.ldw = hidden; f2.M = M; f2.N = C; f2.K = hidden; f2.bias = w->fc2_b;
    f2.residual = x; f2.ldr = C; f2.out_f32 = x; f2.ldo32 = C; f2.act = CMDIAD_ACT_NONE; f2.group_rows = 1; f2.split_k = 1;
    if (flags & CMDIAD_BLOCK_PREP_NEXT) { f2.ln_xb = h; f2.ld_xb = C; f2.ln_part = part; f2.add2 = pos; f2.ld_add2 = C; }
    if ((rc = cmdiad_gemm_bf16(&f2, stream))) return rc;
    if (flags & CMDIAD_BLOCK_PREP_NEXT) return cmdiad_ln_stats_finalize(part, M, C / 64, eps, rstd, nullptr, stream);
    return CMDIAD_OK;
}
