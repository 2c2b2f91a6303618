// Weight-gradient product for the distillation trainer (hallucination_network_pretrain.py:102-159 -> loss.backward()):
//   C[N1, N2] = sum_m P[m, n1] . Q[m, n2]          (dW = dZ^T . A: P = dZ [M, dout], Q = A [M, din], both ROW-major)
// i.e. a GEMM whose reduction index is the ROW of both operands ("TN").  The NT kernels of gemm_core.h need it K-contiguous,
// which cost the trainer one transposition pass per operand per step (2.7 of 17.4 ms).  Here the [64 m][128 n] tiles go to
// LDS as they lie in memory (LDS-DMA, 256-byte rows) and the MFMA fragments -- 8 consecutive m for one n per lane -- are
// gathered by gfx950's transposing LDS read ds_read_b64_tr_b16 (a 4 x 16 block per 16-lane group, delivered column-major).
//
// LDS image of a tile: 256-byte rows, 16-byte chunk ch of row r at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))): with
// that XOR the transposed reads of the 16x16x32 operand (two 16-lane groups 8 rows apart per 32-lane half) are
// conflict-free.  The swizzle goes on the DMA SOURCE address (lane l lands on row l >> 4, physical chunk l & 15).
// Block = 128 x 128 outputs, 4 waves of 64 x 64, K-step = 64 rows of m, 2 stages (64 KiB: two blocks per CU), split-K over
// grid.y into fp32 slabs (cmdiad_reduce_slabs sums them).  The Q fragment is fed as the MFMA "A" operand, so a lane holds 4
// consecutive n2 of one n1: 16-byte row-major stores.
#include <mutex>

#include "gemm_core.h"

namespace {

using namespace gemm;

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int TM = 64, TN = 128;
constexpr int TILE_BYTES = TM * TN * 2;      // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // P tile + Q tile
constexpr int LDS_BYTES = 2 * STAGE_BYTES;

struct TnParams {
    const bf16_t* P; int ldp;
    const bf16_t* Q; int ldq;
    int M, N1, N2;
    float* out; int ldo;
    int split;
    float* colsum;  // optional [split][N1]: sum over m of P[m, n1] (the bias gradient), produced by the n2-tile-0 blocks
};

__device__ __forceinline__ int row_xor(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// one [64][128] tile: 16 pieces of 4 rows x 256 B, 4 per wave
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* g, int ld, int ncols, int m0, int c0, int wave, int lane)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;
        const int r = piece * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ row_xor(r);
        const int col = min(c0 + ch * 8, ncols - 8);  // columns past the matrix: any valid data, masked at the store
        const bf16_t* src = g + (size_t)(m0 + r) * ld + col;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tile + piece * 1024), 16, 0, 0);
    }
}

// 8 consecutive m (rows kk*32 + g*8 .. +7) of column 16 t + (lane & 15): two transposed 4 x 16 block reads
__device__ __forceinline__ bf16x8 tr_fragment(const char* tile, int t, int kk, int lane)
{
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    s16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = kk * 32 + g * 8 + hh * 4 + q;
        const int ch = t * 2 + (p >> 1);
        const int off = 256 * row + 16 * (ch ^ row_xor(row)) + 8 * (p & 1);
        h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off));
    }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tiles2 = (p.N2 + TN - 1) / TN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int n1_0 = (wg / tiles2) * TN, n2_0 = (wg % tiles2) * TN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int KT = p.M / TM;
    const int per = (KT + p.split - 1) / p.split;
    const int kt0 = blockIdx.y * per;
    const int count = min(per, KT - kt0);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient: P^T . 1 as four more MFMAs per 32 rows, in the blocks of the first n2 tile only (their wc == 0 waves)
    const bool want_colsum = p.colsum != nullptr && n2_0 == 0 && wc == 0;
    f32x4 csum[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) csum[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    auto stage = [&](int slot, int kt) {
        char* buf = lds + slot * STAGE_BYTES;
        stage_tile(buf, p.P, p.ldp, p.N1, kt * TM, n1_0, wave, lane);
        stage_tile(buf + TILE_BYTES, p.Q, p.ldq, p.N2, kt * TM, n2_0, wave, lane);
    };
    if (count > 0) {
        stage(0, kt0);
        wait_vmcnt<0>();
        block_barrier();
    }
    for (int it = 0; it < count; ++it) {
        const int slot = it & 1;
        const char* tp = lds + slot * STAGE_BYTES;
        const char* tq = tp + TILE_BYTES;
        // every fragment of the K-step is read BEFORE the next stage's DMA is issued: the compiler guards a transposing
        // read that follows an LDS-DMA in program order with s_waitcnt vmcnt(0) (it cannot tell the two stages apart), which
        // would serialise the DMA with the MFMAs; in this order the wait only covers loads that have already landed
        bf16x8 pf[2][4], qf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int a = 0; a < 4; ++a) pf[kk][a] = tr_fragment(tp, wr * 4 + a, kk, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) qf[kk][b] = tr_fragment(tq, wc * 4 + b, kk, lane);
        }
        if (it + 1 < count) stage(slot ^ 1, kt0 + it + 1);  // that stage was last read in step it-1
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(qf[kk][b], pf[kk][a], acc[a][b]);
        if (want_colsum) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < 4; ++a) csum[a] = mfma16(ones, pf[kk][a], csum[a]);
        }
        __builtin_amdgcn_s_setprio(0);
        wait_vmcnt<0>();
        block_barrier();
    }
    if (want_colsum && lane < 16) {  // every row of the ones-product is the column sum: take row 0 (lanes 0..15, element 0)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n1 = n1_0 + wr * 64 + a * 16 + lane;
            if (n1 < p.N1) p.colsum[(size_t)blockIdx.y * p.N1 + n1] = csum[a][0];
        }
    }

    float* out = p.out + (size_t)blockIdx.y * p.N1 * p.ldo;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n1 = n1_0 + wr * 64 + a * 16 + (lane & 15);
        if (n1 >= p.N1) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n2 = n2_0 + wc * 64 + b * 16 + (lane >> 4) * 4;
            if (n2 >= p.N2) continue;
            *reinterpret_cast<f32x4*>(out + (size_t)n1 * p.ldo + n2) = acc[a][b];
        }
    }
}

}  // namespace

extern "C" int cmdiad_gemm_tn_bf16(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, int M, int N1, int N2,
                                   int split_k, float* out_f32, int ldo, float* colsum_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(P && Q && out_f32, CMDIAD_ERR_ARG, "cmdiad_gemm_tn_bf16: null pointer");
    CMDIAD_REQUIRE(M > 0 && M % 64 == 0 && N1 >= 8 && N2 >= 8 && N1 % 8 == 0 && N2 % 8 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_gemm_tn_bf16: need M%%64==0, N1%%8==0, N2%%8==0 (M=%d N1=%d N2=%d)", M, N1, N2);
    CMDIAD_REQUIRE(ldp % 8 == 0 && ldq % 8 == 0 && ldp >= N1 && ldq >= N2 && ((uintptr_t)P & 15) == 0 && ((uintptr_t)Q & 15) == 0 &&
                       ldo % 4 == 0 && ldo >= N2 && ((uintptr_t)out_f32 & 15) == 0,
                   CMDIAD_ERR_ARG, "cmdiad_gemm_tn_bf16: operands must be 16-byte aligned with ld%%8==0 (out: ld%%4==0)");
    const int split = split_k > 1 ? split_k : 1;
    CMDIAD_REQUIRE(split <= M / 64, CMDIAD_ERR_ARG, "cmdiad_gemm_tn_bf16: split_k=%d exceeds the %d K-steps", split, M / 64);
    static std::once_flag once;
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] { attr = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); });
    CMDIAD_REQUIRE(attr == hipSuccess, CMDIAD_ERR_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", LDS_BYTES);
    TnParams p{(const bf16_t*)P, ldp, (const bf16_t*)Q, ldq, M, N1, N2, out_f32, ldo, split, colsum_out};
    const unsigned blocks = (unsigned)(((N1 + TN - 1) / TN) * ((N2 + TN - 1) / TN));
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(blocks, split), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
