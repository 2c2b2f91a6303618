// Point-MAE encoder, second half, in ONE kernel (models/models.py:204-215 after the first max-pool):
//   h3  = ReLU(W3b . h2 + gb[group])          [rows, 512]   (gb = W3a . groupmax(h2) + b3, the broadcast half of conv3)
//   tok = max over the group's rows of (W4 . h3 + b4)        [groups, 384]
// The separate kernels write h3 (4.3 GB per batch of 32) and read it back; here a block of 8 waves owns 128 rows, keeps its
// h2 tile (64 KiB) in LDS, produces h3 in four 128-column chunks that only ever exist in LDS (32 KiB), and accumulates the
// 128 x 384 output across the chunks in registers (96 per lane).  LDS = 64 + 32 + 4 x 16 (weight stages) = 160 KiB, one
// block (two waves per SIMD) per CU.  Every phase is a 128 x 128 x 64 product against one streamed weight tile
// (40 tiles per block: 4 chunks x (4 K-steps of W3b + 3 output chunks x 2 K-steps of W4)), one barrier per phase.
// A phase is only ~0.2 us of MFMAs and the weights come from L2 (~0.6 us away), so the weight stream runs THREE tiles
// ahead with a counted s_waitcnt vmcnt (never drained inside the block): with one tile ahead and a drain per phase every
// phase waited for an L2 round trip (5.8 ms per batch; see profiles/r1_notes.md for the staged comparison).
// Arithmetic order equals the two-kernel path (same bf16 rounding of h3, same K order of the fp32 accumulation), so
// the tokens are bit-identical to gemm_bf16 + gemm_groupmax.  The group maximum is combined across the four 32-row wave
// rows of the block through LDS; a block always owns whole groups, so every output is stored exactly once.
#include "gemm_core.h"

namespace {

using namespace gemm;

constexpr int TM = 128;                         // rows per block
constexpr int TW = 8;                           // waves per block: 4 x 2 grid of 32 x 64 accumulator tiles
constexpr int KB_BYTES = TM * BK * 2;           // one [64][64] bf16 k-block: 8 KiB
constexpr int A2_BYTES = 4 * KB_BYTES;          // h2 tile, K = 256
constexpr int A3_BYTES = 2 * KB_BYTES;          // h3 chunk, 128 columns = K of the next product
constexpr int W_STAGE = 128 * BK * 2;           // one [128][64] weight tile: 16 KiB
#ifndef CMDIAD_TAIL_STAGES
#define CMDIAD_TAIL_STAGES 4
#endif
constexpr int NST = CMDIAD_TAIL_STAGES;         // weight stages in LDS: NST - 1 tiles stay in flight across the barriers
constexpr int AHEAD = NST - 1;
constexpr int TAIL_LDS = A2_BYTES + A3_BYTES + NST * W_STAGE;  // 160 KiB at 4 stages
static_assert(NST >= 2 && NST <= 4 && TAIL_LDS <= 160 * 1024, "weight stages");

struct TailParams {
    int M, Mg;
    const float* gb;   // [groups, 512]
    const float* b4;   // [384]
    float* tok;        // [groups, 384]
};

// acc[2][4] += A(32 rows of this wave, k-block `ta`) . W(64 columns of this wave, tile `tw`)^T, swapped orientation
__device__ __forceinline__ void phase(f32x4 (&acc)[2][4], const char* ta, const char* tw, int wr, int wc, int lane)
{
    bf16x8 af[2][2], wf[2][4];  // both 32-deep halves requested up front: the second half's LDS latency hides under the first's MFMAs
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) af[kk][i] = *reinterpret_cast<const bf16x8*>(ta + lds_off(wr * 32 + i * 16 + (lane & 15), chunk));
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[kk][j] = *reinterpret_cast<const bf16x8*>(tw + lds_off(wc * 64 + j * 16 + (lane & 15), chunk));
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[kk][j], af[kk][i], acc[i][j]);
}

__global__ __launch_bounds__(TW * 64, 2) void encoder_tail_kernel(GlobalTile H2, GlobalTile W3, GlobalTile W4, TailParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* A2 = lds;
    char* A3 = lds + A2_BYTES;
    char* WS = A3 + A3_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * TM;

#pragma unroll
    for (int kb = 0; kb < 4; ++kb) H2.stage<TM, TW>(A2 + kb * KB_BYTES, m0, kb * BK, tid);
    // weight tile t of the block's sequence: chunk c = t / 10; u = t % 10: u < 4 -> W3b rows [128c, +128), K-step u;
    // else output chunk o = (u - 4) / 2, K-step k2 = (u - 4) % 2 of W4 rows [128 o, +128), columns 128 c + 64 k2
    auto stage_w = [&](int t, int slot) {
        const int c = t / 10, u = t - c * 10;
        char* buf = WS + slot * W_STAGE;
        if (u < 4) W3.stage<128, TW>(buf, c * 128, u * BK, tid);
        else W4.stage<128, TW>(buf, ((u - 4) >> 1) * 128, c * 128 + ((u - 4) & 1) * BK, tid);
    };
    // Group-bias values of a chunk (16 per lane) are fetched long before they are used, by inline asm: a load the compiler
    // can see gets a compiler-placed s_waitcnt vmcnt(0) at its use -- a drain of the weight stream per chunk.  In-order
    // retirement makes the counted waits of the phases in between (>= 4) cover them.
    const int row0 = m0 + wr * 32;                 // this wave's 32 rows share a group (32 | Mg)
    const float* gb = p.gb + (size_t)(min(row0, p.M - 1) / p.Mg) * 512 + wc * 64 + (lane >> 4) * 4;
    f32x4 gbv[4];
    auto fetch_gb = [&](int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gbv[j]) : "v"(gb + c * 128 + j * 16) : "memory");
    };
    fetch_gb(0);
#pragma unroll
    for (int t0 = 0; t0 < AHEAD; ++t0) stage_w(t0, t0);
    // tile 0 (and the h2 tile, issued before it: the counter retires in order) landed; AHEAD - 1 tiles stay in flight
    if constexpr (AHEAD == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    block_barrier();

    f32x4 acc3[2][4], acco[3][2][4];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acco[o][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one phase = (issue the weight tile AHEAD steps on, into the stage the previous phase just left) + 16 MFMAs per wave
    // + counted wait + barrier; t counts the 40 tiles of the block
    int t = 0;
    auto step = [&](f32x4 (&acc)[2][4], const char* ta) {
        if (t + AHEAD < 40) stage_w(t + AHEAD, (t + AHEAD) % NST);
        phase(acc, ta, WS + (t % NST) * W_STAGE, wr, wc, lane);
        ++t;
    };
    int gb_young = 0;  // syncs for which the 4 group-bias loads are still YOUNGER than the tile being waited for
    auto sync = [&]() {  // tile t must have landed; the loads issued after it (2 pieces per wave per tile) may stay in flight
        const int n = min(t - 1 + AHEAD, 39) - t;
        if (gb_young > 0) {  // only in the body of the stream, where n == AHEAD - 1
            --gb_young;
            if constexpr (AHEAD == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (AHEAD == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (n >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_barrier();
    };
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            step(acc3, A2 + u * KB_BYTES);
            if (u < 3) sync();
        }
        // h3 chunk c: + group bias, ReLU, bf16, into LDS in the A-operand layout of the next product
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = wc * 64 + j * 16 + (lane >> 4) * 4;  // column inside the chunk
            const f32x4 b = gbv[j];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = wr * 32 + i * 16 + (lane & 15);
                const f32x4 v = acc3[i][j];
                bf16x4 o = {f2bf(fmaxf(v[0] + b[0], 0.f)), f2bf(fmaxf(v[1] + b[1], 0.f)), f2bf(fmaxf(v[2] + b[2], 0.f)),
                            f2bf(fmaxf(v[3] + b[3], 0.f))};
                *reinterpret_cast<bf16x4*>(A3 + (n >> 6) * KB_BYTES + lds_off(m, (n & 63) >> 3) + (n & 7) * 2) = o;
            }
        }
        if (c < 3) { fetch_gb(c + 1); gb_young = AHEAD; }  // used six phases on; younger than the awaited tile for AHEAD syncs
        sync();
        step(acco[0], A3); sync(); step(acco[0], A3 + KB_BYTES); sync();
        step(acco[1], A3); sync(); step(acco[1], A3 + KB_BYTES); sync();
        step(acco[2], A3); sync(); step(acco[2], A3 + KB_BYTES); sync();
    }

    // Group maximum.  A block's 128 rows are whole groups (Mg divides 128 and blocks start at multiples of 128), so nothing
    // is shared between blocks: the four 32-row partial maxima meet in LDS (the h2 tile is dead by now) and each output is
    // stored once -- no atomics (they were 1 536 per block, 50 M per batch, 1.2 GB of L2 atomic traffic) and no pre-fill.
    float* s_part = reinterpret_cast<float*>(A2);  // [4 row blocks][384]
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = o * 128 + wc * 64 + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaxf(acco[o][0][j][r], acco[o][1][j][r]);
                v = row16_max(v);
                if ((lane & 15) == 0) s_part[wr * 384 + n + r] = v;
            }
        }
    __syncthreads();
    const int per = p.Mg / 32;                     // 32-row blocks per group: 1, 2 or 4
    for (int e = tid; e < (4 / per) * 384; e += TW * 64) {
        const int g = e / 384, n = e - g * 384;
        const int row = m0 + g * p.Mg;
        if (row >= p.M) continue;
        float v = s_part[(g * per) * 384 + n];
        for (int q = 1; q < per; ++q) v = fmaxf(v, s_part[(g * per + q) * 384 + n]);
        p.tok[(size_t)(row / p.Mg) * 384 + n] = v + p.b4[n];
    }
}

}  // namespace

extern "C" int cmdiad_encoder_tail(const uint16_t* h2, const float* gb, const uint16_t* W3b, const uint16_t* W4, const float* b4,
                                   int groups, int Mg, float* tok_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(h2 && gb && W3b && W4 && b4 && tok_out, CMDIAD_ERR_ARG, "cmdiad_encoder_tail: null pointer");
    CMDIAD_REQUIRE(groups > 0 && (Mg == 32 || Mg == 64 || Mg == 128), CMDIAD_ERR_ARG, "cmdiad_encoder_tail: Mg in {32,64,128} (Mg=%d)", Mg);
    CMDIAD_REQUIRE(((((uintptr_t)h2 | (uintptr_t)W3b | (uintptr_t)W4 | (uintptr_t)gb | (uintptr_t)b4) & 15) == 0), CMDIAD_ERR_ARG,
                   "cmdiad_encoder_tail: 16-byte alignment");
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)encoder_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) != hipSuccess) {
            cmdiad_set_error("cmdiad_encoder_tail: hipFuncSetAttribute failed");
            return CMDIAD_ERR_LAUNCH;
        }
        attr = true;
    }
    const int M = groups * Mg;
    hipStream_t s = (hipStream_t)stream;
    GlobalTile H2{(const bf16_t*)h2, 256, M}, W3{(const bf16_t*)W3b, 256, 512}, W4t{(const bf16_t*)W4, 512, 384};
    TailParams p{M, Mg, gb, b4, tok_out};
    hipLaunchKernelGGL(encoder_tail_kernel, dim3((M + TM - 1) / TM), dim3(TW * 64), TAIL_LDS, s, H2, W3, W4t, p);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
