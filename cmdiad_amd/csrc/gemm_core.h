// bf16 MFMA GEMM core for gfx950: C[M,N] = A[M,K] . W[N,K]^T ("NT": both operands K-contiguous,
// the layout of nn.Linear weights, of token matrices and of the query / bank matrices).
//
// Tile 128 x 128 x 64, 256 threads = 4 waves in a 2 x 2 grid, each wave owns a 64 x 64 output as
// 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (16 f32x4 accumulators = 64 VGPRs).
// LDS: two 32 KiB stages, each = A tile + W tile, [128 rows][64 bf16] with 128-byte rows whose 16-byte
// chunks are XOR-swizzled by (row & 7): the ds_read_b128 fragment reads (16 rows x 4 k-chunks per
// 16-lane service group) then touch 16 distinct 16-byte slots of the 256-byte bank row -> conflict-free.
// Staging is register-based and split (guide T14): the global loads of step t+1 are issued before
// the MFMAs of step t and written to the other LDS stage after them; ONE barrier per K-step.
// The (n-tile, k-tile) iteration space is flattened so a block that owns several N tiles (the
// distance GEMM's running-min loop) keeps the pipeline full across tile boundaries.
//
// Orientation: with SWAP = true the weight fragment is fed as the MFMA "A" operand, so the
// accumulator tile is C^T: every lane then holds 4 CONSECUTIVE n for one m, which makes the
// row-major epilogue stores 16-byte (f32) / 8-byte (bf16) vectors and lets a per-query running
// min live in one lane.  SWAP = false gives 4 consecutive m per lane (used for transposed stores).
#pragma once
#include "common.h"

namespace gemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB
constexpr int kLdsBytes = 2 * kStageBytes;       // 64 KiB -> 2 blocks / CU

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ (row & 7)) << 4); }

// Operand tile loaders: fill 4 x 16 B per thread (row = (tid>>3) + 32*i, chunk = tid&7).
struct GlobalTile {
    const bf16_t* base;
    int ld;    // elements
    int rows;  // rows beyond are clamped (their results are masked by the epilogue)
    __device__ __forceinline__ void load(uint4 (&r)[4], int row0, int k0, int tid) const
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(row0 + (tid >> 3) + 32 * i, rows - 1);
            r[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + k0 + (tid & 7) * 8);
        }
    }
};

// Point-MAE first conv (3 -> 128, BN folded, ReLU) evaluated while staging: models/models.py:188-190.
struct Conv1Tile {
    const float* neigh;  // [rows,3]
    const float4* wb;    // [128] = {w_x, w_y, w_z, b} with BatchNorm folded in
    int rows;
    __device__ __forceinline__ void load(uint4 (&r)[4], int row0, int k0, int tid) const
    {
        float4 w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = wb[k0 + (tid & 7) * 8 + e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(row0 + (tid >> 3) + 32 * i, rows - 1);
            const float x = neigh[(size_t)row * 3], y = neigh[(size_t)row * 3 + 1], z = neigh[(size_t)row * 3 + 2];
            bf16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = f2bf(fmaxf(w[e].x * x + w[e].y * y + w[e].z * z + w[e].w, 0.0f));
            r[i] = __builtin_bit_cast(uint4, h);
        }
    }
};

__device__ __forceinline__ void stage_store(char* tile, const uint4 (&r)[4], int tid)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<uint4*>(tile + lds_off((tid >> 3) + 32 * i, tid & 7)) = r[i];
}

typedef f32x4 Acc[4][4];

template <bool SWAP>
__device__ __forceinline__ void compute_stage(Acc& acc, const char* ta, const char* tw, int wr, int wc, int lane)
{
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[4], wf[4];
        const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(ta + lds_off(wr * 64 + i * 16 + (lane & 15), chunk));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wf[j] = *reinterpret_cast<const bf16x8*>(tw + lds_off(wc * 64 + j * 16 + (lane & 15), chunk));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
            }
    }
}

// Runs n_tiles consecutive 128-wide N tiles (starting at tile index nt0) against the block's M tile.
// epi(acc, nt) is called once per finished N tile.  lds: kLdsBytes, 16-byte aligned.
template <bool SWAP, class ALoader, class WLoader, class Epi>
__device__ __forceinline__ void run(const ALoader& A, const WLoader& W, int m0, int nt0, int n_tiles, int KT,
                                    char* lds, Epi&& epi)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    Acc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rw[4];
    A.load(ra, m0, 0, tid);
    W.load(rw, nt0 * BN, 0, tid);
    stage_store(lds, ra, tid);
    stage_store(lds + BM * BK * 2, rw, tid);
    __syncthreads();

    const int total = n_tiles * KT;
    int kt = 0, nt = nt0;
    for (int it = 0; it < total; ++it) {
        char* cur = lds + (it & 1) * kStageBytes;
        char* nxt = lds + ((it + 1) & 1) * kStageBytes;
        int kt_n = kt + 1, nt_n = nt;
        if (kt_n == KT) { kt_n = 0; nt_n = nt + 1; }
        const bool more = it + 1 < total;
        if (more) {
            A.load(ra, m0, kt_n * BK, tid);
            W.load(rw, nt_n * BN, kt_n * BK, tid);
        }
        compute_stage<SWAP>(acc, cur, cur + BM * BK * 2, wr, wc, lane);
        if (more) {
            stage_store(nxt, ra, tid);
            stage_store(nxt + BM * BK * 2, rw, tid);
        }
        if (kt == KT - 1) {
            epi(acc, nt);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        kt = kt_n; nt = nt_n;
    }
}

// XCD-aware bijective remap of the linear workgroup id (guide T1): blocks b and b+8 share an XCD
// (and its L2), so give every XCD a CONTIGUOUS chunk of the tile order.
__device__ __forceinline__ int xcd_remap(int orig, int nwg)
{
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

}  // namespace gemm
