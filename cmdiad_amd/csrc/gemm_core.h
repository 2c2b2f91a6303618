// bf16 MFMA GEMM core for gfx950: C[M,N] = A[M,K] . W[N,K]^T ("NT": both operands K-contiguous,
// the layout of nn.Linear weights, of token matrices and of the query / bank matrices).
//
// Tile 128 x 128 x 64, 256 threads = 4 waves in a 2 x 2 grid, each wave owns a 64 x 64 output as
// 4 x 4 tiles of v_mfma_f32_16x16x32_bf16 (16 f32x4 accumulators = 64 VGPRs).
// LDS: two 32 KiB stages, each = A tile + W tile, [128 rows][64 bf16] with 128-byte rows whose 16-byte
// chunks are XOR-swizzled by (row & 7): the ds_read_b128 fragment reads (16 rows x 4 k-chunks per
// 16-lane service group) then touch 16 distinct 16-byte slots of the 256-byte bank row -> conflict-free.
// Staging is LDS-DMA (global_load_lds_dwordx4): the loads of step t+1 are issued before the MFMAs of
// step t straight into the other LDS stage; ONE barrier per K-step.
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

// Operand tile stagers.  stage(tile, row0, k0, tid) fills one [128][64] bf16 LDS tile.
//
// GlobalTile uses LDS-DMA (global_load_lds_dwordx4, guide §5): no staging VGPRs, no ds_write.  One
// wave-instruction writes 1 KiB = 8 rows x 128 B at (wave-uniform base) + lane*16, i.e. lane l lands on
// row l>>3, PHYSICAL chunk l&7; the XOR swizzle therefore goes on the per-lane SOURCE address
// (logical chunk = (l&7) ^ (row&7), guide rule 21) and the same XOR is applied by the fragment reads.
// Each of the 4 waves issues 4 such instructions per operand tile.  Completion is tracked by vmcnt;
// the __syncthreads() that ends a K-step drains it (2-phase schedule of guide T3/T4 "minimum").
struct GlobalTile {
    const bf16_t* base;
    int ld;    // elements
    int rows;  // rows beyond are clamped (their results are masked by the epilogue)
    __device__ __forceinline__ void stage(char* tile, int row0, int k0, int tid) const
    {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave * 32 + j * 8 + (lane >> 3);           // row within the tile
            const int row = min(row0 + r, rows - 1);
            const int chunk = (lane & 7) ^ (r & 7);                   // logical chunk stored at physical l&7
            const bf16_t* src = base + (size_t)row * ld + k0 + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + (wave * 32 + j * 8) * (BK * 2)),
                                             16, 0, 0);
        }
    }
};

// Point-MAE first conv (3 -> 128, BN folded, ReLU) evaluated while staging: models/models.py:188-190.
// Computed values go through registers and ds_write_b128 (thread t: rows (t>>3) + 32 i, chunk t&7).
struct Conv1Tile {
    const float* neigh;  // [rows,3]
    const float4* wb;    // [128] = {w_x, w_y, w_z, b} with BatchNorm folded in
    int rows;
    __device__ __forceinline__ void stage(char* tile, int row0, int k0, int tid) const
    {
        float4 w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = wb[k0 + (tid & 7) * 8 + e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 3) + 32 * i;
            const int row = min(row0 + r, rows - 1);
            const float x = neigh[(size_t)row * 3], y = neigh[(size_t)row * 3 + 1], z = neigh[(size_t)row * 3 + 2];
            bf16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = f2bf(fmaxf(w[e].x * x + w[e].y * y + w[e].z * z + w[e].w, 0.0f));
            *reinterpret_cast<bf16x8*>(tile + lds_off(r, tid & 7)) = h;
        }
    }
};

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
                                    char* lds, Epi&& epi, int kt_begin = 0)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    Acc acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    A.stage(lds, m0, kt_begin * BK, tid);
    W.stage(lds + BM * BK * 2, nt0 * BN, kt_begin * BK, tid);
    __syncthreads();

    const int total = n_tiles * KT;
    int kt = 0, nt = nt0;
    for (int it = 0; it < total; ++it) {
        char* cur = lds + (it & 1) * kStageBytes;
        char* nxt = lds + ((it + 1) & 1) * kStageBytes;
        int kt_n = kt + 1, nt_n = nt;
        if (kt_n == KT) { kt_n = 0; nt_n = nt + 1; }
        if (it + 1 < total) {  // every wave finished reading `nxt` before the barrier that ended step it-1
            A.stage(nxt, m0, (kt_begin + kt_n) * BK, tid);
            W.stage(nxt + BM * BK * 2, nt_n * BN, (kt_begin + kt_n) * BK, tid);
        }
        compute_stage<SWAP>(acc, cur, cur + BM * BK * 2, wr, wc, lane);
        if (kt == KT - 1) {
            epi(acc, nt);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();  // also drains the LDS-DMA of the next stage (vmcnt(0) before s_barrier)
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
