// bf16 MFMA GEMM core for gfx950: C[M,N] = A[M,K] . W[N,K]^T ("NT": both operands K-contiguous,
// the layout of nn.Linear weights, of token matrices and of the query / bank matrices).
//
// Shape<BM, BN, WAVES, STAGES>: block tile BM x BN x 64; waves in a (WAVES / (BN/64)) x (BN/64) grid, every wave owning
// (MI*16) x 64 outputs as MI x 4 tiles of v_mfma_f32_16x16x32 (MI = 4: 64 accumulator VGPRs, MI = 8: 128).
//   S128 = Shape<128,128,4,2>:  64 KiB LDS, 2 blocks/CU -- every network GEMM (two 4-wave blocks per CU hide each
//                               other's LDS-DMA issue and barrier stalls; measured fastest on all ViT / Point-MAE shapes)
//   S2x2 = Shape<256,256,8,2>: 128 KiB LDS, 1 block/CU, 8 waves of 128 x 64 -- distance GEMM fallback / A-B reference for
//                               the 4-wave 256 x 256 shape of gemm_wide.h, which is what the distance GEMM runs
// LDS stage = A tile + W tile, [rows][64 bf16] with 128-byte rows whose 16-byte chunks are XOR-swizzled by
// (row & 7): the ds_read_b128 fragment reads (16 rows x 4 k-chunks per 16-lane service group) touch 16
// distinct 16-byte slots of the 256-byte bank row -> conflict-free.
// Staging is LDS-DMA (global_load_lds_dwordx4, no staging VGPRs): step t issues the loads of step
// t+STAGES-1, computes on stage t, then waits for step t+1 to land (with 3 stages a COUNTED s_waitcnt vmcnt leaves the
// youngest step in flight across the raw s_barrier; guide T3/T4) -- ONE barrier per K-step.
// The (n-tile, k-tile) iteration space is flattened so a block that owns several N tiles (the distance
// GEMM's running-min loop) keeps the pipeline full across tile boundaries.
//
// Orientation: with SWAP = true the weight fragment is fed as the MFMA "A" operand, so the accumulator
// tile is C^T: every lane then holds 4 CONSECUTIVE n for one m, which makes the row-major epilogue
// stores 16-byte (f32) / 8-byte (bf16) vectors and lets a per-query running min live in one lane.
// SWAP = false gives 4 consecutive m per lane (used for transposed stores).
#pragma once
#include <type_traits>

#include "common.h"

namespace gemm {

constexpr int BK = 64;

// Shape<BM, BN, WAVES, STAGES>: waves in a (WAVES/WN) x WN grid with WN = BN/64; a wave owns (MI*16) x 64 outputs.
template <int BM_, int BN_, int WAVES_, int STAGES_>
struct Shape {
    static_assert(STAGES_ == 2 || STAGES_ == 3, "2 or 3 LDS stages");
    static constexpr int BM = BM_, BN = BN_, WAVES = WAVES_, STAGES = STAGES_, THREADS = WAVES_ * 64;
    static constexpr int WN = BN_ / 64, WM = WAVES_ / WN, MI = BM_ / WM / 16;
    static constexpr int STAGE_BYTES = (BM_ + BN_) * BK * 2;
    static constexpr int LDS_BYTES = STAGES_ * STAGE_BYTES;
    static constexpr int GL = BM_ / WAVES_ / 8 + BN_ / WAVES_ / 8;  // LDS-DMA instructions per wave per K-step
    static constexpr int WAVES_PER_SIMD = (160 * 1024 / LDS_BYTES >= 2 ? 2 : 1) * WAVES_ / 4;
    static_assert(WM * WN == WAVES_ && MI * 16 * WM == BM_ && (MI == 4 || MI == 8), "wave grid");
};
typedef Shape<128, 128, 4, 2> S128;   //  64 KiB LDS, 2 blocks/CU: small grids
typedef Shape<256, 256, 8, 2> S2x2;   // 128 KiB, 1 block/CU: half the L2->LDS bytes per FLOP of S128

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ (row & 7)) << 4); }

// Operand tile stagers.  stage<ROWS, WAVES>(tile, row0, k0, tid) fills one [ROWS][64] bf16 LDS tile.
//
// GlobalTile uses LDS-DMA: one wave-instruction writes 1 KiB = 8 rows x 128 B at (wave-uniform base) +
// lane*16, i.e. lane l lands on row l>>3, PHYSICAL chunk l&7; the XOR swizzle therefore goes on the
// per-lane SOURCE address (logical chunk = (l&7) ^ (row&7), guide rule 21) and the same XOR is applied
// by the fragment reads.
struct GlobalTile {
    const bf16_t* base;
    int ld;    // elements
    int rows;  // rows beyond are clamped (their results are masked by the epilogue)
    template <int ROWS, int WAVES>
    __device__ __forceinline__ void stage(char* tile, int row0, int k0, int tid) const
    {
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int PER_WAVE = ROWS / WAVES;
#pragma unroll
        for (int j = 0; j < PER_WAVE / 8; ++j) {
            const int r = wave * PER_WAVE + j * 8 + (lane >> 3);      // row within the tile
            const int row = min(row0 + r, rows - 1);
            const int chunk = (lane & 7) ^ (r & 7);                   // logical chunk stored at physical l&7
            const bf16_t* src = base + (size_t)row * ld + k0 + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + (wave * PER_WAVE + j * 8) * (BK * 2)),
                                             16, 0, 0);
        }
    }
    // The K loop's form of the same pieces.  stage() above computes every piece's 64-bit source address and its LDS base with vector
    // arithmetic (~9 VALU instructions + a v_readfirstlane per piece, 75 per K-step of a 128 x 128 block), in the very phase that
    // issues the LDS-DMA -- and that issue path, not the L2, is what bounds the 128 x 128 kernel (profiles/r4_notes.md section 12:
    // no DMA at all 1 110 TFLOP/s, sources L1-hot 860, production 735-760 on the K = 3 072 product).  Here a lane's byte offset
    // from the tile's first row is computed ONCE per tile (offsets()); the row base, the K offset and the LDS destination stay
    // scalar, so a piece is an s_mov m0 and one global_load_lds with an SGPR base.
    // PRECONDITION (both functions): row0 < rows -- a tile starts on an existing row; only rows INSIDE the tile are clamped.  Every
    // caller derives row0 from a grid sized by ceil(rows / tile), so this holds; with row0 >= rows the unsigned offset would wrap.
    template <int ROWS, int WAVES>
    __device__ __forceinline__ void offsets(unsigned (&voff)[ROWS / WAVES / 8], int row0, int lane, int wave) const
    {
        constexpr int PER_WAVE = ROWS / WAVES;
        const int chunk = (lane & 7) ^ (lane >> 3);   // (r & 7) of stage(): the piece's first row is a multiple of 8
#pragma unroll
        for (int j = 0; j < PER_WAVE / 8; ++j) {
            const int row = min(row0 + wave * PER_WAVE + j * 8 + (lane >> 3), rows - 1);
            voff[j] = (unsigned)(row - row0) * (unsigned)(ld * 2) + (unsigned)(chunk * 16);   // row0 < rows: never negative
        }
    }
    template <int ROWS, int WAVES>
    __device__ __forceinline__ void stage_lean(const unsigned (&voff)[ROWS / WAVES / 8], char* tile, int row0, int k0, int wave) const
    {
        constexpr int PER_WAVE = ROWS / WAVES;
        const char* sb = reinterpret_cast<const char*>(base) + ((size_t)row0 * ld + k0) * 2;   // block-uniform
#pragma unroll
        for (int j = 0; j < PER_WAVE / 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voff[j]),
                                             (__attribute__((address_space(3))) void*)(tile + (wave * PER_WAVE + j * 8) * (BK * 2)),
                                             16, 0, 0);
    }
};

// One output of the Point-MAE first conv (3 -> 128, BatchNorm folded, ReLU), models/models.py:188-190.  The FMA chain is
// spelled out so that every kernel evaluating it (and every compiler schedule) rounds the same way.
__device__ __forceinline__ float conv1_act(const float4& w, float x, float y, float z)
{
    return fmaxf(__builtin_fmaf(w.z, z, __builtin_fmaf(w.y, y, __builtin_fmaf(w.x, x, w.w))), 0.0f);
}

// Point-MAE first conv evaluated while staging.
// Computed values go through registers and ds_write_b128 (thread t: rows (t>>3) + (THREADS/8) i, chunk t&7).
struct Conv1Tile {
    const float* neigh;  // [rows,3]
    const float4* wb;    // [128] = {w_x, w_y, w_z, b} with BatchNorm folded in
    int rows;
    // A block stages the SAME rows once per (N tile, K tile) step -- four times in the encoder's first stage -- so the
    // coordinates a thread needs are fetched from memory once per block and kept: every further fetch would put one HBM
    // round trip in front of that step's MFMAs (1.78 -> see profiles/r1_notes.md).
    mutable float px[4], py[4], pz[4];
    mutable int have_row0 = -1;
    template <int ROWS, int WAVES>
    __device__ __forceinline__ void stage(char* tile, int row0, int k0, int tid) const
    {
        constexpr int STEP = WAVES * 8;  // rows covered per pass
        static_assert(ROWS / STEP <= 4, "coordinate cache size");
        float4 w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = wb[k0 + (tid & 7) * 8 + e];
        if (row0 != have_row0) {  // block-uniform
            have_row0 = row0;
#pragma unroll
            for (int i = 0; i < ROWS / STEP; ++i) {
                const int row = min(row0 + (tid >> 3) + STEP * i, rows - 1);
                px[i] = neigh[(size_t)row * 3]; py[i] = neigh[(size_t)row * 3 + 1]; pz[i] = neigh[(size_t)row * 3 + 2];
            }
        }
#pragma unroll
        for (int i = 0; i < ROWS / STEP; ++i) {
            const int r = (tid >> 3) + STEP * i;
            const float x = px[i], y = py[i], z = pz[i];
            bf16x8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = f2bf(conv1_act(w[e], x, y, z));
            *reinterpret_cast<bf16x8*>(tile + lds_off(r, tid & 7)) = h;
        }
    }
};


typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// F16 = true: the staged 16-bit operands are IEEE half (same MFMA rate, 3 more mantissa bits: used by the
// distance GEMM, whose operands are normalised features); false: bfloat16 (networks).
template <class S, bool SWAP, bool F16, class AccT>
__device__ __forceinline__ void compute_stage(AccT& acc, const char* ta, const char* tw, int wr, int wc, int lane)
{
    using frag = typename std::conditional<F16, f16x8, bf16x8>::type;
    if constexpr (S::MI == 4) {
        // all fragment reads of the K-step are issued up front (both 32-deep halves, 64 VGPRs): the second
        // half's LDS latency hides behind the first half's MFMAs instead of being exposed a second time
        frag af[2][S::MI], wf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < S::MI; ++i)
                af[kk][i] = *reinterpret_cast<const frag*>(ta + lds_off(wr * (S::MI * 16) + i * 16 + (lane & 15), chunk));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[kk][j] = *reinterpret_cast<const frag*>(tw + lds_off(wc * 64 + j * 16 + (lane & 15), chunk));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < S::MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = SWAP ? mfma16(wf[kk][j], af[kk][i], acc[i][j]) : mfma16(af[kk][i], wf[kk][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    } else {  // 128 x 64 per wave: 128 accumulator VGPRs leave room for one half's fragments at a time
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            frag af[S::MI], wf[4];
            const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < S::MI; ++i)
                af[i] = *reinterpret_cast<const frag*>(ta + lds_off(wr * (S::MI * 16) + i * 16 + (lane & 15), chunk));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const frag*>(tw + lds_off(wc * 64 + j * 16 + (lane & 15), chunk));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < S::MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = SWAP ? mfma16(wf[j], af[i], acc[i][j]) : mfma16(af[i], wf[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N == 0 || N == 6 || N == 8, "add the immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

__device__ __forceinline__ void block_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_writes (Conv1Tile) are in LDS
    __builtin_amdgcn_s_barrier();
}

// What an N tile's accumulators start from: zeros (every network product; the distance GEMM has its own, l2min.hip)
struct ZeroInit {
    template <class Acc>
    __device__ __forceinline__ void operator()(Acc& acc, int) const
    {
#pragma unroll
        for (auto& row : acc)
#pragma unroll
            for (auto& v : row) v = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};

// Runs n_tiles consecutive BN-wide N tiles (starting at tile index nt0) against the block's M tile, over
// K-steps [kt_begin, kt_begin + KT).  epi(acc, nt, scratch) is called once per finished N tile; init(acc, nt) sets the
// accumulators an N tile starts from.
template <class S, bool SWAP, bool F16 = false, class ALoader, class WLoader, class Epi, class Init = ZeroInit>
__device__ __forceinline__ void run(const ALoader& A, const WLoader& W, int m0, int nt0, int n_tiles, int KT,
                                    char* lds, Epi&& epi, int kt_begin = 0, Init&& init = Init())
{
    constexpr int BM = S::BM, ST = S::STAGES, AHEAD = ST - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool LEAN = std::is_same<ALoader, GlobalTile>::value && std::is_same<WLoader, GlobalTile>::value;
    // (tried and measured slower on MI355X, profiles/r1_notes.md: spreading the LDS-DMA pieces between MFMA
    //  groups -- 735 -> 650 TFLOP/s on the ViT qkv shape -- and a 3-stage counted-vmcnt pipeline at 256x128)
    constexpr int BN = S::BN;
    const int wr = wave / S::WN, wc = wave % S::WN;
    f32x4 acc[S::MI][4];
    init(acc, nt0);

    const int total = n_tiles * KT;
    int kt_s = 0, nt_s = nt0;  // coordinates of the next K-step to STAGE
    unsigned va[LEAN ? BM / S::WAVES / 8 : 1], vw[LEAN ? BN / S::WAVES / 8 : 1];   // GlobalTile::offsets of this wave's pieces
    int vw_tile = nt0;
    if constexpr (LEAN) {
        A.template offsets<BM, S::WAVES>(va, m0, lane, wave);
        W.template offsets<BN, S::WAVES>(vw, nt0 * BN, lane, wave);
    }
    auto stage_next = [&](int slot) {
        char* buf = lds + slot * S::STAGE_BYTES;
        if constexpr (LEAN) {
            if (nt_s != vw_tile) {  // block-uniform: the next N tile's rows (clamped at the ragged edge)
                vw_tile = nt_s;
                W.template offsets<BN, S::WAVES>(vw, nt_s * BN, lane, wave);
            }
            A.template stage_lean<BM, S::WAVES>(va, buf, m0, (kt_begin + kt_s) * BK, wave);
            W.template stage_lean<BN, S::WAVES>(vw, buf + BM * BK * 2, nt_s * BN, (kt_begin + kt_s) * BK, wave);
        } else {
            A.template stage<BM, S::WAVES>(buf, m0, (kt_begin + kt_s) * BK, tid);
            W.template stage<BN, S::WAVES>(buf + BM * BK * 2, nt_s * BN, (kt_begin + kt_s) * BK, tid);
        }
        if (++kt_s == KT) { kt_s = 0; ++nt_s; }
    };
    stage_next(0);
    if constexpr (AHEAD == 2) {
        if (total >= 2) {
            stage_next(1);
            wait_vmcnt<S::GL>();  // step 0 landed, step 1 may still be in flight
        } else wait_vmcnt<0>();
    } else wait_vmcnt<0>();
    block_barrier();

    int kt = 0, nt = nt0, slot = 0, slot_s = AHEAD % ST;
    for (int it = 0; it < total; ++it) {
        const char* cur = lds + slot * S::STAGE_BYTES;
        const bool more = it + AHEAD < total;
        if (more) stage_next(slot_s);  // that stage was last read in step it-1, which every wave has left
        compute_stage<S, SWAP, F16>(acc, cur, cur + BM * BK * 2, wr, wc, lane);
        if (kt == KT - 1) {
            epi(acc, nt, n_tiles == 1 ? lds + slot_s * S::STAGE_BYTES : nullptr);  // scratch: a stage nobody reads or fills (single-tile runs only)
            // (an LDS-staged 16-byte coalesced bf16 store through that scratch measured SLOWER than the direct
            //  8-byte stores: 3.32 vs 3.06 ms on the 4.2M x 512 x 256 product, profiles/r1_notes.md)
            if (it + 1 < total) init(acc, nt + 1);   // block-uniform
        }
        // step it+1 must be in LDS before anyone reads it; with 3 stages the loads issued in THIS step
        // (step it+2) may stay in flight across the barrier
        if constexpr (AHEAD == 2) {
            if (more) wait_vmcnt<S::GL>(); else wait_vmcnt<0>();
        } else wait_vmcnt<0>();
        block_barrier();
        if (++kt == KT) { kt = 0; ++nt; }
        slot = slot + 1 == ST ? 0 : slot + 1;
        slot_s = slot_s + 1 == ST ? 0 : slot_s + 1;
    }
}

// Row-contiguous stores from the MFMA accumulator layout.  In the swapped orientation a lane holds 4 consecutive columns of
// ONE row, so a store instruction of a wave's 16 x 64 block touches 16 rows x 32 bytes (bf16): 64 separate line accesses for
// the texture addresser -- measured as 30 % of the fc1 GEMM (profiles/r2_notes.md).  A wave parks the block (as bf16) in a
// private 2 KiB LDS scratch -- 8-byte slot s = 4 j + g of row r in 16-byte unit (s >> 1) ^ (r >> 1), half s & 1: conflict-free
// for the writes and for the reads -- and takes it back as two 16-byte units per lane: rows R = lane >> 3 and R + 8, columns
// 8 (lane & 7) .. + 7, i.e. 8 rows x 128 contiguous bytes per store instruction.  The read-back is inline asm: a
// compiler-visible LDS read after LDS-DMA gets an s_waitcnt vmcnt(0) (a drain of the prefetch queue).
struct RowStore16 {
    char* scratch;
    unsigned rd0, rd1;
    int r, g, R, u;
    __device__ __forceinline__ void init(char* wave_scratch, int lane)
    {
        scratch = wave_scratch;
        r = lane & 15; g = lane >> 4; R = lane >> 3; u = lane & 7;
        rd0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(scratch + R * 128 + ((u ^ (R >> 1)) << 4));
        rd1 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(scratch + (R + 8) * 128 + ((u ^ ((R + 8) >> 1)) << 4));
    }
    __device__ __forceinline__ void park(const bf16x4 (&h)[4]) const
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s8 = 4 * j + g;
            *reinterpret_cast<bf16x4*>(scratch + r * 128 + (((s8 >> 1) ^ (r >> 1)) << 4) + ((s8 & 1) << 3)) = h[j];
        }
    }
    __device__ __forceinline__ void fetch_issue(uint4& t0, uint4& t1) const
    {
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(t0), "=&v"(t1) : "v"(rd0), "v"(rd1) : "memory");
    }
    // the fetched units may only be STORED after this (memory operations do not cross it)
    __device__ __forceinline__ void fetch_wait() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};
constexpr int kRowStoreScratch = 2048;   // bytes per wave

// The same for fp32 values: a 16-row x 32-column half block (two j) is 2 KiB -- 16-byte unit 4 (j & 1) + g of row r at
// (unit ^ (r >> 1)); read back as rows R / R + 8, columns 4 (lane & 7) .. + 3 of the half.
struct RowStore32 {
    char* scratch;
    unsigned rd0, rd1;
    int r, g, R, u;
    __device__ __forceinline__ void init(char* wave_scratch, int lane)
    {
        scratch = wave_scratch;
        r = lane & 15; g = lane >> 4; R = lane >> 3; u = lane & 7;
        rd0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(scratch + R * 128 + ((u ^ (R >> 1)) << 4));
        rd1 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(scratch + (R + 8) * 128 + ((u ^ ((R + 8) >> 1)) << 4));
    }
    __device__ __forceinline__ void park(const f32x4& v0, const f32x4& v1) const   // columns j = 2 ch and 2 ch + 1
    {
        *reinterpret_cast<f32x4*>(scratch + r * 128 + ((g ^ (r >> 1)) << 4)) = v0;
        *reinterpret_cast<f32x4*>(scratch + r * 128 + (((4 + g) ^ (r >> 1)) << 4)) = v1;
    }
    __device__ __forceinline__ void fetch(f32x4& t0, f32x4& t1) const
    {
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(t0), "=&v"(t1) : "v"(rd0), "v"(rd1) : "memory");
    }
};

// XCD-aware bijective remap of the linear workgroup id (guide T1): blocks b and b+8 share an XCD
// (and its L2), so give every XCD a CONTIGUOUS chunk of the tile order.
__device__ __forceinline__ int xcd_remap(int orig, int nwg)
{
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

}  // namespace gemm
