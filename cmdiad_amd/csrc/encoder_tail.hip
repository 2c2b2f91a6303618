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
#include <type_traits>

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
[[maybe_unused]] constexpr int AHEAD = NST - 1;  // (test-only kernels)
constexpr int TAIL_LDS = A2_BYTES + A3_BYTES + NST * W_STAGE;  // 160 KiB at 4 stages
static_assert(NST >= 2 && NST <= 4 && TAIL_LDS <= 160 * 1024, "weight stages");
constexpr int kTailCUs = 256;                   // one persistent block per CU (MI355X)

__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct TailParams {
    int M, Mg;
    const float* gb;   // [groups, 512]
    const float* b4;   // [384]
    float* tok;        // [groups, 384]
    int ablate;        // test-only build: bit 0 no weight stream after the prologue, 1 no MFMAs, 2 no fragment reads, 3 no h3 write, 4 no phase barriers, 5 no h2 load, 6 no final reduction
};

#ifdef CMDIAD_AB_VARIANTS  // the lock-step form (both waves of a SIMD in the same part of a phase): A/B reference of the test-only build
#include "ab/encoder_tail_lockstep.inc"
#endif  // CMDIAD_AB_VARIANTS

// ------------------------------------------------------------------------------------------------
// PERSISTENT form (production): 256 blocks walk the row tiles, ONE barrier per phase, fragment reads half a phase ahead.
// What the measurements of the kernels above say (profiles/r2_notes.md): a workgroup barrier of 8 waves costs ~170 cycles
// in which nothing issues, against 256 MFMA cycles per wave and phase; with one block per row tile and 160 KiB of LDS
// nothing overlaps the dispatch of the next workgroup and its first loads (1.3 of 3.0 ms with every phase emptied); an
// LDS-DMA piece costs its issuing wave 100-185 cycles inside a busy phase, and with both waves of a SIMD issuing at the same
// point of the phase those cycles are not hidden.  The two-group schedule hides the fragment reads but pays two barriers per
// phase and came out even.  Here:
//   * a phase is   [reads of its second K half | 8 MFMAs of the first half | wait, BARRIER | issue | reads of the NEXT phase's
//     first half | 8 MFMAs of the second half]: every LDS read has 8 MFMAs between issue and use, one barrier per phase;
//   * only waves 0-3 (one per SIMD) issue the weight stream, four pieces per tile, right after the barrier: the partner wave
//     of the SIMD runs its MFMAs meanwhile.  At the barrier of phase t every wave holds all of tile t in registers, so tile
//     t + 4 goes into the stage of tile t: three to four tiles of lead.  The counted waits come from a running count of the
//     pieces a wave has issued (weights, h2 K-blocks, group-bias loads): "tile t + 1 landed" = all but the pieces after it;
//   * the stream never stops: modulo 40 it runs into the next row tile (40 is a multiple of the 4 stages), whose h2 arrives
//     during the last chunk -- K-block u is dead after the barrier of phase 30 + u and is refilled right there;
//   * the second product walks K outside and the three output chunks inside: h3 fragments are read once per K-step;
//   * the h3 chunk is the one true dependency: after the fourth K-step a wave writes its part, one extra barrier, then the
//     first reads of the next product (four exposed LDS latencies per row tile);
//   * the group maxima of a finished tile meet in the h3 buffer (dead until phase 3 of the next tile).
// Same arithmetic per output element (K order per accumulator, bf16 rounding of h3): identical tokens.
// ------------------------------------------------------------------------------------------------
template <bool ABL>   // ABL: timing ablations (results are garbage), instantiated in the test-only build
__global__ __launch_bounds__(TW * 64, 1) void encoder_tail_persist_kernel(GlobalTile H2, GlobalTile W3, GlobalTile W4, TailParams p, int n_tiles)
{
    static_assert(NST == 4, "the schedule is written for four weight stages");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* A2 = lds;
    char* A3 = lds + A2_BYTES;
    char* WS = A3 + A3_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const bool lead = wave < 4;
    const int my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    // ---- issue paths of waves 0-3.  One LDS-DMA piece = 8 rows x 128 B: lane l fetches the 16-byte chunk (l & 7) ^ (row & 7) of row
    // l >> 3 (the swizzle goes on the source address, gemm_core.h).  Inline asm in the scalar-base form (uniform 64-bit base in
    // SGPRs + one 32-bit per-lane offset): with per-lane 64-bit pointers the compiler hoists the addresses of all 40 tiles x 4
    // pieces out of the row-tile loop and spills them.  M0 = LDS byte address of the piece; nothing else here uses M0.
    const int prow = (wave & 3) * 32 + (lane >> 3);
    const unsigned pch = (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    const unsigned off3 = (unsigned)prow * 512u + pch, off4 = (unsigned)prow * 1024u + pch;   // row pitches: 256 / 512 bf16
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // (inline asm is outside the compiler's hazard tracking: an SGPR base that the compiler has just restored with v_readlane -- a
    //  VALU write -- needs 5 wait states before a VMEM instruction reads it, so the base goes through an SALU copy inside the
    //  statement; one wait state between the M0 write and the LDS-DMA instruction)
    auto dma = [&](const char* ubase, unsigned voff, unsigned lds_addr) {
        unsigned long long sb;
        asm volatile("s_mov_b64 %0, %2\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                     : "=&s"(sb) : "v"(voff), "s"(ubase), "s"(lds_addr) : "memory");
    };
    // weight tile (c, u) of the 40-tile sequence: u < 4 -> W3b rows [128 c, +128), K-step u; else v = u - 4: K-step k2 = v / 3 of
    // W4 rows [128 (v % 3), +128), columns 128 c + 64 k2
    auto stage_w = [&](int c, int u, int slot) {   // wave-uniform arguments
        const unsigned dst = lds0 + A2_BYTES + A3_BYTES + slot * W_STAGE + (wave & 3) * 32 * 128;
        if (u < 4) {
            const char* ub = reinterpret_cast<const char*>(W3.base) + ((size_t)c * 128 * 256 + u * BK) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) dma(ub + j * 8 * 512, off3, dst + j * 8 * 128);
        } else {
            const int v = u - 4, k2 = v >= 3 ? 1 : 0, o = v - 3 * k2;
            const char* ub = reinterpret_cast<const char*>(W4.base) + ((size_t)o * 128 * 512 + c * 128 + k2 * BK) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) dma(ub + j * 8 * 1024, off4, dst + j * 8 * 128);
        }
    };
    auto stage_h = [&](int m0n, int kb) {   // K-block kb of the h2 tile starting at row m0n (rows past M clamp: masked by the store)
        const char* ub = reinterpret_cast<const char*>(H2.base) + ((size_t)m0n * 256 + kb * BK) * 2;
        const unsigned dst = lds0 + kb * KB_BYTES + (wave & 3) * 32 * 128;
        const int last = p.M - 1 - m0n;   // >= 0: the tile starts inside the matrix
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(ub, (unsigned)min(prow + j * 8, last) * 512u + pch, dst + j * 8 * 128);
    };
    // Counted wait of phase (c, u), before its barrier: the tile of the NEXT phase has landed.  That tile was issued three phases
    // ago; everything a wave issued after it may stay in flight (vmcnt retires in order): two weight tiles (4 pieces each), plus 4
    // per h2 K-block (issued after the barriers of phases 0-3 of the last chunk when a row tile follows) and 4 for the group-bias
    // loads (second half of phase 3) that fall into the three phases in between.  The table below is that count per u:
    //   any chunk but the last (and the last one... see `zone`):  8 8 8 8 12 12 12 8 8 8
    //   last chunk, a row tile follows:                           8 12 16 20 24 20 16 8 8 8
    //   last chunk of the last row tile (stream ends at tile 39, no group-bias fetch):  8 x 7, then 4, 0, no wait
    auto wait_next = [&](auto UC, int zone) {   // zone 0 / 1 / 2 as listed, wave-uniform
        constexpr int u = decltype(UC)::value;
        constexpr int generic = (u >= 4 && u <= 6) ? 12 : 8;
        constexpr int with_next = u == 1 ? 12 : u == 2 ? 16 : u == 3 ? 20 : u == 4 ? 24 : u == 5 ? 20 : u == 6 ? 16 : 8;
        auto w = [](auto NC) {
            constexpr int n = decltype(NC)::value;
            if constexpr (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (n == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        };
        if (zone == 2) {
            if constexpr (u <= 6) w(std::integral_constant<int, 8>{});
            else if constexpr (u == 7) w(std::integral_constant<int, 4>{});
            else if constexpr (u == 8) w(std::integral_constant<int, 0>{});
        } else if (zone == 1) w(std::integral_constant<int, with_next>{});
        else w(std::integral_constant<int, generic>{});
    };
    f32x4 gbv[4];
    auto fetch_gb = [&](int m0t, int c) {   // this wave's 32 rows share a group (32 | Mg)
        const float* gb = p.gb + (size_t)(min(m0t + wr * 32, p.M - 1) / p.Mg) * 512 + wc * 64 + (lane >> 4) * 4 + c * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gbv[j]) : "v"(gb + j * 16) : "memory");
    };

    // ---- fragment reads: row * 128 + ((chunk ^ (row & 7)) << 4) with chunk = 4 kk + (lane >> 4): kk flips bit 6, the 16-row
    // blocks i / j add 2 KiB.  h3 is written as 8-byte units: column 16 j + 4 (lane >> 4) of the wave's half = chunk 2 j + (g >> 1).
    const int swz = ((lane >> 4) ^ (lane & 7)) << 4;
    const int a_frag = (wr * 32 + (lane & 15)) * 128 + swz, w_frag = (wc * 64 + (lane & 15)) * 128 + swz;
    const int h_dst = wc * KB_BYTES + (wr * 32 + (lane & 15)) * 128 + ((((lane >> 5)) ^ (lane & 7)) << 4) + ((lane >> 4) & 1) * 8;
    bf16x8 af[2][2], wf[2][4];
    auto read_half = [&](int kk, const char* ta, bool load_a, const char* tw) {   // kk, load_a: compile-time at every call
        if (ABL && (p.ablate & 4)) return;
        if (load_a) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[kk][i] = *reinterpret_cast<const bf16x8*>(ta + ((a_frag + i * 2048) ^ (kk << 6)));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[kk][j] = *reinterpret_cast<const bf16x8*>(tw + ((w_frag + j * 2048) ^ (kk << 6)));
    };
    auto mfma_half = [&](int kk, f32x4 (&acc)[2][4]) {
        if (ABL && (p.ablate & 2)) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[kk][j], af[kk][i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: first tile's h2, its first group bias, weight tiles 0..3; first half of phase 0
    fetch_gb((int)blockIdx.x * TM, 0);
    if (lead) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) stage_h((int)blockIdx.x * TM, kb);
        stage_w(0, 0, 0); stage_w(0, 1, 1); stage_w(0, 2, 2); stage_w(0, 3, 3);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // h2, the group bias and tile 0 landed; tiles 1-3 in flight
    }
    block_barrier();
    read_half(0, A2, true, WS);

    float* s_part = reinterpret_cast<float*>(A3);  // [4 row blocks][384]: the h3 buffer is dead between two row tiles
    f32x4 acc3[2][4], acco[3][2][4];

#pragma unroll 1
    for (int k = 0; k < my; ++k) {
        const int m0 = ((int)blockIdx.x + k * (int)gridDim.x) * TM;
        const bool has_next = k + 1 < my;
        const int m0n = m0 + (int)gridDim.x * TM;
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acco[o][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // one chunk = ten phases; S0 = stage of its first weight tile ((10 c) & 3: 0 for even c, 2 for odd c)
        auto chunk = [&](auto S0C, int c) {
            constexpr int S0 = decltype(S0C)::value;
            const int zone = c < 3 ? 0 : has_next ? 1 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc3[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto phase = [&](auto UC) {
                constexpr int u = decltype(UC)::value;
                const int slot = (S0 + u) & 3;
                const char* ta = u < 4 ? A2 + u * KB_BYTES : A3 + ((u - 4) / 3) * KB_BYTES;
                const bool load_a = u < 4 || (u - 4) % 3 == 0;
                const char* tw = WS + slot * W_STAGE;
                // the phase after this one: (c, u + 1), or phase 0 of the next chunk / the next row tile
                const int un = u == 9 ? 0 : u + 1;
                const char* tan = un < 4 ? A2 + un * KB_BYTES : A3 + ((un - 4) / 3) * KB_BYTES;
                const bool load_an = un < 4 || (un - 4) % 3 == 0;
                const char* twn = WS + ((slot + 1) & 3) * W_STAGE;

                read_half(1, ta, load_a, tw);
                if (u < 4) mfma_half(0, acc3);
                else mfma_half(0, acco[(u - 4) % 3]);
                if (lead && !(ABL && (p.ablate & 1))) wait_next(UC, zone);
                if (!(ABL && (p.ablate & 16))) block_barrier();
                if (lead && !(ABL && (p.ablate & 1))) {
                    // every wave holds all of this phase's tile in registers: its stage takes the tile four ahead -- (c, u + 4),
                    // (c + 1, u - 6), or the next row tile's (0, u - 6); K-block u of the h2 tile is dead in the last chunk
                    const int ic = u + 4 < 10 ? c : c + 1, iu = u + 4 < 10 ? u + 4 : u - 6;
                    if (ic < 4) stage_w(ic, iu, slot);
                    else if (has_next) stage_w(0, iu, slot);
                    if (c == 3 && u < 4 && has_next) stage_h(m0n, u);
                }
                if (u != 3) {
                    read_half(0, tan, load_an, twn);
                    if (u < 4) mfma_half(1, acc3);
                    else mfma_half(1, acco[(u - 4) % 3]);
                } else {
                    mfma_half(1, acc3);
                    // h3 chunk c: + group bias, ReLU, bf16, into LDS in the A-operand layout of the next product.  The group bias was
                    // fetched ten phases ago: the first four waves' counted waits have long covered it.
                    if (!lead || (ABL && (p.ablate & 1))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (!(ABL && (p.ablate & 8)))
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 b = gbv[j];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const f32x4 v = acc3[i][j];
                            bf16x4 o = {f2bf(fmaxf(v[0] + b[0], 0.f)), f2bf(fmaxf(v[1] + b[1], 0.f)), f2bf(fmaxf(v[2] + b[2], 0.f)),
                                        f2bf(fmaxf(v[3] + b[3], 0.f))};
                            *reinterpret_cast<bf16x4*>(A3 + ((h_dst + i * 2048) ^ (j << 5))) = o;
                        }
                    }
                    if (c < 3) fetch_gb(m0, c + 1);
                    else if (has_next) fetch_gb(m0n, 0);
                    if (!(ABL && (p.ablate & 16))) block_barrier();
                    read_half(0, tan, load_an, twn);
                }
            };
            phase(std::integral_constant<int, 0>{}); phase(std::integral_constant<int, 1>{}); phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{}); phase(std::integral_constant<int, 4>{}); phase(std::integral_constant<int, 5>{});
            phase(std::integral_constant<int, 6>{}); phase(std::integral_constant<int, 7>{}); phase(std::integral_constant<int, 8>{});
            phase(std::integral_constant<int, 9>{});
        };
#pragma unroll 1
        for (int cc = 0; cc < 2; ++cc) {   // (rolled: 20 phases of code, not 40)
            chunk(std::integral_constant<int, 0>{}, 2 * cc);
            chunk(std::integral_constant<int, 2>{}, 2 * cc + 1);
        }

        // ---- group maximum of the finished tile: in-lane over the wave's two 16-row blocks, over the 16 row lanes by DPP, the
        // four 32-row partial maxima through LDS; waves 4-7 combine and store (the first four go on to issue).  A block owns whole
        // groups (Mg | 128): every output is stored exactly once.  The conv4 bias of a thread's (up to six) outputs is fetched by
        // inline asm BEFORE the reduction: a load at its use would put an L2 round trip between two row tiles.
        // Thread te of waves 4-7 owns output columns te and te + 256 (< 384) of every group in the tile.
        float b4r[2] = {0.f, 0.f};
        int te = tid - 256;
        asm volatile("" : "+v"(te));   // (opaque per row tile: otherwise the offsets are hoisted out of the tile loop and spilled)
        if (!lead && !(ABL && (p.ablate & (64 | 128)))) {
#pragma unroll
            for (int q = 0; q < 2; ++q)   // unconditional (second column wrapped into the table) and early-clobber: the result register
                                          // is written when the load returns -- it must not double as an address or pass through a copy
            {
                unsigned long long sb;   // (SALU copy of the base: see dma())
                asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dword %0, %2, %1"
                             : "=&v"(b4r[q]), "=&s"(sb) : "v"((unsigned)((te + q * 256) % 384) * 4u), "s"(p.b4) : "memory");
            }
        }
        if (!(ABL && (p.ablate & 64)))
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float red[16];   // the lane's 4 x 4 columns of this output chunk, maximum over its two 16-row blocks
#pragma unroll
            for (int q = 0; q < 16; ++q) red[q] = fmaxf(acco[o][0][q >> 2][q & 3], acco[o][1][q >> 2][q & 3]);
            row16_max_batch(red);
            if ((lane & 15) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(s_part + wr * 384 + o * 128 + wc * 64 + j * 16 + (lane >> 4) * 4) =
                        f32x4{red[4 * j], red[4 * j + 1], red[4 * j + 2], red[4 * j + 3]};
            }
        }
        if (!lead) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias values (and nothing else of these waves)
        block_barrier();
        if (!lead && !(ABL && (p.ablate & (64 | 256)))) {
            const int Mg = p.Mg;                           // 32, 64 or 128: one, two or four 32-row blocks per group
            float* out = p.tok + (size_t)(m0 / Mg) * 384;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int n = te + q * 256;
                if (n < 384) {
                    const float v0 = s_part[n], v1 = s_part[384 + n], v2 = s_part[768 + n], v3 = s_part[1152 + n], b = b4r[q];
                    if (Mg == 128) out[n] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) + b;      // (a tile starts inside the matrix)
                    else if (Mg == 64) {
                        out[n] = fmaxf(v0, v1) + b;
                        if (m0 + 64 < p.M) out[384 + n] = fmaxf(v2, v3) + b;
                    } else {
                        out[n] = v0 + b;
                        if (m0 + 32 < p.M) out[384 + n] = v1 + b;
                        if (m0 + 64 < p.M) out[768 + n] = v2 + b;
                        if (m0 + 96 < p.M) out[1152 + n] = v3 + b;
                    }
                }
            }
        }
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
        bool ok = hipFuncSetAttribute((const void*)encoder_tail_persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess;
#ifdef CMDIAD_AB_VARIANTS
        ok = ok && hipFuncSetAttribute((const void*)encoder_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess &&
             hipFuncSetAttribute((const void*)encoder_tail_pp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess &&
             hipFuncSetAttribute((const void*)encoder_tail_pp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess &&
             hipFuncSetAttribute((const void*)encoder_tail_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess;
#endif
        if (!ok) {
            cmdiad_set_error("cmdiad_encoder_tail: hipFuncSetAttribute failed");
            return CMDIAD_ERR_LAUNCH;
        }
        attr = true;
    }
    const int M = groups * Mg;
    hipStream_t s = (hipStream_t)stream;
    GlobalTile H2{(const bf16_t*)h2, 256, M}, W3{(const bf16_t*)W3b, 256, 512}, W4t{(const bf16_t*)W4, 512, 384};
    TailParams p{M, Mg, gb, b4, tok_out, 0};
    const int n_tiles = (M + TM - 1) / TM;
    const dim3 grid_p((unsigned)(n_tiles < kTailCUs ? n_tiles : kTailCUs)), block(TW * 64);
#ifdef CMDIAD_AB_VARIANTS
    // test-only build: CMDIAD_TAIL_PP=0 the lock-step kernel, =1 the two-group kernel with one block per row tile (A/B runs,
    // identity test); CMDIAD_TAIL_ABLATE=bits timing ablations of the persistent (or, with CMDIAD_TAIL_PP=1, that) kernel
    const char* e = getenv("CMDIAD_TAIL_PP");
    const char* ea = getenv("CMDIAD_TAIL_ABLATE");
    p.ablate = ea ? atoi(ea) : 0;
    if (e && e[0] == '0') hipLaunchKernelGGL(encoder_tail_kernel, dim3(n_tiles), block, TAIL_LDS, s, H2, W3, W4t, p);
    else if (e && e[0] == '1') {
        if (p.ablate) hipLaunchKernelGGL(encoder_tail_pp_kernel<true>, dim3(n_tiles), block, TAIL_LDS, s, H2, W3, W4t, p);
        else hipLaunchKernelGGL(encoder_tail_pp_kernel<false>, dim3(n_tiles), block, TAIL_LDS, s, H2, W3, W4t, p);
    } else if (p.ablate) hipLaunchKernelGGL(encoder_tail_persist_kernel<true>, grid_p, block, TAIL_LDS, s, H2, W3, W4t, p, n_tiles);
    else
#endif
    hipLaunchKernelGGL(encoder_tail_persist_kernel<false>, grid_p, block, TAIL_LDS, s, H2, W3, W4t, p, n_tiles);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
