// 256 x 256 block tile computed by FOUR waves of 128 x 128 (one wave per SIMD, the whole 512-entry register file):
// 256 accumulator registers per lane in AGPRs + 96 fragment registers in VGPRs.  Against the 8-wave 256 x 256
// shape of gemm_core.h this halves the fragment ds_reads per FLOP (16 reads per 64 MFMAs) and removes the lock-step
// of eight waves that all read, then all multiply; the single wave hides its own LDS latency by requesting both
// 32-deep halves of a K-step before the first MFMA.
//
// Accumulators are NOT C++ values here.  hipcc cannot keep 256 loop-carried accumulator registers in place: with
// builtin MFMAs it shuttled them VGPR <-> AGPR <-> scratch around every K-step (212 v_accvgpr moves + 140 scratch
// loads per iteration), with "a"-class asm operands it spilled whole tiles, with physical-register operands it
// copied every accumulator out and back each iteration.  So accumulator tile T = i*8 + j lives in a[4T : 4T+3] BY
// CONVENTION: the MFMAs name those registers in the asm text, every asm statement lists all 256 AGPRs as clobbers
// (so the compiler never keeps a value of its own in an AGPR across one, and counts them in the kernel descriptor),
// and the epilogue reads them back with v_accvgpr_read asm.  The first K-step of an N tile uses a literal 0 as C.
// tools/isa_lint.py checks that the compiled kernel contains no compiler-generated AGPR traffic and no scratch.
#pragma once
#include <type_traits>
#include <utility>

#include "gemm_core.h"

namespace gemm {

// NJ = 8: 256 x 256 block (the shape described above).  NJ = 4: 256 x 128 block, 128 x 64 per wave (128 accumulator
// registers) for products whose N is not a multiple of 256 (the Point-MAE encoder's 384) -- still 256 rows per pass
// over the weights, i.e. 25 % fewer operand bytes through the L1 / LDS-DMA path per FLOP than 128 x 128 tiles.
template <int NJ_>
struct WideShape {
    static constexpr int BM = 256, NJ = NJ_, BN = 32 * NJ_, WAVES = 4, THREADS = 256, MI = 8;
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 2, LDS_BYTES = 2 * STAGE_BYTES;  // one block per CU
    static constexpr int WCOLS = 16 * NJ_;                                               // columns per wave
};
typedef WideShape<8> SWide;

#define CMDIAD_A8(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define CMDIAD_ALL_AGPRS                                                                                                   \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", CMDIAD_A8(1), CMDIAD_A8(2), CMDIAD_A8(3), CMDIAD_A8(4),     \
        CMDIAD_A8(5), CMDIAD_A8(6), CMDIAD_A8(7), CMDIAD_A8(8), CMDIAD_A8(9), CMDIAD_A8(10), CMDIAD_A8(11), CMDIAD_A8(12), \
        CMDIAD_A8(13), CMDIAD_A8(14), CMDIAD_A8(15), CMDIAD_A8(16), CMDIAD_A8(17), CMDIAD_A8(18), CMDIAD_A8(19),           \
        CMDIAD_A8(20), CMDIAD_A8(21), CMDIAD_A8(22), CMDIAD_A8(23), CMDIAD_A8(24), "a250", "a251", "a252", "a253", "a254", \
        "a255"

// acc tile T (+)= a . b   (16x16x32, A operand = a, B operand = b)
template <int T, bool ZERO>
__device__ __forceinline__ void wide_mfma(bf16x8 a, bf16x8 b)
{
    if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3) : CMDIAD_ALL_AGPRS);
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3) : CMDIAD_ALL_AGPRS);
}
template <int T, bool ZERO>
__device__ __forceinline__ void wide_mfma(f16x8 a, f16x8 b)
{
    if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3) : CMDIAD_ALL_AGPRS);
    else asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(4 * T), "n"(4 * T + 3) : CMDIAD_ALL_AGPRS);
}

template <int R>
__device__ __forceinline__ float wide_read()
{
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
    return x;
}

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one accumulator row block (NJ tiles) of one 32-deep half of a K-step
template <int I, bool SWAP, bool ZERO, class Frag, int NJ>
__device__ __forceinline__ void wide_row(const Frag& a, const Frag (&wf)[NJ])
{
    static_for<NJ>([&](auto J) {
        constexpr int T = I * NJ + decltype(J)::value;
        if constexpr (SWAP) wide_mfma<T, ZERO>(wf[decltype(J)::value], a);
        else wide_mfma<T, ZERO>(a, wf[decltype(J)::value]);
    });
}

// Same contract as gemm::run (flattened (n-tile, k-tile) pipeline, one barrier per K-step), except that the epilogue is
// called per accumulator ROW BLOCK: epi(std::integral_constant<int, I>, f32x4 (&row)[NJ], nt) for I = 0..7, where
// row[j][r] is the element the SWAP / non-SWAP layouts of gemm_core.h put at acc[I][j][r].
struct NoTileHook {
    __device__ __forceinline__ void operator()(int) const {}
};

// pre(nt) runs at the first K-step of every N tile, before that step's MFMAs (e.g. early fetches for the tile's epilogue)
template <bool SWAP, bool F16, int NJ = 8, class Epi, class Pre = NoTileHook>
__device__ __forceinline__ void run_wide(const GlobalTile& A, const GlobalTile& W, int m0, int nt0, int n_tiles, int KT, char* lds,
                                         Epi&& epi, Pre&& pre = Pre())
{
    using frag = typename std::conditional<F16, f16x8, bf16x8>::type;
    using S = WideShape<NJ>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int total = n_tiles * KT;
    int kt_s = 0, nt_s = nt0;
    // LDS-DMA staging.  Interior tiles (no row clamping needed -- all but the last M / N tile) use ONE per-lane 64-bit
    // pointer per operand plus wave-uniform offsets (K-step, 8-row piece), i.e. one 64-bit add per piece; GlobalTile's
    // generic stager (16 independent clamped addresses) costs ~8 VALU and a VGPR pair per piece, which at this register
    // budget spilled and put scratch reloads -- with their vmcnt(0) waits -- into the MFMA stream.
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int r_in = swave * 64 + (lane >> 3);                    // row of piece 0 inside the tile
    const int lchunk = (lane & 7) ^ ((lane >> 3) & 7);            // logical 16-byte chunk this lane fetches (XOR swizzle)
    const bool a_full = m0 + S::BM <= A.rows;
    const char* pa = reinterpret_cast<const char*>(A.base) + ((size_t)(m0 + r_in) * A.ld + lchunk * 8) * 2;
    auto dma_tile = [&](const char* p, int ld, int k0, char* tile, auto ROWS_PER_WAVE) {
        constexpr int RPW = decltype(ROWS_PER_WAVE)::value;
#pragma unroll
        for (int j = 0; j < RPW / 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)k0 * 2 + (size_t)j * 8 * ld * 2),
                                             (__attribute__((address_space(3))) void*)(tile + (swave * RPW + j * 8) * (BK * 2)), 16, 0, 0);
    };
    constexpr int WRPW = S::BN / 4;  // W rows staged per wave
    const int rw_in = swave * WRPW + (lane >> 3);
    auto stage_a = [&](int slot) {
        char* tile = lds + slot * S::STAGE_BYTES;
        if (a_full) dma_tile(pa, A.ld, kt_s * BK, tile, std::integral_constant<int, 64>{});
        else A.template stage<S::BM, S::WAVES>(tile, m0, kt_s * BK, tid);
    };
    auto stage_w = [&](int slot) {
        char* tile = lds + slot * S::STAGE_BYTES + S::BM * BK * 2;
        if (nt_s * S::BN + S::BN <= W.rows) {
            const char* pw = reinterpret_cast<const char*>(W.base) + ((size_t)(nt_s * S::BN + rw_in) * W.ld + lchunk * 8) * 2;
            dma_tile(pw, W.ld, kt_s * BK, tile, std::integral_constant<int, WRPW>{});
        } else W.template stage<S::BN, S::WAVES>(tile, nt_s * S::BN, kt_s * BK, tid);
        if (++kt_s == KT) { kt_s = 0; ++nt_s; }
    };
    stage_a(0);
    stage_w(0);
    wait_vmcnt<0>();
    block_barrier();

    int kt = 0, nt = nt0;
    for (int it = 0; it < total; ++it) {
        const char* ta = lds + (it & 1) * S::STAGE_BYTES;
        const char* tw = ta + S::BM * BK * 2;
        // Fragment schedule (96 VGPRs): both halves' W fragments and the first half's A fragments are requested up
        // front; the second half's A fragment i replaces the first half's as soon as row block i has been issued, so
        // those reads travel under the remaining MFMAs of the first half.
        frag af[8], wf[2][NJ];
        const int c0 = lane >> 4;
        auto a_frag = [&](int i, int chunk) { return *reinterpret_cast<const frag*>(ta + lds_off(wr * 128 + i * 16 + (lane & 15), chunk)); };
        auto w_frag = [&](int j, int chunk) { return *reinterpret_cast<const frag*>(tw + lds_off(wc * S::WCOLS + j * 16 + (lane & 15), chunk)); };
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[0][j] = w_frag(j, c0);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = a_frag(i, c0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[1][j] = w_frag(j, 4 + c0);
        // The next K-step's LDS-DMA (that stage was last read in step it-1, which every wave has left) is issued from
        // inside the first half, after row blocks 0 and 1: its address arithmetic then runs in the shadow of MFMAs that
        // are already executing instead of in front of the first one.
        const bool more = it + 1 < total;
        const int nslot = (it + 1) & 1;
        auto first_half = [&](auto ZERO) {
            static_for<8>([&](auto I) {
                constexpr int i = decltype(I)::value;
                wide_row<i, SWAP, decltype(ZERO)::value>(af[i], wf[0]);
                af[i] = a_frag(i, 4 + c0);
                if constexpr (i == 0) { if (more) stage_a(nslot); }
                if constexpr (i == 1) { if (more) stage_w(nslot); }
            });
        };
        if (kt == 0) {
            pre(nt);
            first_half(std::true_type{});
        } else first_half(std::false_type{});
        static_for<8>([&](auto I) { wide_row<decltype(I)::value, SWAP, false>(af[decltype(I)::value], wf[1]); });
        if (kt == KT - 1) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the compiler does not know the asm above are MFMAs
            static_for<8>([&](auto I) {
                f32x4 row[NJ];
                static_for<NJ>([&](auto J) {
                    constexpr int R = (decltype(I)::value * NJ + decltype(J)::value) * 4;
                    row[decltype(J)::value] = f32x4{wide_read<R>(), wide_read<R + 1>(), wide_read<R + 2>(), wide_read<R + 3>()};
                });
                epi(I, row, nt);
                // keep the row blocks apart: without this the scheduler hoists all 256 accumulator reads above the
                // first row's arithmetic (251 live VGPRs in the epilogue -> loop-carried values spilled in the K loop)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        wait_vmcnt<0>();
        block_barrier();
        if (++kt == KT) { kt = 0; ++nt; }
    }
}

}  // namespace gemm
