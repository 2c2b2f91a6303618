// Two-group ("ping-pong") 256 x 256 x 64 MFMA pipeline as a PERSISTENT tile loop with a caller-supplied epilogue: the
// schedule of l2_min_pp3_kernel (l2min.hip, where the phase / staging / counted-wait design is derived and measured) walking
// a list of (M tile, N tile) jobs instead of one query tile against a range of library tiles.
//
//   * 8 waves of 128 x 64 in two groups of four (one wave of each group per SIMD) half a phase apart: one group's fragment
//     reads and LDS-DMA issue sit under the other group's MFMAs.  A K-tile = 4 phases of 16 MFMAs per wave.
//   * LDS: three W buffers (3 x 32 KiB) + three A half slots (3 x 16 KiB) = 144 KiB.  Waves 0-3 feed the W stream (seven
//     half-units ahead, counted wait vmcnt(14)), waves 4-7 the A stream (three ahead, vmcnt(6)); the s_waitcnt counters are
//     never drained inside a stream, and BOTH streams run on across tile boundaries: the next job's first K-tiles are
//     in flight while the current tile's last MFMAs and its epilogue execute -- no per-tile fill or drain, and with one block
//     per CU walking jobs [j0, j1) no partial last round.
//   * epi(acc, mt, nt) runs after a tile's last phase and returns the number of vector-memory operations it is GUARANTEED to
//     have issued (0 when unsure).  Loads and stores count in vmcnt like the DMA pieces and retire in order, so for the
//     phases in which the epilogue's operations are younger than the piece a counted wait protects (7 phases for the W
//     stream, 3 for the A stream) the wait's immediate is raised by that number: vmcnt(14) right after 32 stores would wait
//     for 20 of them to reach L2 -- with all eight waves of the CU at the next barrier.
//   * pre(mt, nt) runs in phase 0 of a tile's LAST K-tile: the place to fetch epilogue operands (bias) by inline asm; eight
//     DMA pieces are issued between that point and the epilogue (or the stream has ended and drained), so the epilogue opens
//     with s_waitcnt vmcnt(8) instead of a compiler-placed vmcnt(0) that would drain the prefetch queue.
// Whole 256-column tiles only (N % 256 == 0); ragged M is clamped on the A stream and masked by the epilogue; K >= 192.
#pragma once
#include "gemm_core.h"

namespace gemm {

struct SPP3 {
    static constexpr int BM = 256, BN = 256, THREADS = 512;
    static constexpr int BUF = 32768, HALF = 16384;
    static constexpr int A_OFF = 3 * BUF;
    static constexpr int LDS_BYTES = A_OFF + 3 * HALF;
};

__device__ __forceinline__ void pp3_barrier() { asm volatile("s_barrier" ::: "memory"); }

// acc[i][j][r]: row m = mt*256 + wr*128 + i*16 + (lane & 15), column n = nt*256 + wc*64 + j*16 + (lane >> 4)*4 + r
// (swapped orientation: a lane holds four consecutive columns of one row).
// Job order: M tiles in groups of GM; inside a group the M tile runs fastest, then the N tile (job -> (g, nt, m) with
// g = job / (rows(g) * NT)); GM = 1 is plain N-fastest.  A block's consecutive jobs then share the W tile (kept hot in
// L2) while the A tiles change; the blocks of an XCD (consecutive job ranges) work on few M groups at a time.
struct JobCursor {
    int g, nt, mi, rows, GM, NT, MT;
    __device__ __forceinline__ void init(int job, int GM_, int NT_, int MT_)
    {
        GM = GM_; NT = NT_; MT = MT_;
        const int per = GM * NT;             // jobs of a full group
        g = job / per;
        rows = min(GM, MT - g * GM);
        const int rem = job - g * per;       // (the last, short group is only ever entered at its first job or walked into)
        nt = rem / rows;
        mi = rem - nt * rows;
    }
    __device__ __forceinline__ int mt() const { return g * GM + mi; }
    __device__ __forceinline__ void next()
    {
        if (++mi == rows) {
            mi = 0;
            if (++nt == NT) { nt = 0; ++g; rows = min(GM, MT - g * GM); }
        }
    }
};

template <int N>
__device__ __forceinline__ void pp3_wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI_OPS: the operation count a FULL-tile epilogue returns (compile time: it becomes an s_waitcnt immediate)
template <bool F16, int EPI_OPS, class Pre, class Epi>
__device__ __forceinline__ void run_pp3_jobs(const GlobalTile& A, const GlobalTile& W, int j0, int j1, int NT, int KT, char* lds,
                                             Pre&& pre, Epi&& epi, int GM = 1, int MT = 1 << 30)
{
    using S = SPP3;
    using frag = typename std::conditional<F16, f16x8, bf16x8>::type;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int T_total = (j1 - j0) * KT;
    if (T_total <= 0) return;  // block-uniform

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Everything below is instantiated twice, once per stream: a wave only ever executes its own (lean) issue path.
    auto body = [&](auto BANK) {
    constexpr bool bank_wave = decltype(BANK)::value;
    const int sw = wave & 3;
    const int src_chunk = ((lane & 7) ^ (lane >> 3)) * 8;  // element offset of the 16-byte chunk this lane fetches
    const int row_w = bank_wave ? (sw >> 1) * 64 + (sw & 1) * 16 : sw * 16;  // this wave's share of every half-unit
    const size_t ld2 = (size_t)(bank_wave ? W.ld : A.ld) * 2;                  // row pitch in bytes
    JobCursor sj;                                                              // the stream's current job
    sj.init(j0, GM, NT, MT);
    int s_mt = sj.mt();
    auto tile_ptr = [&]() {
        return bank_wave ? reinterpret_cast<const char*>(W.base + (size_t)(sj.nt * S::BN + row_w + (lane >> 3)) * W.ld + src_chunk)
                         : reinterpret_cast<const char*>(A.base + (size_t)(s_mt * S::BM + row_w + (lane >> 3)) * A.ld + src_chunk);
    };
    const char* ptr = tile_ptr();
    bool a_full = s_mt * S::BM + S::BM <= A.rows;
    int hT = 0, hK = 0;            // stream cursor: K-tile index over the whole job range, k tile inside the job
    int slot_lo = 0, slot_hi = 1;  // W: both = buffer of K-tile hT;  A: half slots of (lo, hi) of K-tile hT
    if (bank_wave) slot_hi = 0;
    auto issue_part = [&](auto PART) {  // -> true when the half-unit was issued
        constexpr int part = decltype(PART)::value, hi = part >> 1, hsel = part & 1;
        if (hT >= T_total) return false;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const char* src;
            char* dst;
            if (bank_wave) {
                constexpr int rows = hsel * 128 + hi * 32;
                src = ptr + (size_t)(rows + e * 8) * ld2;
                dst = lds + slot_lo * S::BUF + (row_w + rows + e * 8) * 128;
            } else {
                constexpr int rows = hsel * 128 + hi * 64;
                if (a_full) src = ptr + (size_t)(rows + e * 8) * ld2;
                else src = reinterpret_cast<const char*>(A.base + (size_t)min(s_mt * S::BM + row_w + rows + e * 8 + (lane >> 3), A.rows - 1) * A.ld + hK * BK + src_chunk);
                dst = lds + S::A_OFF + (hi ? slot_hi : slot_lo) * S::HALF + (row_w + hsel * 64 + e * 8) * 128;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        if constexpr (part == 3) {  // next K-tile of this stream
            ++hT;
            if (++hK == KT) {       // next job
                hK = 0;
                sj.next();
                s_mt = sj.mt();
                ptr = tile_ptr();
                a_full = s_mt * S::BM + S::BM <= A.rows;
            } else ptr += BK * 2;
            if (bank_wave) { slot_lo = slot_lo == 2 ? 0 : slot_lo + 1; slot_hi = slot_lo; }
            else { slot_lo = slot_lo == 0 ? 2 : slot_lo - 1; slot_hi = slot_hi == 0 ? 2 : slot_hi - 1; }  // (x + 2) mod 3
        }
        return true;
    };
    // phase j issues W part (j + 2) % 4 and A part (j + 1) % 4 (W half-unit P + 10, A half-unit P + 5)
    auto issue_phase = [&](auto J) {
        constexpr int j = decltype(J)::value;
        return bank_wave ? issue_part(std::integral_constant<int, (j + 2) % 4>{}) : issue_part(std::integral_constant<int, (j + 1) % 4>{});
    };
    int ep_age = 1 << 20;  // phases since an epilogue that issued EPI_OPS operations (wave-uniform)
    auto phase_wait = [&](bool issued) {
        constexpr int lead = bank_wave ? 7 : 3, base = bank_wave ? 14 : 6;
        constexpr int raised = base + EPI_OPS > 63 ? 63 : base + EPI_OPS;
        if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (ep_age < lead) pp3_wait_vmcnt<raised>();
        else pp3_wait_vmcnt<base>();
        ++ep_age;
    };
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        issue_part(I0{}); issue_part(I1{}); issue_part(I2{}); issue_part(I3{}); issue_part(I0{});  // half-units 0..4
        if (bank_wave) { issue_part(I1{}); issue_part(I2{}); issue_part(I3{}); issue_part(I0{}); issue_part(I1{}); }  // 5..9
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp3_barrier();
    if (wr == 1) pp3_barrier();  // the second group runs one barrier (half a phase) behind the first

    // fragment addresses: row*128 + ((chunk ^ (row & 7)) << 4), chunk = kk*4 + (lane >> 4); kk = 1 flips bit 6
    const int swz = (((lane >> 4)) ^ (lane & 7)) << 4;
    const int a_off = (wr * 64 + (lane & 15)) * 128 + swz, b_off = (wc * 64 + (lane & 15)) * 128 + swz;
    int a_lo = 0, a_hi = 0, b_base = 0;
    auto lda = [&](int i, int kk) { return *reinterpret_cast<const frag*>(lds + (((i < 4 ? a_lo : a_hi) + (i & 3) * 2048) ^ (kk << 6))); };
    auto ldb = [&](int j, int kk) { return *reinterpret_cast<const frag*>(lds + ((b_base + j * 2048) ^ (kk << 6))); };

    frag af[4][2], wlo[2][2], whi[2][2];
    JobCursor cj;
    cj.init(j0, GM, NT, MT);
    int kt_c = 0;
    for (int T = 0; T < T_total; ++T) {
        a_lo = S::A_OFF + ((2 * T) % 3) * S::HALF + a_off;
        a_hi = S::A_OFF + ((2 * T + 1) % 3) * S::HALF + a_off;
        b_base = (T % 3) * S::BUF + b_off;
        // ================= phase 0: W lo + A lo
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wlo[j][kk] = ldb(j, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[i][kk] = lda(i, kk);
        if (kt_c == KT - 1) pre(cj.mt(), cj.nt);
        phase_wait(issue_phase(std::integral_constant<int, 0>{}));
        pp3_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(wlo[j][kk], af[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        pp3_barrier();
        // ================= phase 1: W hi
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) whi[j][kk] = ldb(2 + j, kk);
        phase_wait(issue_phase(std::integral_constant<int, 1>{}));
        pp3_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16(whi[j][kk], af[i][kk], acc[i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        pp3_barrier();
        // ================= phase 2: A hi
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[i][kk] = lda(4 + i, kk);
        phase_wait(issue_phase(std::integral_constant<int, 2>{}));
        pp3_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16(whi[j][kk], af[i][kk], acc[4 + i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        pp3_barrier();
        // ================= phase 3: no reads (W lo is still in registers)
        phase_wait(issue_phase(std::integral_constant<int, 3>{}));
        pp3_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16(wlo[j][kk], af[i][kk], acc[4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (kt_c == KT - 1) {  // tile finished
            __builtin_amdgcn_sched_barrier(0);
            const int ops = epi(acc, cj.mt(), cj.nt);
            ep_age = ops >= EPI_OPS ? 0 : 1 << 20;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
        }
        pp3_barrier();
        if (++kt_c == KT) {
            kt_c = 0;
            cj.next();
        }
    }
    if (wr == 0) pp3_barrier();  // both groups execute the same number of barriers
    };
    if (wave < 4) body(std::true_type{});
    else body(std::false_type{});
}

}  // namespace gemm
