// Stream-K form of the residual products of the transformer blocks (models/models.py:126-132,177-180: x = x + fc2(...), and the
// same algebra in timm's ViT block reached at models.py:48):   out_f32 = A . W^T + bias + residual,   N % 256 == 0.
//
// Why.  The N = 768 products of ViT-B/8 are 99 x 3 = 297 tiles of 256 x 256: on 256 CUs a tile walk is two rounds for 1.16 rounds
// of work, which is why they ran on the 128 x 128 kernel (1 188 tiles, three rounds for 2.3).  Here the unit of work is a K-STEP:
// the (tile, k-tile) list -- 297 x 48 = 14 256 steps for fc2 -- is cut into one contiguous range per CU (55.7 steps each), so
// every CU does the same amount of MFMA work on the two-group 256 x 256 pipeline of gemm_pp3.h / l2min.hip.  A range covers the
// END of one tile, whole tiles, and the BEGINNING of another; a tile that two blocks share is finished IN ORDER:
//   * the block that owns the tile's first k-tiles accumulates from zero and parks its 256 x 256 fp32 accumulators in a
//     workspace slot (1 KiB per wave store: fully coalesced), then raises the slot's counter;
//   * the block that owns the rest loads them as the INITIAL accumulators and carries on -- the chain of MFMA accumulations per
//     output is exactly the one a single block would execute, so the result is bit-identical to the 128 x 128 kernel's
//     (tests/test_gpu_kernels.py), not a sum of two partial sums.
// Order inside a block: the head piece it must HAND OVER comes first, whole tiles next, the tail piece it must TAKE OVER last --
// its predecessor parked that piece at the very start of its own run, so nobody waits (ranges are at least one tile long: no
// block both takes and hands over the same tile).  Blocks use their hardware index (no XCD remap): dispatch is in index order,
// so a block's predecessor is resident before it is.
// Cross-XCD visibility: the parked accumulators are stored and loaded with sc0 sc1 (write-through / bypass: the L2 of an XCD is not
// coherent with the others'), the counters are agent-scope atomics; no cache-wide write-back or invalidate.
// The partial accumulators and the epilogue's residual rows arrive by inline-asm loads with explicit counted waits: a
// compiler-visible load inside the K loop gets an s_waitcnt vmcnt(0) in every iteration (tools/isa_lint.py).
#include <stdlib.h>

#include <mutex>

#include "gemm_pp3.h"

namespace {

using namespace gemm;

struct SkParams {
    int M, N, K;
    const float* bias;
    const float* residual; int ldr;
    float* out; int ldo;
    float* partial;        // [blocks][8 waves][32][64 lanes] f32x4
    unsigned* counter;     // [blocks][2]: waves that have parked slot b / waves that have taken it over
};

constexpr int kSkSlotFloats = 256 * 256;
constexpr int kSkEpiOps = 32;   // vector-memory stores of one wave at the end of a job (partial hand-over or full epilogue)

__global__ __launch_bounds__(512, 1) void gemm_sk_kernel(GlobalTile A, GlobalTile W, SkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using S = SPP3;
    using frag = bf16x8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int NT = p.N / S::BN, MT = (p.M + S::BM - 1) / S::BM, KT = p.K / BK;
    const long units = (long)MT * NT * KT;
    const int b = blockIdx.x, nb = gridDim.x;
    const int u0 = (int)(units * b / nb), u1 = (int)(units * (b + 1) / nb);
    const int T_total = u1 - u0;
    if (T_total <= 0) return;
    // the block's jobs, in execution order (see the header): [tile tB: k 0 .. kBe) | whole tiles tA + 1 .. tB - 1 | [tile tA: k kA .. KT)
    const int tA = u0 / KT, kA = u0 - tA * KT;
    const int tB = (u1 - 1) / KT, kBe = (u1 - 1) - tB * KT + 1;
    const int nj = tB - tA + 1;
    auto job_tile = [&](int j) __attribute__((always_inline)) { return nj == 1 ? tA : (j == 0 ? tB : (j == nj - 1 ? tA : tA + j)); };
    auto job_k0 = [&](int j) __attribute__((always_inline)) { return (nj == 1 || j == nj - 1) ? kA : 0; };
    auto job_kc = [&](int j) __attribute__((always_inline)) { return nj == 1 ? kBe - kA : (j == 0 ? kBe : (j == nj - 1 ? KT - kA : KT)); };

    RowStore32 rs;
    rs.init(lds + S::LDS_BYTES + wave * kRowStoreScratch, lane);
    // hand-over slots: wave w, accumulator register q = 4 i + j, lane l at float4 index (w * 32 + q) * 64 + l -- addressed as
    // (uniform base of (slot, i) in SGPRs) + (per-lane byte offset) + (j * 1 KiB immediate): no per-access address registers
    const unsigned slot_voff = (unsigned)((wave * 32 * 64 + lane) * 16);
    const float* slot_out = p.partial + (size_t)b * kSkSlotFloats;
    const float* slot_in = p.partial + (size_t)(b - 1) * kSkSlotFloats;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // a job that continues a tile: its first accumulators are what the previous block parked (inline asm: header comment)
    auto take_over = [&]() __attribute__((always_inline)) {
        if (lane == 0) {
            while (__hip_atomic_load(p.counter + 2 * (b - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) __builtin_amdgcn_s_sleep(2);
        }
        // the parked values were written by another CU, maybe on another XCD (whose L2 is not this one's): sc0 sc1 loads go to the
        // coherence point instead of this XCD's L2 -- an acquire FENCE at agent scope would invalidate the whole L2 under every
        // block of the XCD (measured: +60 us per launch)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* base = slot_in + (size_t)i * 4 * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc0 sc1" : "=v"(acc[i][j]) : "v"(slot_voff), "s"(base), "n"(j * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the slot is free again once all 8 waves have taken their part: the last one clears both counters for the next launch
        if (lane == 0) {
            const unsigned got = __hip_atomic_fetch_add(p.counter + 2 * (b - 1) + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got == 7u) {
                __hip_atomic_store(p.counter + 2 * (b - 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.counter + 2 * (b - 1) + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    // a job that stops before the tile's last k-tile: park the accumulators for the next block
    auto hand_over = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* base = slot_out + (size_t)i * 4 * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc0 sc1" ::"v"(slot_voff), "v"(acc[i][j]), "s"(base), "n"(j * 1024) : "memory");
        }
        // write-through stores (sc0 sc1), complete at the coherence point when the counter says so: this wave's count goes up only
        // after they are (a release FENCE at agent scope would write back the whole L2 instead)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(p.counter + 2 * b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // out = acc + bias + residual, rows through the per-wave LDS transposition (8 rows x 128 contiguous bytes per access);
    // residual rows of 16-row block i + 1 are in flight while block i is combined and stored.  Residual addresses: a uniform
    // base (the wave's 64 columns) in SGPRs + a 32-bit per-lane byte offset (row x pitch + 16 u; rows past M clamp to M - 1).
    auto epilogue = [&](int mt, int ntile) __attribute__((always_inline)) {
        const int nw = ntile * S::BN + wc * 64, mw = mt * S::BM + wr * 128;
        f32x4 bj[4];
        {
            const float* bp = p.bias + nw;
            const unsigned boff = (unsigned)((lane >> 4) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(bj[j]) : "v"(boff), "s"(bp), "n"(j * 64) : "memory");
        }
        const bool full = mt * S::BM + S::BM <= p.M;
        const float* rbase = p.residual + nw;
        f32x4 res[2][4];   // [parity of the 16-row block][ch * 2 + (row R | row R + 8)]
        auto load_res = [&](int i) __attribute__((always_inline)) {
            const int m0r = mw + i * 16 + rs.R;
            const unsigned oa = (unsigned)min(m0r, p.M - 1) * (unsigned)(p.ldr * 4) + (unsigned)(rs.u * 16);
            const unsigned ob = (unsigned)min(m0r + 8, p.M - 1) * (unsigned)(p.ldr * 4) + (unsigned)(rs.u * 16);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(res[i & 1][0]) : "v"(oa), "s"(rbase) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(res[i & 1][1]) : "v"(ob), "s"(rbase) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=v"(res[i & 1][2]) : "v"(oa), "s"(rbase) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=v"(res[i & 1][3]) : "v"(ob), "s"(rbase) : "memory");
        };
        load_res(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 7) load_res(i + 1);
            // operations issued after block i's loads: block i - 1's stores (4) and block i + 1's loads (4); a ragged tile skips
            // stores, so its count is not known: it waits for everything
            if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (i == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // (the bias loads are older still)
            else if (i < 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            const int m0r = mw + i * 16 + rs.R;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int col = nw + ch * 32 + rs.u * 4;
                rs.park(acc[i][2 * ch] + bj[2 * ch], acc[i][2 * ch + 1] + bj[2 * ch + 1]);
                f32x4 t0, t1;
                rs.fetch(t0, t1);
                const f32x4 o0 = t0 + res[i & 1][ch * 2], o1 = t1 + res[i & 1][ch * 2 + 1];
                if (full || m0r < p.M) *reinterpret_cast<f32x4*>(p.out + (size_t)m0r * p.ldo + col) = o0;
                if (full || m0r + 8 < p.M) *reinterpret_cast<f32x4*>(p.out + (size_t)(m0r + 8) * p.ldo + col) = o1;
            }
        }
        return full;
    };

    // Everything below is instantiated twice, once per stream: a wave only ever executes its own (lean) issue path.
    auto body = [&](auto BANK) {
    constexpr bool bank_wave = decltype(BANK)::value;
    const int sw = wave & 3;
    const int src_chunk = ((lane & 7) ^ (lane >> 3)) * 8;  // element offset of the 16-byte chunk this lane fetches
    const int row_w = bank_wave ? (sw >> 1) * 64 + (sw & 1) * 16 : sw * 16;  // this wave's share of every half-unit
    const size_t ld2 = (size_t)(bank_wave ? W.ld : A.ld) * 2;                  // row pitch in bytes
    int s_job = 0, s_left = job_kc(0), s_mt = job_tile(0) / NT, s_k = job_k0(0);   // the stream's current job
    auto tile_ptr = [&]() {
        const int t = job_tile(s_job);
        s_mt = t / NT;
        const int nt = t - s_mt * NT;
        return bank_wave ? reinterpret_cast<const char*>(W.base + (size_t)(nt * S::BN + row_w + (lane >> 3)) * W.ld + s_k * BK + src_chunk)
                         : reinterpret_cast<const char*>(A.base + (size_t)(s_mt * S::BM + row_w + (lane >> 3)) * A.ld + s_k * BK + src_chunk);
    };
    const char* ptr = tile_ptr();
    bool a_full = s_mt * S::BM + S::BM <= A.rows;
    int hT = 0;                    // stream cursor: K-tile index over the whole unit range
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
                else src = reinterpret_cast<const char*>(A.base + (size_t)min(s_mt * S::BM + row_w + rows + e * 8 + (lane >> 3), A.rows - 1) * A.ld + s_k * BK + src_chunk);
                dst = lds + S::A_OFF + (hi ? slot_hi : slot_lo) * S::HALF + (row_w + hsel * 64 + e * 8) * 128;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        if constexpr (part == 3) {  // next K-tile of this stream
            ++hT;
            ++s_k;
            if (--s_left == 0 && s_job + 1 < nj) {   // next job
                ++s_job;
                s_left = job_kc(s_job);
                s_k = job_k0(s_job);
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
    int ep_age = 1 << 20;  // phases since a job end that issued kSkEpiOps stores (wave-uniform)
    auto phase_wait = [&](bool issued) {
        constexpr int lead = bank_wave ? 7 : 3, base = bank_wave ? 14 : 6;
        constexpr int raised = base + kSkEpiOps > 63 ? 63 : base + kSkEpiOps;
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
    int c_job = 0, c_left = job_kc(0);
    if (job_k0(0) > 0) take_over();          // (only when the whole range lies inside one tile)
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
        if (--c_left == 0) {  // job finished
            __builtin_amdgcn_sched_barrier(0);
            const int t = job_tile(c_job);
            bool counted = true;
            if (job_k0(c_job) + job_kc(c_job) < KT) hand_over();
            else counted = epilogue(t / NT, t - (t / NT) * NT);
            ep_age = counted ? 0 : 1 << 20;
            ++c_job;
            if (c_job < nj) {
                c_left = job_kc(c_job);
                if (job_k0(c_job) > 0) take_over();
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pp3_barrier();
    }
    if (wr == 0) pp3_barrier();  // both groups execute the same number of barriers
    };
    if (wave < 4) body(std::true_type{});
    else body(std::false_type{});
}

bool aligned16(const void* q) { return ((uintptr_t)q & 15) == 0; }
constexpr int kSkBlocks = 256;   // one block per CU (MI355X)

}  // namespace

// The same kernel with ONE WHOLE TILE per block (grid = tiles: every block's unit range is exactly one tile, so nothing is parked
// or taken over and the workspace is never touched).  Stand-alone this loses to the 128 x 128 kernel (294 tiles = a full round of
// 256 CUs and a round of 38), but per CU-second a 256 x 256 tile is cheaper (half the LDS-DMA pieces per FLOP, the bound of the
// 128 x 128 kernel: profiles/r4_notes.md section 12) -- and inside the pipeline the CUs its second round leaves idle are not idle:
// the other streams' kernels take them.  Selected by cmdiad_gemm_bf16 (gemm.hip) for the residual products it is legal for.
int gemm_residual_tiles_launch(const cmdiad_gemm_args* a, hipStream_t stream)
{
    constexpr int kLds = SPP3::LDS_BYTES + 8 * kRowStoreScratch;
    static std::mutex mu;
    static bool attr = false;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr) {
            if (hipFuncSetAttribute((const void*)gemm_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) {
                cmdiad_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", kLds);
                return CMDIAD_ERR_LAUNCH;
            }
            attr = true;
        }
    }
    GlobalTile A{(const bf16_t*)a->A, a->lda, a->M}, W{(const bf16_t*)a->W, a->ldw, a->N};
    SkParams p{a->M, a->N, a->K, a->bias, a->residual, a->ldr, a->out_f32, a->ldo32, nullptr, nullptr};
    const long tiles = ((long)(a->M + 255) / 256) * (a->N / 256);
    hipLaunchKernelGGL(gemm_sk_kernel, dim3((unsigned)tiles), dim3(512), kLds, stream, A, W, p);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" size_t cmdiad_gemm_streamk_workspace_bytes(void)
{
    return (size_t)kSkBlocks * kSkSlotFloats * sizeof(float) + (size_t)kSkBlocks * 2 * sizeof(unsigned);
}

extern "C" int cmdiad_gemm_streamk_eligible(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || N % 256 != 0 || K % 64 != 0 || K < 192) return 0;
    const long tiles = ((long)(M + 255) / 256) * (N / 256), KT = K / 64;
    // every block's range must be at least one tile long (no block takes over and hands over the same tile), and the split
    // must be worth it: more than one tile and less than two per CU
    return tiles * KT / kSkBlocks >= KT && tiles > kSkBlocks && tiles < 2 * kSkBlocks ? 1 : 0;
}

extern "C" int cmdiad_gemm_streamk_bf16(const cmdiad_gemm_args* a, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(a && a->A && a->W && a->bias && a->residual && a->out_f32 && workspace, CMDIAD_ERR_ARG,
                   "cmdiad_gemm_streamk_bf16: needs A, W, bias, residual, out_f32 and a workspace");
    CMDIAD_REQUIRE(!a->out_bf16 && !a->group_bias && a->act == CMDIAD_ACT_NONE && !a->out_pre_bf16 && !a->dact_of && a->split_k <= 1 &&
                       !a->m_count && !a->row_scale && !a->ln_xb && !a->ln_part && !a->add2,
                   CMDIAD_ERR_ARG, "cmdiad_gemm_streamk_bf16: the residual form only (out_f32 = A.W^T + bias + residual)");
    CMDIAD_REQUIRE(cmdiad_gemm_streamk_eligible(a->M, a->N, a->K), CMDIAD_ERR_ARG,
                   "cmdiad_gemm_streamk_bf16: shape M=%d N=%d K=%d is not eligible (cmdiad_gemm_streamk_eligible)", a->M, a->N, a->K);
    CMDIAD_REQUIRE(a->lda % 8 == 0 && a->ldw % 8 == 0 && aligned16(a->A) && aligned16(a->W) && a->ldo32 % 4 == 0 && a->ldr % 4 == 0 &&
                       aligned16(a->out_f32) && aligned16(a->residual) && aligned16(a->bias) && aligned16(workspace),
                   CMDIAD_ERR_ARG, "cmdiad_gemm_streamk_bf16: 16-byte alignment, lda / ldw %% 8, ldo / ldr %% 4");
    CMDIAD_REQUIRE(workspace_bytes >= cmdiad_gemm_streamk_workspace_bytes(), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_gemm_streamk_bf16: workspace %zu < %zu bytes", workspace_bytes, cmdiad_gemm_streamk_workspace_bytes());
    constexpr int kLds = SPP3::LDS_BYTES + 8 * kRowStoreScratch;
    static std::mutex mu;
    static bool attr = false;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr) {
            if (hipFuncSetAttribute((const void*)gemm_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) {
                cmdiad_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", kLds);
                return CMDIAD_ERR_LAUNCH;
            }
            attr = true;
        }
    }
    GlobalTile A{(const bf16_t*)a->A, a->lda, a->M}, W{(const bf16_t*)a->W, a->ldw, a->N};
    float* partial = (float*)workspace;
    SkParams p{a->M, a->N, a->K, a->bias, a->residual, a->ldr, a->out_f32, a->ldo32, partial,
               (unsigned*)(partial + (size_t)kSkBlocks * kSkSlotFloats)};
    hipLaunchKernelGGL(gemm_sk_kernel, dim3(kSkBlocks), dim3(512), kLds, (hipStream_t)stream, A, W, p);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
