// Patch-library nearest-neighbour scoring kernels (reference feature_extractors/features.py:186-190
// calculate_dist = torch.cdist, :227 torch.min(dist, dim=1), :235-254 re-weighting scan).
//
// cmdiad_l2_min_keys: the Q x Nb x D distance contraction never leaves the chip.  Each block owns one query tile
// (256 queries on the production shape) and a contiguous range of bank tiles; per bank tile the 16-bit MFMA mainloop
// (fp16 or bf16 operands, fp32 accumulate; swapped orientation: a lane holds 4 bank rows for ONE query) starts from
// -(|q|^2 + |b|^2) / 2, so a finished accumulator IS -d2 / 2 and the epilogue has no arithmetic left: it tags every value
// with its place in the lane (RowMin below) and keeps a per-lane running (min, first index).  After the last tile
// the four 16-lane groups are merged by shuffles and every query gets ONE 64-bit atomicMin per block on its packed key
// (value bits << 32 | global row: integer order = (distance, row) order).  Blocks are dealt to the XCDs in groups of
// `qgroup` query tiles x `splits` bank ranges, so a streamed bank tile is shared through that XCD's L2.
// Shapes: l2_min_pp3_kernel (256 x 256, 8 waves in two groups, per-stream issuer waves) from Q >= 512; l2_min_kernel<S128>
// below that and for the last Nb % 256 bank rows.  A third formulation, the lock-step l2_min_kernel<S2x2>, is compiled into the
// test-only build alone (make ab, -DCMDIAD_AB_VARIANTS; CMDIAD_L2_TILE=2); all three return the same keys.  (Rounds 1-4 also kept
// a 4-wave 128 x 128-per-wave kernel, a two-buffer two-group kernel and a 32-MFMA-per-phase kernel there: measured in
// profiles/r1_notes.md .. r4_notes.md, removed in round 5 when the running minimum changed its definition.)
#include <stdlib.h>

#include "gemm_core.h"

namespace {

using namespace gemm;

struct L2Params {
    int Q, Nb, D;
    const float* q_sqnorm;
    const float* b_sqnorm;
    unsigned row_offset;
    unsigned long long* keys;
    unsigned long long* keys2;   // runner-up per query (RowMin), or null
    int nq_tiles, n_bank_tiles, splits, qgroup;
    unsigned* diag;    // test-only build: in-kernel stamps of one workgroup (l2_min_pp3_kernel<F16, true>), else null
    int diag_wg;
    const int* q_count;   // device-resident number of live query rows (<= Q), or null: cmdiad_l2_min_keys_counted
    const int* seg_counts;  // cmdiad_l2_min_keys_segments: live rows of each of n_seg query segments (device), or null
    int n_seg, seg_stride;  // segment w = query rows [w * seg_stride, w * seg_stride + min(seg_counts[w], seg_stride))
};

// Live query rows known only on the device (the compacted query set of cmdiad_rows_dedup_plan): the grid is sized for Q, every
// block reads the count once and the blocks of query tiles beyond it leave.
// Returns the number of workgroups that have work: the XCD remap must run over THAT count -- it hands every XCD a contiguous range
// of (query tile, library range) blocks, so a remap over the launched grid would leave the live query tiles to the first XCDs only.
// With segments (cmdiad_l2_min_keys_segments: the gathered query sets of W ranks, each compacted on its own rank, laid out at a
// fixed stride) the live query tiles of ALL segments form one tile list: p.nq_tiles = sum of ceil(count[w] / BM).
template <int BM>
__device__ __forceinline__ int live_rows(GlobalTile& A, L2Params& p)
{
    if (p.seg_counts) {
        int tiles = 0;
        for (int w = 0; w < p.n_seg; ++w) tiles += (min(max(__builtin_amdgcn_readfirstlane(p.seg_counts[w]), 0), p.seg_stride) + BM - 1) / BM;
        p.nq_tiles = tiles;
        const int qg = min(p.qgroup, max(tiles, 1));
        p.qgroup = qg;
        return (tiles + qg - 1) / qg * qg * p.splits;
    }
    if (!p.q_count) return gridDim.x;
    const int q = __builtin_amdgcn_readfirstlane(*p.q_count);
    p.Q = q;
    A.rows = q;
    p.nq_tiles = (q + BM - 1) / BM;
    const int qg = min(p.qgroup, max(p.nq_tiles, 1));   // the launcher's clamp of qgroup, for the live tile count
    p.qgroup = qg;
    return (p.nq_tiles + qg - 1) / qg * qg * p.splits;
}

// First query row of live tile `qt`.  Plain / counted launches: qt * BM.  Segments: the tile's segment is found by walking the
// (<= 64) counts again; p.Q / A.rows become the END of that segment's live rows, which is all the kernels use them for (the row
// clamp of a partial tile, the norm fetch and the final atomics).
template <int BM>
__device__ __forceinline__ int tile_first_row(GlobalTile& A, L2Params& p, int qt)
{
    if (!p.seg_counts) return qt * BM;
    int before = 0, w = 0, cnt = 0;
    for (; w < p.n_seg; ++w) {
        cnt = min(max(__builtin_amdgcn_readfirstlane(p.seg_counts[w]), 0), p.seg_stride);
        const int t = (cnt + BM - 1) / BM;
        if (qt < before + t) break;
        before += t;
    }
    const int end = w * p.seg_stride + cnt;
    p.Q = end;
    A.rows = end;
    return w * p.seg_stride + (qt - before) * BM;
}


// ------------------------------------------------------------------------------------------------
// The running minimum of EVERY formulation (they must return identical keys: a row is searched by the 256 x 256 kernel in one
// launch and by the 128 x 128 kernel in another -- the ragged last rows of a shard, tests/test_gpu_fakeworld.py).
//   * A library tile's accumulators start at qh + bh, qh = -|q|^2 / 2, bh = -|b|^2 / 2 (one add per element where the zeroing
//     move stood), so the finished accumulator is a = q.b - (|q|^2 + |b|^2) / 2 = -d2 / 2: no arithmetic per element is left
//     (round 4: d2 = (|q|^2 + |b|^2) - 2 acc, compare, two selects = 5 operations per element, 8.5 % of the kernel).
//   * The low four mantissa bits of a are replaced by the element's place e = 4 j + r among the lane's 16 columns of the tile
//     (ascending library rows): one v_and_or_b32.  Tagged values compare as UNSIGNED integers: negative floats order by
//     magnitude (smaller |a| = smaller d2 first) and a positive a (d2 < 0: rounding at an exact match) is below every
//     negative one; the minimum of a row is a tree of v_min3_u32 (half an operation per element) and equal values resolve to
//     the lowest e.  What the truncation gives up is 2^-19 of the value, three orders below the operand rounding.
//   * Across tiles the candidates compare on the truncated value alone, strictly: of equal values the earlier tile stays.
// So the result is, per query, the row with the smallest TRUNCATED -d2 / 2 in unsigned order, lowest row first -- a
// definition that does not depend on the tile shape, the launch geometry or the shard a row lives in.
// ------------------------------------------------------------------------------------------------
constexpr unsigned kRowMinNone = 0xFFFFFFF0u;   // above every tagged value of a finite or infinite accumulator
constexpr unsigned kRowMinDead = 0xFF800000u;   // -inf (masked columns start there and stay): from here on "no candidate"

struct RowMin {
    unsigned u;    // truncated bits of the best accumulator so far
    unsigned u2;   // the runner-up: the best of the OTHER 16-row groups (see below)
    unsigned at;   // places of both: bits 15..0 = (library tile - the block's first tile) << 4 | e of the best, bits 31..16 the
                   // runner-up's (packed: the two-group kernel has no eight registers to spare; a block walks <= 4096 tiles)
};

// The runner-up (ABI v6, keys2): a lane's 16 columns of one library tile are the 16 rows {64 a + 16 j + 4 g + r: j, r = 0..3} of
// the library -- "group" (a, g) = (row >> 6, (row >> 2) & 3), the same rows in the 128-column and in the 256-column kernels --
// and the running state keeps the two smallest GROUP MINIMA (same order: truncated value, then lowest row), each with its place.
// So per query the search returns (1) the nearest row and (2) the nearest row outside the nearest row's group of 16: a definition
// in terms of library rows alone, independent of tile shape, launch geometry and shard (row_offset is a multiple of 64 wherever
// keys of different launches are merged).  The exact fp32 re-score of BOTH (cmdiad_l2_rescore2) then decides: a near-tie that the
// 16-bit operands resolve the wrong way is repaired unless the true nearest row shares its group with the 16-bit winner (15 of
// Nb - 1 rows) or a THIRD row lies inside the operand noise as well.  Cost: 5 selects per row and tile instead of 2.
__device__ __forceinline__ unsigned rowmin_tag(float a, unsigned e) { return (__float_as_uint(a) & ~15u) | e; }
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }

// one finished row (16 columns of one query in this lane = one group) against the running two best; tile: relative to the block's
// first library tile (< 4096)
__device__ __forceinline__ void rowmin_update(RowMin& best, const f32x4 (&row)[4], unsigned tile)
{
    unsigned t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[j * 4 + r] = rowmin_tag(row[j][r], (unsigned)(j * 4 + r));
    const unsigned m0 = umin3(t[0], t[1], t[2]), m1 = umin3(t[3], t[4], t[5]), m2 = umin3(t[6], t[7], t[8]);
    const unsigned m3 = umin3(t[9], t[10], t[11]), m4 = umin3(t[12], t[13], t[14]);
    const unsigned m = umin3(umin3(m0, m1, m2), umin3(m3, m4, t[15]), kRowMinNone);
    const unsigned mt = m & ~15u, mat = (m & 15u) | (tile << 4);
    const bool b1 = m < best.u;    // best.u / best.u2 are multiples of 16: m < u  <=>  (m & ~15) < u: of equal values the earlier
    const bool b2 = m < best.u2;   // tile (= the lower row: a lane walks its groups in ascending row order) stays
    best.u2 = b1 ? best.u : (b2 ? mt : best.u2);
    best.u = b1 ? mt : best.u;
    const unsigned shifted = (best.at << 16) | mat;             // new best: the old best becomes the runner-up
    const unsigned second = (best.at & 0xFFFFu) | (mat << 16);  // new runner-up only
    best.at = b1 ? shifted : (b2 ? second : best.at);
}

// column of the element inside the launch's library (the lane's columns of tile t start at t * BN + col0)
template <int BN>
__device__ __forceinline__ unsigned rowmin_col(unsigned at16, unsigned tile0, unsigned col0)
{
    const unsigned e = at16 & 15u;
    return (tile0 + ((at16 & 0xFFFFu) >> 4)) * BN + col0 + (e >> 2) * 16 + (e & 3u);
}

__device__ __forceinline__ unsigned long long rowmin_key(unsigned u, unsigned row)
{
    if (u >= kRowMinDead) return ~0ull;
    const float d2 = -2.0f * __uint_as_float(u);
    return pack_key(d2 > 0.0f ? d2 : 0.0f, row);
}

__device__ __forceinline__ unsigned long long umin64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a < b ? b : a; }

// merge the four 16-lane groups of a wave and publish the query row's best key -- and, with keys2, its runner-up:
//   old = atomicMin(best slot, k1) (returning); whichever of (old, k1) lost moves on to the runner-up slot together with k2.
// Whatever the order in which the blocks and waves of a query arrive, slot 1 ends as the smallest and slot 2 as the second
// smallest of everything published: a key enters slot 2 exactly when something smaller holds or takes slot 1, the second smallest
// of all is displaced from (or kept out of) slot 1 by the smallest alone, and (query, row) pairs are unique, so no key is ever
// compared with itself.  "No candidate" is ~0 here and ops.KEY_EMPTY (2^63 - 1) in the caller's arrays: both lose to every key.
__device__ __forceinline__ void rowmin_publish(unsigned long long k1, unsigned long long k2, int lane, bool live,
                                               unsigned long long* dst, unsigned long long* dst2)
{
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
        const unsigned long long o1 = shfl_xor_u64(k1, m), o2 = shfl_xor_u64(k2, m);
        k2 = umin64(umax64(k1, o1), umin64(k2, o2));
        k1 = umin64(k1, o1);
    }
    if (lane < 16 && live) {
        if (dst2) {
            const unsigned long long old = atomicMin(dst, k1);
            const unsigned long long c = umin64(umax64(old, k1), k2);
            if (c != ~0ull) atomicMin(dst2, c);
        } else atomicMin(dst, k1);
    }
}

template <class S, bool F16>
__global__ __launch_bounds__(S::THREADS, S::WAVES_PER_SIMD) void l2_min_kernel(GlobalTile A, GlobalTile W, L2Params p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int MI = S::MI;
    const int nwg = live_rows<S::BM>(A, p);
    if ((int)blockIdx.x >= nwg) return;
    const int wg = xcd_remap(blockIdx.x, nwg);
    // L2-aware 2-D arrangement: consecutive workgroup ids (= co-resident blocks of one XCD after the remap)
    // form groups of `qgroup` query tiles x `splits` bank ranges.  Per XCD the L2 then holds a handful of query
    // tiles (re-read every bank tile) while each streamed bank tile is shared by `qgroup` blocks -- with one
    // bank range per XCD the 32+ different query tiles (12 MB) thrash the 4 MB L2 and every K-step is fed
    // from the Infinity Cache instead.
    const int gsz = p.qgroup * p.splits;
    const int within = wg % gsz;
    const int split = within / p.qgroup, qt = (wg / gsz) * p.qgroup + within % p.qgroup;
    if (qt >= p.nq_tiles) return;
    const int per = (p.n_bank_tiles + p.splits - 1) / p.splits;
    const int nt0 = split * per;
    const int ntc = min(per, p.n_bank_tiles - nt0);
    if (ntc <= 0) return;
    const int m0 = tile_first_row<S::BM>(A, p, qt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / S::WN, wc = wave % S::WN;

    RowMin best[MI];
    float qh[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        best[i] = RowMin{kRowMinNone, kRowMinNone, 0u};
        const int m = m0 + wr * (MI * 16) + i * 16 + (lane & 15);
        qh[i] = m < p.Q ? -0.5f * p.q_sqnorm[m] : 0.0f;
    }

    run<S, true, F16>(A, W, m0, nt0, ntc, p.D / BK, lds, [&](auto& acc, int ntile, char*) {
#pragma unroll
        for (int i = 0; i < MI; ++i) rowmin_update(best[i], acc[i], (unsigned)(ntile - nt0));
    }, 0, [&](auto& acc, int ntile) {   // a tile's accumulators start at -(|q|^2 + |b|^2) / 2; columns past Nb at -inf
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ntile * S::BN + wc * 64 + j * 16 + (lane >> 4) * 4;
            float bh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bh[r] = n + r < p.Nb ? -0.5f * p.b_sqnorm[n + r] : -__builtin_inff();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = qh[i] + bh[r];
        }
    });

#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wr * (MI * 16) + i * 16 + (lane & 15);
        const unsigned col0 = (unsigned)(wc * 64 + (lane >> 4) * 4);
        rowmin_publish(rowmin_key(best[i].u, p.row_offset + rowmin_col<S::BN>(best[i].at, (unsigned)nt0, col0)),
                       rowmin_key(best[i].u2, p.row_offset + rowmin_col<S::BN>(best[i].at >> 16, (unsigned)nt0, col0)), lane, m < p.Q,
                       p.keys + m, p.keys2 ? p.keys2 + m : nullptr);
    }
}

__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// The two-group pipeline with MORE BANK BYTES IN FLIGHT (the ablations above put the remaining time in the latency of the
// streamed bank tiles against one K-tile of prefetch): three bank buffers (96 KiB) + three query HALF slots (48 KiB: a
// query tile's lo rows are read in phase 0 and its hi rows in phase 2, so halves rotate through three slots) = 144 KiB.
// The two operands are issued by DIFFERENT waves -- waves 0-3 (the earlier group) feed the bank stream, waves 4-7 the query
// stream -- because s_waitcnt vmcnt retires in order per wave: in one queue the short-lead query pieces would force the
// long-lead bank pieces out early.  Each stream is a plain sequence of half-units (8 pieces) [lo h0, lo h1, hi h0, hi h1] per
// K-tile, one per phase:   bank half-unit (P + 10) and query half-unit (P + 5) are issued in phase P.
//   bank:  lo(T') in phases 4T'-10, -9 (its buffer held tile T'-3, whose lo rows were last read in phase 4T'-12: an
//          earlier-group issuer needs two phases of distance), hi(T') in 4T'-8, -7 (last read 4T'-11); read in 4T', 4T'+1:
//          every half-unit has >= 7 phases, so vmcnt(14) (the 7 newest half-units) is the counted wait of the bank waves;
//   query: lo(T') in 4T'-5, -4 (slot of hi(T'-2), last read 4T'-6: a later-group issuer needs one phase), hi(T') in
//          4T'-3, -2 (slot of lo(T'-1), last read 4T'-4); the earlier group reads half a phase before the issuing group's
//          wait, so a half-unit issued in phase P is readable from P + 3: vmcnt(6) for the query waves.
// ------------------------------------------------------------------------------------------------
struct SPingPong3 {
    static constexpr int BM = 256, BN = 256, THREADS = 512;
    static constexpr int BUF = 32768, HALF = 16384;
    static constexpr int A_OFF = 3 * BUF, BN_OFF = A_OFF + 3 * HALF;
    static constexpr int LDS_BYTES = BN_OFF + 2 * 256 * 4;
};

// DIAG (test-only build): waves 0 and 4 of workgroup p.diag_wg stamp s_memtime at five points of every phase into LDS (no global
// traffic inside the loop: stores count in vmcnt and would shift the counted waits) and copy the stamps out at the end.
constexpr int kDiagStamps = 1280;   // per wave: 64 K-tiles x 4 phases x 5
template <bool F16, bool DIAG = false>
__global__ __launch_bounds__(512, 1) void l2_min_pp3_kernel(GlobalTile A, GlobalTile W, L2Params p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using S = SPingPong3;
    using frag = typename std::conditional<F16, f16x8, bf16x8>::type;
    const int nwg = live_rows<S::BM>(A, p);
    if ((int)blockIdx.x >= nwg) return;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int gsz = p.qgroup * p.splits;
    const int within = wg % gsz;
    const int split = within / p.qgroup, qt = (wg / gsz) * p.qgroup + within % p.qgroup;
    if (qt >= p.nq_tiles) return;
    const int per = (p.n_bank_tiles + p.splits - 1) / p.splits;
    const int nt0 = split * per;
    const int ntc = min(per, p.n_bank_tiles - nt0);
    if (ntc <= 0) return;
    const int m0 = tile_first_row<S::BM>(A, p, qt);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int KT = p.D / BK, T_total = ntc * KT;
    unsigned* diag_lds = reinterpret_cast<unsigned*>(lds + S::LDS_BYTES) + (wave >> 2) * kDiagStamps;
    const bool diag_on = DIAG && wg == p.diag_wg && (wave & 3) == 0;
    int diag_n = 0, diag_k = 0;
    unsigned long long diag_t[5] = {0, 0, 0, 0, 0};
    // s_memtime is a scalar-memory read (~100+ cycles, counted in lgkmcnt): the five stamps of a phase stay in SGPRs and are
    // written to LDS once per phase, after the MFMAs have been issued -- a wait per stamp would serialise the fragment reads
    auto stamp = [&]() {
        if constexpr (DIAG) {
            if (diag_on) {
                diag_t[diag_k] = __builtin_amdgcn_s_memtime();
                if (++diag_k == 5) {
                    diag_k = 0;
                    if (diag_n + 5 <= kDiagStamps) {
#pragma unroll
                        for (int e = 0; e < 5; ++e)
                            if (lane == 0) diag_lds[diag_n + e] = (unsigned)diag_t[e];
                    }
                    diag_n += 5;
                }
            }
        }
    };
    // LDS byte address of the [2][256] bank-norm area (the inline-asm accesses take raw LDS addresses)
    const unsigned bn_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + S::BN_OFF);

    RowMin best[8];
    float qh[8];   // -|q|^2 / 2 of the lane's eight query rows
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        best[i] = RowMin{kRowMinNone, kRowMinNone, 0u};
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        qh[i] = m < p.Q ? -0.5f * p.q_sqnorm[m] : 0.0f;
    }
    f32x4 acc[8][4];
    // The -|b|^2 / 2 of library tile t wait in LDS slot t & 1 ([2][256] floats): wave 0 fetches tile t + 1's norms in phase 0 of
    // tile t's first K-tile and parks them in phase 3; every wave takes its 16 columns' values when tile t is finished and
    // starts tile t + 1's accumulators from qh + bh (the first tile's are fetched in front of the prologue's DMA pieces).
    auto bh_fetch = [&](f32x4& bnv, int tile) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bnv) : "v"(p.b_sqnorm + (size_t)tile * S::BN + lane * 4) : "memory");
    };
    auto bh_park = [&](f32x4& bnv, int tile, auto VM) {   // VM: vector-memory operations issued after the fetch that may stay in flight
        constexpr int vm = decltype(VM)::value;
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bnv) : "n"(vm) : "memory");   // (in-out: the scaling below cannot move above the wait)
        const f32x4 h = bnv * -0.5f;
        const unsigned wa = bn_lds + (unsigned)((tile & 1) * 1024 + lane * 16);
        asm volatile("ds_write_b128 %0, %1" : : "v"(wa), "v"(h) : "memory");
    };
    auto acc_start = [&](int tile) {
        f32x4 b4[4];
        const unsigned ra = bn_lds + (unsigned)((tile & 1) * 1024 + (wc * 64 + (lane >> 4) * 4) * 4);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                     "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(b4[0]), "=&v"(b4[1]), "=&v"(b4[2]), "=&v"(b4[3]) : "v"(ra) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = qh[i] + b4[j][r];
    };

    // Everything below is instantiated twice, once per stream: a wave only ever executes its own (lean) issue path.
    auto body = [&](auto BANK) {
    constexpr bool bank_wave = decltype(BANK)::value;
    // ---- staging: waves 0-3 feed the BANK stream, waves 4-7 the QUERY stream; each stream is a sequence of half-units
    // (8 pieces = 4 waves x 2): [lo h0, lo h1, hi h0, hi h1] per K-tile, one half-unit per phase.  The issue path is kept to a
    // handful of instructions (it has to fit beside the other group's 16 MFMAs): one per-lane pointer per wave that walks the
    // K-tiles, compile-time row offsets per (part, piece), rotating slot counters instead of modulo arithmetic.
    const int sw = wave & 3;
    const int src_chunk = ((lane & 7) ^ (lane >> 3)) * 8;  // element offset of the 16-byte chunk this lane fetches (rule 21)
    const int row_w = bank_wave ? (sw >> 1) * 64 + (sw & 1) * 16 : sw * 16;  // this wave's share of every half-unit
    const size_t ld2 = (size_t)(bank_wave ? W.ld : A.ld) * 2;                  // row pitch in bytes
    // per-lane pointer to (first row of the current tile + row_w + lane / 8, k-tile column + chunk); query rows past Q clamp
    const bool a_full = m0 + S::BM <= A.rows;
    const char* ptr = bank_wave ? reinterpret_cast<const char*>(W.base + (size_t)(nt0 * S::BN + row_w + (lane >> 3)) * W.ld + src_chunk)
                                : reinterpret_cast<const char*>(A.base + (size_t)(m0 + row_w + (lane >> 3)) * A.ld + src_chunk);
    int hT = 0, hK = 0;           // stream cursor: tile index, k tile
    int slot_lo = 0, slot_hi = 1;  // bank: both = buffer of tile hT;  query: half slots of (lo, hi) of tile hT
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
                else src = reinterpret_cast<const char*>(A.base + (size_t)min(m0 + row_w + rows + e * 8 + (lane >> 3), A.rows - 1) * A.ld + hK * BK + src_chunk);
                dst = lds + S::A_OFF + (hi ? slot_hi : slot_lo) * S::HALF + (row_w + hsel * 64 + e * 8) * 128;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
        if constexpr (part == 3) {  // next K-tile of this stream
            ++hT;
            if (++hK == KT) {
                hK = 0;
                ptr += bank_wave ? (size_t)S::BN * ld2 - (size_t)(KT - 1) * BK * 2 : (size_t)0 - (size_t)(KT - 1) * BK * 2;
            } else ptr += BK * 2;
            if (bank_wave) { slot_lo = slot_lo == 2 ? 0 : slot_lo + 1; slot_hi = slot_lo; }
            else { slot_lo = slot_lo == 0 ? 2 : slot_lo - 1; slot_hi = slot_hi == 0 ? 2 : slot_hi - 1; }  // (x + 2) mod 3
        }
        return true;
    };
    // phase j issues bank part (j + 2) % 4 and query part (j + 1) % 4 (bank half-unit P + 10, query half-unit P + 5)
    auto issue_phase = [&](auto J) {
        constexpr int j = decltype(J)::value;
        return bank_wave ? issue_part(std::integral_constant<int, (j + 2) % 4>{}) : issue_part(std::integral_constant<int, (j + 1) % 4>{});
    };
    // counted wait of a phase: the bank stream keeps 7 half-units in flight, the query stream 3 (see the header comment)
    auto phase_wait = [&](bool issued) {
        if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (bank_wave) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    // prologue: bank half-units 0..9 (tiles 0, 1 and the lo unit of tile 2), query half-units 0..4 (tile 0 and lo h0 of tile 1);
    // wave 0's fetch of the first library tile's norms is older than its pieces and lands with them
    f32x4 bnv0;
    if (bank_wave && wave == 0) bh_fetch(bnv0, nt0);
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        issue_part(I0{}); issue_part(I1{}); issue_part(I2{}); issue_part(I3{}); issue_part(I0{});  // half-units 0..4
        if (bank_wave) { issue_part(I1{}); issue_part(I2{}); issue_part(I3{}); issue_part(I0{}); issue_part(I1{}); }  // 5..9
    }
    if (bank_wave && wave == 0) {
        bh_park(bnv0, nt0, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pp_barrier();
    acc_start(nt0);
    if (wr == 1) pp_barrier();  // the second group runs one barrier (half a phase) behind the first

    // ---- fragment addresses: row*128 + ((chunk ^ (row & 7)) << 4), chunk = kk*4 + (lane >> 4); kk = 1 flips bit 6.
    // Query rows of a half slot: wr*64 + i*16 + (lane & 15); bank rows of a buffer: wc*64 + j*16 + (lane & 15).
    const int swz = (((lane >> 4)) ^ (lane & 7)) << 4;
    const int a_off = (wr * 64 + (lane & 15)) * 128 + swz, b_off = (wc * 64 + (lane & 15)) * 128 + swz;
    int a_lo = 0, a_hi = 0, b_base = 0;
    auto lda = [&](int i, int kk) { return *reinterpret_cast<const frag*>(lds + (((i < 4 ? a_lo : a_hi) + (i & 3) * 2048) ^ (kk << 6))); };
    auto ldb = [&](int j, int kk) { return *reinterpret_cast<const frag*>(lds + ((b_base + j * 2048) ^ (kk << 6))); };

    frag af[4][2], wlo[2][2], whi[2][2];
    int nt_c = nt0, kt_c = 0;
    for (int T = 0; T < T_total; ++T) {
        a_lo = S::A_OFF + ((2 * T) % 3) * S::HALF + a_off;
        a_hi = S::A_OFF + ((2 * T + 1) % 3) * S::HALF + a_off;
        b_base = (T % 3) * S::BUF + b_off;
        // ================= phase 0: B lo + A lo
        stamp();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wlo[j][kk] = ldb(j, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[i][kk] = lda(i, kk);
        f32x4 bnv;  // wave 0: the next library tile's squared norms on their way to LDS
        const bool bn_fetch = bank_wave && wave == 0 && kt_c == 0 && nt_c + 1 < nt0 + ntc;  // wave-uniform
        if (bn_fetch) bh_fetch(bnv, nt_c + 1);
        { const bool is = issue_phase(std::integral_constant<int, 0>{}); stamp(); phase_wait(is); }
        stamp();
        pp_barrier();
        stamp();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(wlo[j][kk], af[i][kk], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        stamp();
        pp_barrier();
        // ================= phase 1: B hi
        stamp();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) whi[j][kk] = ldb(2 + j, kk);
        { const bool is = issue_phase(std::integral_constant<int, 1>{}); stamp(); phase_wait(is); }
        stamp();
        pp_barrier();
        stamp();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16(whi[j][kk], af[i][kk], acc[i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        stamp();
        pp_barrier();
        // ================= phase 2: A hi
        stamp();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[i][kk] = lda(4 + i, kk);
        { const bool is = issue_phase(std::integral_constant<int, 2>{}); stamp(); phase_wait(is); }
        stamp();
        pp_barrier();
        stamp();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16(whi[j][kk], af[i][kk], acc[4 + i][2 + j]);
        __builtin_amdgcn_s_setprio(0);
        stamp();
        pp_barrier();
        // ================= phase 3: no reads (B lo is still in registers)
        stamp();
        if (bn_fetch) bh_park(bnv, nt_c + 1, std::integral_constant<int, 6>{});  // 6 DMA pieces were issued after the fetch (phases 0-2)
        { const bool is = issue_phase(std::integral_constant<int, 3>{}); stamp(); phase_wait(is); }
        stamp();
        pp_barrier();
        stamp();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16(wlo[j][kk], af[i][kk], acc[4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        stamp();
        if (kt_c == KT - 1) {  // library tile finished: the accumulators are -d2 / 2 (RowMin); then the next tile's start values
#pragma unroll
            for (int i = 0; i < 8; ++i) rowmin_update(best[i], acc[i], (unsigned)(nt_c - nt0));
            if (T + 1 < T_total) acc_start(nt_c + 1);
        }
        pp_barrier();
        if (++kt_c == KT) { kt_c = 0; ++nt_c; }
    }
    if (wr == 0) pp_barrier();  // both groups execute the same number of barriers

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
        const unsigned col0 = (unsigned)(wc * 64 + (lane >> 4) * 4);
        rowmin_publish(rowmin_key(best[i].u, p.row_offset + rowmin_col<S::BN>(best[i].at, (unsigned)nt0, col0)),
                       rowmin_key(best[i].u2, p.row_offset + rowmin_col<S::BN>(best[i].at >> 16, (unsigned)nt0, col0)), lane, m < p.Q,
                       p.keys + m, p.keys2 ? p.keys2 + m : nullptr);
    }
    if constexpr (DIAG) {
        if (diag_on) {
            for (int e = lane; e < kDiagStamps; e += 64) p.diag[(wave >> 2) * kDiagStamps + e] = e < min(diag_n, kDiagStamps) ? diag_lds[e] : 0u;
        }
    }
    };
    if (wave < 4) body(std::true_type{});
    else body(std::false_type{});
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

template <class S, bool F16> struct L2Kernel { static constexpr auto fn = l2_min_kernel<S, F16>; };
template <bool F16> struct L2Kernel<SPingPong3, F16> { static constexpr auto fn = l2_min_pp3_kernel<F16>; };

// Optional device-side row counts of a launch: one live count for the whole query set (counted) or one per segment.
struct L2Live {
    const int* q_count = nullptr;
    const int* seg_counts = nullptr;
    int n_seg = 0, seg_stride = 0;
};

// Library ranges per query tile of the segments launch (the W-way row-sharded search: W x the query tiles of one rank against
// 1/W of the library).  Measured on the bagel library at W = 1 / 2 / 4 / 8 (299 / 150 / 74 / 38 library tiles, 213 live query
// tiles per segment; tools/l2_segments_sweep.sh, profiles/r4_notes.md): ranges of ~15 tiles as in the single-library launch, but
// never fewer than 8 ranges -- a group of 4 query tiles x 8 ranges is what one XCD's 32 CUs hold at a time, with fewer ranges its
// L2 has to keep more query tiles than it can (W = 8: 1 range 6.35 ms, 8 ranges 5.96 ms) -- and ranges of at least 4 tiles.
static int segment_splits(int nbt)
{
    int sp = nbt / 15;
    sp = sp < 8 ? 8 : (sp > 20 ? 20 : sp);
    const int most = nbt / 4 < 1 ? 1 : nbt / 4;
    return sp > most ? most : sp;
}

template <class S, bool F16>
int launch_l2(const uint16_t* q, const float* q_sqnorm, const uint16_t* bank, const float* bank_sqnorm, int Q, int Nb,
              int D, uint32_t row_offset, unsigned long long* keys, unsigned long long* keys2, hipStream_t stream,
              const L2Live& live = L2Live())
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)L2Kernel<S, F16>::fn, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS_BYTES) != hipSuccess) {
            cmdiad_set_error("cmdiad_l2_min_keys: hipFuncSetAttribute failed");
            return CMDIAD_ERR_LAUNCH;
        }
        attr = true;
    }
    // segments: the launch is sized for every segment's cap (seg_stride rows); Q = n_seg * seg_stride
    const int nq = live.seg_counts ? live.n_seg * ((live.seg_stride + S::BM - 1) / S::BM) : (Q + S::BM - 1) / S::BM;
    const int nbt = (Nb + S::BN - 1) / S::BN;
    // enough blocks to fill the chip a few times over, but long bank ranges per block so the running
    // min stays in registers and the per-block atomics stay negligible
    static const int env_splits = getenv("CMDIAD_L2_SPLITS") ? atoi(getenv("CMDIAD_L2_SPLITS")) : 0;
    static const int env_qgroup = getenv("CMDIAD_L2_QGROUP") ? atoi(getenv("CMDIAD_L2_QGROUP")) : 0;
    const char* env_seg_splits = getenv("CMDIAD_L2_SEG_SPLITS");   // read per call: tools/l2_segments.py sweeps it
    // measured on the bagel xyz library (profiles/r1_notes.md): 8 bank ranges for the 8-wave shapes; the 4-wave wide
    // shape gains another 5 % from 16-32 (shorter ranges, better tail balance), as long as a range keeps >= 4 tiles.
    // Round 3 (profiles/r3_notes.md, the two-group kernel on the de-duplicated 54 401 rows): 20 ranges of 15 tiles 5.40 ms against
    // 5.55 ms for 30 ranges of 10 -- a block's pipeline fill and drain are paid per range, and 213 live query tiles leave no
    // tail to balance; all 100 352 rows: within 1 % from 15 to 50 ranges.
    int splits = env_splits > 0 ? env_splits : 8;
    if (env_splits <= 0 && S::BM == 256 && !std::is_same<S, S2x2>::value) splits = nbt / 4 < 1 ? 1 : (nbt / 4 > 20 ? 20 : nbt / 4);
    if (live.seg_counts && S::BM == 256) splits = env_seg_splits && atoi(env_seg_splits) > 0 ? atoi(env_seg_splits) : segment_splits(nbt);
    splits = splits > nbt ? nbt : splits;
    if ((nbt + splits - 1) / splits > 4096) splits = (nbt + 4095) / 4096;   // RowMin keeps a tile's place in 12 bits
    int qgroup = env_qgroup > 0 ? env_qgroup : 4;
    qgroup = qgroup > nq ? nq : qgroup;
    GlobalTile A{(const bf16_t*)q, D, Q}, W{(const bf16_t*)bank, D, Nb};
    L2Params p{Q, Nb, D, q_sqnorm, bank_sqnorm, row_offset, keys, keys2, nq, nbt, splits, qgroup, nullptr, -1, live.q_count,
               live.seg_counts, live.n_seg, live.seg_stride};
    const int ngroups = (nq + qgroup - 1) / qgroup;
    hipLaunchKernelGGL((L2Kernel<S, F16>::fn), dim3(ngroups * qgroup * splits), dim3(S::THREADS), S::LDS_BYTES, stream, A, W, p);
    return CMDIAD_OK;
}

static int l2_min_keys_impl(const uint16_t* q, const float* q_sqnorm, const L2Live& q_count, const uint16_t* bank,
                           const float* bank_sqnorm, int Q, int Nb, int D, uint32_t row_offset,
                           unsigned long long* keys, unsigned long long* keys2, int dtype, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(Q >= 0 && Nb >= 0 && D > 0 && D % 64 == 0, CMDIAD_ERR_ARG, "cmdiad_l2_min_keys: need D%%64==0 (D=%d)", D);
    CMDIAD_REQUIRE(dtype == CMDIAD_DT_BF16 || dtype == CMDIAD_DT_F16, CMDIAD_ERR_ARG, "cmdiad_l2_min_keys: dtype");
    // nothing to search (an empty row shard: engine.Bank on a rank beyond the library's last row; a 0-row tensor has no address):
    // the keys stay what the caller put there
    if (Q == 0 || Nb == 0) return CMDIAD_OK;
    CMDIAD_REQUIRE(q && q_sqnorm && bank && bank_sqnorm && keys, CMDIAD_ERR_ARG, "cmdiad_l2_min_keys: null pointer");
    CMDIAD_REQUIRE(aligned16(q) && aligned16(bank), CMDIAD_ERR_ARG, "cmdiad_l2_min_keys: 16-byte alignment");
    // the runner-up is defined on groups of 16 library rows (row >> 6, (row >> 2) & 3): launches whose keys are merged must agree on them
    CMDIAD_REQUIRE(!keys2 || row_offset % 64 == 0, CMDIAD_ERR_ARG, "cmdiad_l2_min_keys: keys2 needs row_offset %% 64 == 0 (row_offset=%u)", row_offset);
    // production: the two-group 256 x 256 pipeline (l2_min_pp3_kernel) from Q >= 512, the 128 x 128 kernel below that, for the
    // last Nb % 256 library rows and for D < 192 (the two-group schedule assumes >= 3 K-tiles per library tile).
    // CMDIAD_L2_TILE (read per call: the parity tests force each shape on small inputs) = 0 / 5 for those two; the test-only
    // build (make ab) also knows 2 = the lock-step 256 x 256 shape of gemm::run.
    const char* env_tile = getenv("CMDIAD_L2_TILE");
    const int force = env_tile ? atoi(env_tile) : -1;
    int tile = force >= 0 ? force : (Q >= 512 ? 5 : 0);
    if (q_count.seg_counts && tile != 0) tile = 5;    // segments exist for the production shapes only
#ifndef CMDIAD_AB_VARIANTS
    if (tile == 2) {
        cmdiad_set_error("cmdiad_l2_min_keys: CMDIAD_L2_TILE=2 names an A/B variant that only the test build (make ab) contains");
        return CMDIAD_ERR_ARG;
    }
#endif
    if (tile != 0 && tile != 2 && tile != 5) {
        cmdiad_set_error("cmdiad_l2_min_keys: CMDIAD_L2_TILE=%d is not a distance-GEMM shape (0, 5; test build: 2)", tile);
        return CMDIAD_ERR_ARG;
    }
    if (tile == 5 && D < 192) tile = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == CMDIAD_DT_F16;
    int rc;
#define L2_ARGS q, q_sqnorm, bank, bank_sqnorm, Q, Nb, D, row_offset, keys, keys2, s, q_count
    if (tile == 5) {
        const int full = Nb / 256 * 256, rest = Nb - full;
        rc = CMDIAD_OK;
        if (full > 0) rc = h ? launch_l2<SPingPong3, true>(q, q_sqnorm, bank, bank_sqnorm, Q, full, D, row_offset, keys, keys2, s, q_count)
                             : launch_l2<SPingPong3, false>(q, q_sqnorm, bank, bank_sqnorm, Q, full, D, row_offset, keys, keys2, s, q_count);
        if (rc == CMDIAD_OK && rest > 0) {
            const uint16_t* b2 = bank + (size_t)full * D;
            rc = h ? launch_l2<S128, true>(q, q_sqnorm, b2, bank_sqnorm + full, Q, rest, D, row_offset + full, keys, keys2, s, q_count)
                   : launch_l2<S128, false>(q, q_sqnorm, b2, bank_sqnorm + full, Q, rest, D, row_offset + full, keys, keys2, s, q_count);
        }
    }
#ifdef CMDIAD_AB_VARIANTS
    else if (tile == 2) rc = h ? launch_l2<S2x2, true>(L2_ARGS) : launch_l2<S2x2, false>(L2_ARGS);
#endif
    else rc = h ? launch_l2<S128, true>(L2_ARGS) : launch_l2<S128, false>(L2_ARGS);
#undef L2_ARGS
    if (rc) return rc;
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_l2_min_keys(const uint16_t* q, const float* q_sqnorm, const uint16_t* bank,
                                  const float* bank_sqnorm, int Q, int Nb, int D, uint32_t row_offset,
                                  unsigned long long* keys, unsigned long long* keys2, int dtype, cmdiad_stream_t stream)
{
    return l2_min_keys_impl(q, q_sqnorm, L2Live(), bank, bank_sqnorm, Q, Nb, D, row_offset, keys, keys2, dtype, stream);
}

extern "C" int cmdiad_l2_min_keys_counted(const uint16_t* q, const float* q_sqnorm, const int* q_count, int Q_max,
                                          const uint16_t* bank, const float* bank_sqnorm, int Nb, int D,
                                          uint32_t row_offset, unsigned long long* keys, unsigned long long* keys2, int dtype,
                                          cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q_count, CMDIAD_ERR_ARG, "cmdiad_l2_min_keys_counted: null count");
    L2Live live;
    live.q_count = q_count;
    return l2_min_keys_impl(q, q_sqnorm, live, bank, bank_sqnorm, Q_max, Nb, D, row_offset, keys, keys2, dtype, stream);
}

extern "C" int cmdiad_l2_min_keys_segments(const uint16_t* q, const float* q_sqnorm, const int* seg_counts, int n_seg,
                                           int seg_stride, const uint16_t* bank, const float* bank_sqnorm, int Nb, int D,
                                           uint32_t row_offset, unsigned long long* keys, unsigned long long* keys2, int dtype,
                                           cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(seg_counts, CMDIAD_ERR_ARG, "cmdiad_l2_min_keys_segments: null counts");
    CMDIAD_REQUIRE(n_seg >= 1 && n_seg <= 64 && seg_stride >= 0 && (long long)n_seg * seg_stride < (1ll << 31), CMDIAD_ERR_ARG,
                   "cmdiad_l2_min_keys_segments: need 1 <= n_seg <= 64 (n_seg=%d) and n_seg * seg_stride < 2^31", n_seg);
    L2Live live;
    live.seg_counts = seg_counts;
    live.n_seg = n_seg;
    live.seg_stride = seg_stride;
    return l2_min_keys_impl(q, q_sqnorm, live, bank, bank_sqnorm, n_seg * seg_stride, Nb, D, row_offset, keys, keys2, dtype, stream);
}

#ifdef CMDIAD_AB_VARIANTS
// Test-only build: the production distance GEMM with in-kernel stamps of workgroup `wg` (after the XCD remap): waves 0 and 4
// write kDiagStamps 32-bit s_memtime values each to stamps[2][kDiagStamps] (tools/l2_stamps.py).  fp16 operands, Nb % 256 == 0.
extern "C" int cmdiad_l2_diag(const uint16_t* q, const float* q_sqnorm, const uint16_t* bank, const float* bank_sqnorm, int Q,
                              int Nb, int D, unsigned long long* keys, int wg, unsigned* stamps, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && q_sqnorm && bank && bank_sqnorm && keys && stamps && Nb % 256 == 0 && D % 64 == 0 && D >= 192, CMDIAD_ERR_ARG,
                   "cmdiad_l2_diag: bad args");
    using S = SPingPong3;
    const int lds_bytes = S::LDS_BYTES + 2 * kDiagStamps * (int)sizeof(unsigned);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)l2_min_pp3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
            cmdiad_set_error("cmdiad_l2_diag: hipFuncSetAttribute failed");
            return CMDIAD_ERR_LAUNCH;
        }
        attr = true;
    }
    const int nq = (Q + S::BM - 1) / S::BM, nbt = Nb / S::BN;
    int splits = nbt / 4 < 1 ? 1 : (nbt / 4 > 32 ? 32 : nbt / 4);
    int qgroup = 4 > nq ? nq : 4;
    GlobalTile A{(const bf16_t*)q, D, Q}, W{(const bf16_t*)bank, D, Nb};
    L2Params p{Q, Nb, D, q_sqnorm, bank_sqnorm, 0u, keys, nullptr, nq, nbt, splits, qgroup, stamps, wg, nullptr};
    const int ngroups = (nq + qgroup - 1) / qgroup;
    hipLaunchKernelGGL((l2_min_pp3_kernel<true, true>), dim3(ngroups * qgroup * splits), dim3(S::THREADS), lds_bytes, (hipStream_t)stream, A, W, p);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
#endif

