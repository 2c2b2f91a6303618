// k-nearest-neighbour grouping for gfx950: replaces knn_cuda.KNN(k=128, transpose_mode=True) and
// the gather / centre-subtract of Group.forward (reference models/models.py:86-113).
// Semantics = oracle/cmdiad_oracle.c:orc_knn_group, bit-for-bit: squared distance
// (dx*dx + dy*dy) + dz*dz in fp32 (compiled with -ffp-contract=off), the K smallest in ascending
// (d2, index) order.
//
// Design (HBM/L2-bound streaming select, no G x N distance matrix):
//   * one 256-thread workgroup owns CPB = 4 centres and streams the cloud ONCE for all four
//     (coalesced 12-byte point loads shared by the four distance evaluations);
//   * selection is a threshold filter: each centre keeps a 64-bit key (d2 bits << 32 | index)
//     threshold tau = its current K-th best; a point passes only if key < tau (rare after the
//     first chunks), passing keys are appended to a per-centre LDS buffer with ONE wave-aggregated
//     LDS atomic per wave (ballot + popcount), and when a buffer is half full it is pruned by an
//     in-LDS bitonic sort that also refreshes tau.  Expected appends per centre ~ K ln(N/K), so
//     steady-state cost is the streaming distance evaluation, not the selection.
//   * chunks are visited in a coprime-strided order (see the kernel) so the threshold converges fast
//     even though organised clouds arrive in raster order;
//   * the final sort leaves the K winners in ascending order; the epilogue gathers p[idx] - c.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr unsigned long long kInf = ~0ull;

#ifdef CMDIAD_AB_VARIANTS  // block-wide formulation (1024-key LDS bitonic sort): test-only build (make ab), A/B reference
#include "ab/knn_block.inc"
#endif  // CMDIAD_AB_VARIANTS


// ------------------------------------------------------------------------------------------------
// Second formulation: a WAVE owns its centres end to end -- no workgroup barriers, no LDS atomics, no 1024-key LDS sort.
// The block-wide kernel above spends most of its time in bitonic_sort_1024 (55 barrier-separated LDS passes per prune,
// ~18 prunes per block) and holds 32 KiB of LDS per block, which also keeps the ViT's GEMM blocks off the CU while it runs.
// Here: 4 waves per block (1 for small batches), 4 centres per wave (one stream of the cloud feeds four distance evaluations), per centre
//   * the running K best (K <= 128) live SORTED in registers: element e = r*64 + lane, r = 0, 1 (two 64-bit keys per lane);
//   * a passing key (key < tau, tau = current K-th best, a wave-uniform scalar) is appended to a 128-key LDS list at
//     cnt + popcount(earlier passing lanes) -- cnt is a scalar of the wave, no atomic;
//   * when more than 64 candidates wait, the list is sorted in registers (bitonic network over 2 x 64 keys: cross-lane
//     compare-exchange by shuffles, the distance-64 step inside the lane), reversed, min-merged against the running best
//     (the result is bitonic and holds the 128 smallest of both) and re-sorted with the 7-step bitonic merge.
// ~550 instructions per prune, ~12 prunes per centre.  Keys are unique ((d2, index) pairs), so the selected set and its
// ascending order are those of the oracle bit for bit, whatever the visiting order.
// ------------------------------------------------------------------------------------------------
constexpr int kStepPts = 128;     // points per streaming step (two per lane)
#ifndef CMDIAD_KNN_TRIG
#define CMDIAD_KNN_TRIG 112       // candidates waiting for a prune (see knn_wave_kernel)
#endif
#ifdef CMDIAD_KNN_SCALAR
constexpr bool kKnnScalar = true;    // timing-only build: one centre per arithmetic instruction (the form before round 4)
#else
constexpr bool kKnnScalar = false;
#endif
#ifdef CMDIAD_KNN_BRANCH_PER_CENTRE
constexpr bool kKnnBranchPerCentre = true;   // timing-only build: no common branch in front of the per-centre ones
#else
constexpr bool kKnnBranchPerCentre = false;
#endif

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// The partner's key at lane distance j, without the LDS: the sorting networks below are chains of ~70 dependent
// compare-exchanges per candidate batch, and a ds_bpermute round trip per 32-bit half of each was most of the kernel's time.
//   j = 1, 2:  DPP quad permutes;  j = 4 = 7 ^ 3: half-row mirror, then quad reverse;  j = 8 = 15 ^ 7: half-row mirror, then row mirror;
//   j = 16, 32: v_permlane16_swap / v_permlane32_swap of (v, v) leave {own, partner} in the two registers in a lane-dependent
//   order -- which is all a compare-exchange needs (min and max are symmetric).
__device__ __forceinline__ unsigned dpp_u32(unsigned v, int ctrl)   // ctrl: compile-time after unrolling
{
    switch (ctrl) {
    case 0xB1: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    case 0x4E: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    case 0x1B: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, false);
    case 0x141: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
    default: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
    }
}

// compare-exchange across lanes at distance j (< 64, compile-time after unrolling) for one register: ascending block if `up`
__device__ __forceinline__ unsigned long long cex_lane(unsigned long long v, int j, bool up, int lane)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    unsigned long long x, y;   // {own, partner} in some order
    if (j >= 16) {
        unsigned lo2 = lo, hi2 = hi;
        if (j == 16) {
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
        } else {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
        }
        x = ((unsigned long long)hi << 32) | lo;
        y = ((unsigned long long)hi2 << 32) | lo2;
    } else {
        unsigned olo, ohi;
        if (j == 1) { olo = dpp_u32(lo, 0xB1); ohi = dpp_u32(hi, 0xB1); }
        else if (j == 2) { olo = dpp_u32(lo, 0x4E); ohi = dpp_u32(hi, 0x4E); }
        else if (j == 4) { olo = dpp_u32(dpp_u32(lo, 0x141), 0x1B); ohi = dpp_u32(dpp_u32(hi, 0x141), 0x1B); }
        else { olo = dpp_u32(dpp_u32(lo, 0x141), 0x140); ohi = dpp_u32(dpp_u32(hi, 0x141), 0x140); }
        x = v;
        y = ((unsigned long long)ohi << 32) | olo;
    }
    const bool lower = (lane & j) == 0;            // this lane holds the lower-indexed element of the pair
    const bool take_min = lower == up;
    const unsigned long long mn = y < x ? y : x, mx = y < x ? x : y;
    return take_min ? mn : mx;
}

// ascending bitonic sort of 128 keys held as (a = elements 0..63, b = elements 64..127), element = r*64 + lane
__device__ __forceinline__ void sort128(unsigned long long& a, unsigned long long& b, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            // direction of the k-block containing element e: up when (e & k) == 0; register b is elements 64 + lane
            const bool up_a = (lane & k) == 0;
            const bool up_b = k == 64 ? false : ((lane & k) == 0);  // (64 + lane) & 64 != 0 when k == 64
            a = cex_lane(a, j, up_a, lane);
            b = cex_lane(b, j, up_b, lane);
        }
    }
    // k = 128: one ascending block; j = 64 is the in-lane pair (a, b), then j = 32 .. 1 across lanes
    {
        const unsigned long long mn = b < a ? b : a, mx = b < a ? a : b;
        a = mn; b = mx;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        a = cex_lane(a, j, true, lane);
        b = cex_lane(b, j, true, lane);
    }
}

// (ta, tb) sorted ascending, (ca, cb) sorted ascending -> (ta, tb) = the 128 smallest of the 256, sorted ascending
__device__ __forceinline__ void merge128(unsigned long long& ta, unsigned long long& tb, unsigned long long ca, unsigned long long cb, int lane)
{
    // reversed candidates: element e <- element 127 - e, i.e. register swap + lane mirror
    const unsigned long long ra = shfl_u64(cb, 63 - lane), rb = shfl_u64(ca, 63 - lane);
    ta = ra < ta ? ra : ta;   // bitonic sequence holding the 128 smallest
    tb = rb < tb ? rb : tb;
    {
        const unsigned long long mn = tb < ta ? tb : ta, mx = tb < ta ? ta : tb;
        ta = mn; tb = mx;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        ta = cex_lane(ta, j, true, lane);
        tb = cex_lane(tb, j, true, lane);
    }
}

// kWaves waves per block, kWaveCentres centres per wave: <4, 4> when that grid fills the chip, <1, 1> for small batches
// (a wave is a serial walk over the cloud: fewer centres per wave = shorter walk, more waves)
template <int kWaves, int kWaveCentres>
__global__ __launch_bounds__(kWaves * 64) void knn_wave_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ n_valid,
                                                              const float* __restrict__ center, int N, int G, int K,
                                                              int64_t* __restrict__ idx_out, float* __restrict__ neigh_out)
{
    // candidate list of a centre: a prune takes up to 128 of them (what its sorting network holds), so the list is let to grow to
    // kTrig before one is run -- kTrig - 1 + the 64 of one more half step entries at most.  (With 128 slots a prune had to run above
    // 64 entries: late in the stream, where a half step adds a few candidates, every prune then sorted ~65 keys in a 128-key
    // network, 12 prunes per centre instead of 9; the prunes are two thirds of this kernel.)  Measured, 32 x 24 576 points, same
    // box: 128 slots 0.795 ms; kTrig 80 / 96 / 112 / 128: 0.705 / 0.668 / 0.638 / 0.639 (the LDS of 128 costs a wave per SIMD).
    constexpr int kTrig = CMDIAD_KNN_TRIG, kCandCap = kTrig + 64;
    __shared__ unsigned long long s_cand[kWaves][kWaveCentres][kCandCap];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g0 = (blockIdx.x * kWaves + wave) * kWaveCentres;
    if (g0 >= G) return;
    const int n = n_valid ? n_valid[b] : N;
    const float* p = xyz + (size_t)b * N * 3;

    float cx[kWaveCentres], cy[kWaveCentres], cz[kWaveCentres];
    unsigned long long ta[kWaveCentres], tb[kWaveCentres], tau[kWaveCentres];
    unsigned tau_hi[kWaveCentres];   // high word of tau (the distance bits of the K-th best), wave-uniform
    int cnt[kWaveCentres];
#pragma unroll
    for (int c = 0; c < kWaveCentres; ++c) {
        const float* cc = center + ((size_t)b * G + min(g0 + c, G - 1)) * 3;
        cx[c] = cc[0]; cy[c] = cc[1]; cz[c] = cc[2];
        ta[c] = tb[c] = tau[c] = kInf;
        tau_hi[c] = 0xFFFFFFFFu;
        cnt[c] = 0;
    }
    auto prune = [&](int c) {
        unsigned long long* buf = s_cand[wave][c];
        const int take = min(cnt[c], 128), rest = cnt[c] - take;   // rest <= 63
        unsigned long long ca = lane < take ? buf[lane] : kInf;
        unsigned long long cb = lane + 64 < take ? buf[lane + 64] : kInf;
        sort128(ca, cb, lane);
        merge128(ta[c], tb[c], ca, cb, lane);
        const unsigned long long kth = K <= 64 ? ta[c] : tb[c];   // element K-1 = (r, lane) = ((K-1) >> 6, (K-1) & 63)
        tau[c] = shfl_u64(kth, (K - 1) & 63);
        tau_hi[c] = __builtin_amdgcn_readfirstlane((unsigned)(tau[c] >> 32));
        tau[c] = ((unsigned long long)tau_hi[c] << 32) | __builtin_amdgcn_readfirstlane((unsigned)tau[c]);
        // the entries beyond the 128 taken move to the front of the list -- those that still beat the new threshold
        cnt[c] = 0;
        if (rest > 0) {  // wave-uniform
            const unsigned long long k2 = lane < rest ? buf[128 + lane] : kInf;
            const bool keep = lane < rest && k2 < tau[c];
            const unsigned long long mk = __ballot(keep);
            if (keep) buf[__popcll(mk & ((1ull << lane) - 1ull))] = k2;
            cnt[c] = __popcll(mk);
        }
    };

    // scattered (coprime-strided) visiting order of the 128-point steps: an organised cloud arrives in raster order, and a
    // raster walk approaches every centre monotonically (almost every point would beat the running threshold)
    const int nsteps = (n + kStepPts - 1) / kStepPts;
    int sstride = (int)(0.6180339887f * (float)nsteps) | 1;
    for (;; sstride += 2) {
        int a = sstride, bb = nsteps;
        while (bb) { const int t = a % bb; a = bb; bb = t; }
        if (a == 1) break;
    }
    // unconditional loads of a clamped index, both halves of the step issued together (a load under a lane mask is followed by a full
    // wait of its own; lanes past the cloud's end are excluded by `inb` below)
    int sidx = 0;
    for (int step = 0; step < nsteps; ++step) {
        const int base = sidx * kStepPts;
        sidx += sstride;
        if (sidx >= nsteps) { sidx -= nsteps; if (sidx >= nsteps) sidx %= nsteps; }
        float hx[2], hy[2], hz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = min(base + h * 64 + lane, n - 1);
            hx[h] = p[kk * 3 + 0]; hy[h] = p[kk * 3 + 1]; hz[h] = p[kk * 3 + 2];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = base + h * 64 + lane;
            const bool inb = k < n;
            const float x = hx[h], y = hy[h], z = hz[h];
            // two centres per arithmetic instruction (v_pk_add_f32 / v_pk_mul_f32 are IEEE per element: the same single roundings
            // in the same order as the scalar form, so the keys do not change): 8 packed operations per pair instead of 16
            float dist[kWaveCentres];
            if constexpr (kWaveCentres >= 2 && !kKnnScalar) {
                const f32x2 X = {x, x}, Y = {y, y}, Z = {z, z};
#pragma unroll
                for (int pr = 0; pr < kWaveCentres / 2; ++pr) {
                    const f32x2 dx = X - f32x2{cx[2 * pr], cx[2 * pr + 1]}, dy = Y - f32x2{cy[2 * pr], cy[2 * pr + 1]},
                                dz = Z - f32x2{cz[2 * pr], cz[2 * pr + 1]};
                    const f32x2 d2 = (dx * dx + dy * dy) + dz * dz;
                    dist[2 * pr] = d2[0];
                    dist[2 * pr + 1] = d2[1];
                }
            } else {
#pragma unroll
                for (int c = 0; c < kWaveCentres; ++c) {
                    const float dx = x - cx[c], dy = y - cy[c], dz = z - cz[c];
                    dist[c] = (dx * dx + dy * dy) + dz * dz;
                }
            }
            // one wave-uniform branch per 64 points for ALL the wave's centres: once the thresholds have settled almost no point passes
            // any of them, and the per-centre key assembly + 64-bit compare + ballot + branch was most of the loop.  The filter in front
            // compares the distance bits alone (d2 >= 0: its bit pattern orders as an unsigned) with the high word of the threshold, which
            // is wave-uniform -- one 32-bit compare per centre, the lane masks united by scalar ORs; only a step that passes it builds
            // the keys and takes the centres one by one.
            unsigned long long coarse = 0ull;
#pragma unroll
            for (int c = 0; c < kWaveCentres; ++c) coarse |= __ballot(__float_as_uint(dist[c]) <= tau_hi[c]);
            if (!kKnnBranchPerCentre && (coarse & __ballot(inb)) == 0ull) continue;
#pragma unroll
            for (int c = 0; c < kWaveCentres; ++c) {
                const unsigned long long key = pack_key(dist[c], (unsigned)k);
                const bool pass = inb && key < tau[c];
                const unsigned long long m = __ballot(pass);
                if (m) {  // wave-uniform
                    if (pass) s_cand[wave][c][cnt[c] + __popcll(m & ((1ull << lane) - 1ull))] = key;
                    cnt[c] += __popcll(m);
                    if (cnt[c] >= kTrig) prune(c);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < kWaveCentres; ++c) {
        if (cnt[c] > 0) prune(c);   // (fewer than 128 are left: one prune takes them all)
        const int g = g0 + c;
        if (g >= G) continue;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = r * 64 + lane;
            if (k >= K) continue;
            const unsigned long long key = r == 0 ? ta[c] : tb[c];
            const int i = key == kInf ? 0 : (int)(key & 0xFFFFFFFFull);
            const size_t o = ((size_t)b * G + g) * K + k;
            if (idx_out) idx_out[o] = i;
            if (neigh_out) {
                neigh_out[o * 3 + 0] = p[i * 3 + 0] - cx[c];
                neigh_out[o * 3 + 1] = p[i * 3 + 1] - cy[c];
                neigh_out[o * 3 + 2] = p[i * 3 + 2] - cz[c];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Third formulation (round 6): the same selection on the points of a NEIGHBOURHOOD instead of the whole cloud -- exact.
// The streaming kernel above evaluates every centre against every point (1 024 x 24 576 per cloud) although the 128 nearest
// points of a centre are 0.5 % of an MVTec-3D cloud.  Here a cloud is first binned (knn_grid_build_kernel, one workgroup per
// cloud): the bounding box, the TWO axes of largest extent (a depth-camera cloud is a 2.5-D sheet), a 64 x 64 grid of square
// cells on them, a counting sort of the points by cell (x, y, z and the ORIGINAL index as 16 bytes; cells row-major, so a run
// of cells of one grid row is one contiguous run of points).  A wave then owns ONE centre (knn_grid_query_kernel) and scans the
// square rings of cells around it, innermost first, through the wave-level selection of the streaming kernel (same keys
// (d2 bits << 32 | original index), same d2 = (dx*dx + dy*dy) + dz*dz single roundings, same sorted-128 state and prune):
//   * every point NOT in a scanned cell differs from the centre by more than m cells along a grid axis (m = the scanned ring
//     radius), i.e. lies farther than m * h -- so once the K-th best squared distance is below ((m - 0.01) h)^2 the K best
//     of the scanned points are the K best of the cloud (0.01 cells of slack against the roundings of the cell index: 64 *
//     2^-23 = 8e-6 cells);
//   * otherwise the next radius is the one the current K-th best asks for (floor(sqrt(d2_K) / h) + 2), or twice the radius
//     while fewer than K points have been met; radius 64 is the whole cloud.
// Typically two rounds (radius 2, then 4): ~500 distance evaluations and 3-4 prunes per centre instead of 24 576 and 9.
// The selected set and its order are those of the streaming kernel and of the oracle bit for bit: keys are unique, and the
// final state is the K smallest keys of the cloud whatever the visiting order.
// ------------------------------------------------------------------------------------------------
constexpr int kGridSide = 64, kGridCells = kGridSide * kGridSide;
constexpr int kGridHdr = 8;   // floats per cloud: min on axis A, min on axis B, 1 / h, h, axis A, axis B, n, unused

__device__ __forceinline__ int grid_coord(float a, float mn, float inv_h)
{
    return (int)fminf(fmaxf((a - mn) * inv_h, 0.0f), (float)(kGridSide - 1));   // (NaN -> 0; monotone in a)
}

__device__ __forceinline__ float pick3(float x, float y, float z, int axis) { return axis == 0 ? x : (axis == 1 ? y : z); }

__global__ __launch_bounds__(1024) void knn_grid_build_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ n_valid, int N,
                                                              float4* __restrict__ sorted, int* __restrict__ cell_start,
                                                              float* __restrict__ hdr)
{
    __shared__ int s_cnt[kGridCells];
    __shared__ float s_red[16][6];
    __shared__ int s_wave[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = n_valid ? min(n_valid[b], N) : N;
    const float* p = xyz + (size_t)b * N * 3;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int k = tid; k < n; k += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[k * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], m, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], m, 64));
        }
        if (lane == 0) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
    }
    for (int c = tid; c < kGridCells; c += 1024) s_cnt[c] = 0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = s_red[0][a]; mx[a] = s_red[0][3 + a];
        for (int w = 1; w < 16; ++w) { mn[a] = fminf(mn[a], s_red[w][a]); mx[a] = fmaxf(mx[a], s_red[w][3 + a]); }
    }
    // the two axes of largest extent (ties: the lower axis); h = the larger extent / 64
    const float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
    int A, Bx;
    if (e0 >= e1 && e0 >= e2) { A = 0; Bx = e1 >= e2 ? 1 : 2; }
    else if (e1 >= e2) { A = 1; Bx = e0 >= e2 ? 0 : 2; }
    else { A = 2; Bx = e0 >= e1 ? 0 : 1; }
    if (A > Bx) { const int t = A; A = Bx; Bx = t; }
    const float ext = fmaxf(pick3(e0, e1, e2, A), pick3(e0, e1, e2, Bx));
    const float h = ext > 0.0f && ext < __builtin_inff() ? ext * (1.0f / kGridSide) : 0.0f;
    const float inv_h = h > 0.0f ? 1.0f / h : 0.0f;
    const float mnA = pick3(mn[0], mn[1], mn[2], A), mnB = pick3(mn[0], mn[1], mn[2], Bx);
    for (int k = tid; k < n; k += 1024) {
        const float x = p[k * 3], y = p[k * 3 + 1], z = p[k * 3 + 2];
        atomicAdd(&s_cnt[grid_coord(pick3(x, y, z, Bx), mnB, inv_h) * kGridSide + grid_coord(pick3(x, y, z, A), mnA, inv_h)], 1);
    }
    __syncthreads();
    // exclusive scan of the 4 096 counts: four cells per thread, wave scan, 16 wave totals
    int c4[4], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c4[i] = s_cnt[tid * 4 + i]; sum += c4[i]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    int run = base + incl - sum;
    int* cs = cell_start + (size_t)b * (kGridCells + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cs[tid * 4 + i] = run;
        s_cnt[tid * 4 + i] = run;     // now the cell's write cursor
        run += c4[i];
    }
    if (tid == 1023) cs[kGridCells] = run;
    if (tid == 0) {
        float* hd = hdr + (size_t)b * kGridHdr;
        hd[0] = mnA; hd[1] = mnB; hd[2] = inv_h; hd[3] = h; hd[4] = (float)A; hd[5] = (float)Bx; hd[6] = (float)n; hd[7] = 0.0f;
    }
    __syncthreads();
    float4* out = sorted + (size_t)b * N;
    for (int k = tid; k < n; k += 1024) {
        const float x = p[k * 3], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const int cell = grid_coord(pick3(x, y, z, Bx), mnB, inv_h) * kGridSide + grid_coord(pick3(x, y, z, A), mnA, inv_h);
        const int pos = atomicAdd(&s_cnt[cell], 1);
        out[pos] = float4{x, y, z, __int_as_float(k)};
    }
}

template <int kWaves>
__global__ __launch_bounds__(kWaves * 64) void knn_grid_query_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ n_valid,
                                                                    const float* __restrict__ center, const float4* __restrict__ sorted,
                                                                    const int* __restrict__ cell_start, const float* __restrict__ hdr,
                                                                    int N, int G, int K, int64_t* __restrict__ idx_out,
                                                                    float* __restrict__ neigh_out)
{
    constexpr int kTrig = CMDIAD_KNN_TRIG, kCandCap = kTrig + 64;
    __shared__ unsigned long long s_cand[kWaves][kCandCap];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x * kWaves + wave;
    if (g >= G) return;
    const float* p = xyz + (size_t)b * N * 3;
    const float4* sp = sorted + (size_t)b * N;
    const int* cs = cell_start + (size_t)b * (kGridCells + 1);
    const float* hd = hdr + (size_t)b * kGridHdr;
    const float mnA = hd[0], mnB = hd[1], inv_h = hd[2], h = hd[3];
    const int A = (int)hd[4], Bx = (int)hd[5];
    const float* cc = center + ((size_t)b * G + g) * 3;
    const float cx = cc[0], cy = cc[1], cz = cc[2];
    const int ia = __builtin_amdgcn_readfirstlane(grid_coord(pick3(cx, cy, cz, A), mnA, inv_h));
    const int ib = __builtin_amdgcn_readfirstlane(grid_coord(pick3(cx, cy, cz, Bx), mnB, inv_h));

    unsigned long long ta = kInf, tb = kInf, tau = kInf;
    int cnt = 0;
    unsigned long long* buf = s_cand[wave];
    auto prune = [&]() {
        const int take = min(cnt, 128), rest = cnt - take;   // rest <= 63
        unsigned long long ca = lane < take ? buf[lane] : kInf;
        unsigned long long cb = lane + 64 < take ? buf[lane + 64] : kInf;
        sort128(ca, cb, lane);
        merge128(ta, tb, ca, cb, lane);
        const unsigned long long kth = K <= 64 ? ta : tb;
        tau = shfl_u64(kth, (K - 1) & 63);
        tau = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(tau >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)tau);
        cnt = 0;
        if (rest > 0) {  // wave-uniform
            const unsigned long long k2 = lane < rest ? buf[128 + lane] : kInf;
            const bool keep = lane < rest && k2 < tau;
            const unsigned long long mk = __ballot(keep);
            if (keep) buf[__popcll(mk & ((1ull << lane) - 1ull))] = k2;
            cnt = __popcll(mk);
        }
    };
    // the points of the sorted array in [s, e): one contiguous run of cells of one grid row
    auto scan = [&](int s, int e) {
        for (int k0 = s; k0 < e; k0 += 64) {
            const int k = k0 + lane;
            const bool inb = k < e;
            const float4 q = sp[min(k, e - 1)];
            const float dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            const unsigned long long key = pack_key(d2, (unsigned)__float_as_int(q.w));
            const bool pass = inb && key < tau;
            const unsigned long long m = __ballot(pass);
            if (m) {  // wave-uniform
                if (pass) buf[cnt + __popcll(m & ((1ull << lane) - 1ull))] = key;
                cnt += __popcll(m);
                if (cnt >= kTrig) prune();
            }
        }
    };
    auto row_run = [&](int j, int lo, int hi) {   // cells [lo, hi] of grid row j (already clipped, lo <= hi)
        const int s = __builtin_amdgcn_readfirstlane(cs[j * kGridSide + lo]);
        const int e = __builtin_amdgcn_readfirstlane(cs[j * kGridSide + hi + 1]);
        if (e > s) scan(s, e);
    };
    int m_done = -1;          // rings 0 .. m_done have been scanned
    int m = h > 0.0f ? 2 : kGridSide;   // (a degenerate grid -- all points in one cell -- is scanned whole)
    for (;;) {
        m = min(m, kGridSide);
        for (int dj = -m; dj <= m; ++dj) {
            const int j = ib + dj;
            if (j < 0 || j >= kGridSide) continue;
            const int lo = max(ia - m, 0), hi = min(ia + m, kGridSide - 1);
            if (dj < -m_done || dj > m_done || m_done < 0) {
                row_run(j, lo, hi);                       // a row outside the scanned square: all of it
            } else {                                       // a row that crosses the scanned square: the two ends
                if (ia - m_done - 1 >= lo) row_run(j, lo, ia - m_done - 1);
                if (ia + m_done + 1 <= hi) row_run(j, ia + m_done + 1, hi);
            }
        }
        if (cnt > 0) prune();
        m_done = m;
        if (m >= kGridSide) break;
        const unsigned tau_bits = (unsigned)(tau >> 32);
        if (tau != kInf && tau_bits < 0x7F800000u) {       // K points met, with a finite K-th distance
            const float d2k = __uint_as_float(tau_bits);
            const float r = ((float)m - 0.01f) * h;
            if (d2k < r * r) break;                        // certified: nothing outside the scanned square can be nearer
            m = max(m + 1, (int)(sqrtf(d2k) * inv_h) + 2);
        } else m *= 2;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = r * 64 + lane;
        if (k >= K) continue;
        const unsigned long long key = r == 0 ? ta : tb;
        const int i = key == kInf ? 0 : (int)(key & 0xFFFFFFFFull);
        const size_t o = ((size_t)b * G + g) * K + k;
        if (idx_out) idx_out[o] = i;
        if (neigh_out) {
            neigh_out[o * 3 + 0] = p[i * 3 + 0] - cx;
            neigh_out[o * 3 + 1] = p[i * 3 + 1] - cy;
            neigh_out[o * 3 + 2] = p[i * 3 + 2] - cz;
        }
    }
}

}  // namespace

extern "C" int cmdiad_knn_group(const float* xyz, const int32_t* n_valid, const float* center, int B, int N,
                                int G, int K, int64_t* idx_out, float* neigh_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(xyz && center, CMDIAD_ERR_ARG, "cmdiad_knn_group: null pointer");
    CMDIAD_REQUIRE(K <= N, CMDIAD_ERR_ARG, "cmdiad_knn_group: K = %d neighbours requested from clouds of N = %d points", K, N);
    CMDIAD_REQUIRE(B >= 0 && N > 0 && G >= 0 && K > 0 && K <= 128, CMDIAD_ERR_ARG,
                   "cmdiad_knn_group: bad sizes B=%d N=%d G=%d K=%d (K<=128)", B, N, G, K);
    if (B == 0 || G == 0) return CMDIAD_OK;
#ifdef CMDIAD_AB_VARIANTS
    // test-only build: CMDIAD_KNN_WAVE=0 selects the block-wide formulation (A/B runs and the parity tests; read per call)
    const char* e = getenv("CMDIAD_KNN_WAVE");
    const bool wave_form = !(e && e[0] == '0');
#else
    const bool wave_form = true;
#endif
    if (wave_form) {
        if ((long)B * ((G + 15) / 16) >= 512) {
            hipLaunchKernelGGL((knn_wave_kernel<4, 4>), dim3((G + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, xyz, n_valid, center, N, G, K,
                               idx_out, neigh_out);
        } else if ((long)B * ((G + 3) / 4) >= 2048) {
            hipLaunchKernelGGL((knn_wave_kernel<1, 4>), dim3((G + 3) / 4, B), dim3(64), 0, (hipStream_t)stream, xyz, n_valid, center, N, G, K,
                               idx_out, neigh_out);
        } else {
            hipLaunchKernelGGL((knn_wave_kernel<1, 1>), dim3(G, B), dim3(64), 0, (hipStream_t)stream, xyz, n_valid, center, N, G, K, idx_out,
                               neigh_out);
        }
    }
#ifdef CMDIAD_AB_VARIANTS
    else {
        dim3 grid((G + kCPB - 1) / kCPB, B);
        hipLaunchKernelGGL(knn_group_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, xyz, n_valid, center, N, G,
                           K, idx_out, neigh_out);
    }
#endif
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

// The same grouping through the neighbourhood search (knn_grid_*_kernel): needs a workspace for the binned cloud.  Falls back to
// the streaming kernel for small clouds (binning does not pay) and when CMDIAD_KNN_GRID=0 (A/B runs, parity tests; read per call).
extern "C" size_t cmdiad_knn_workspace_bytes(int B, int N)
{
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * ((size_t)N * sizeof(float4) + (size_t)(kGridCells + 1) * sizeof(int) + (size_t)kGridHdr * sizeof(float)) + 64;
}

extern "C" int cmdiad_knn_group_ws(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int G, int K,
                                   int64_t* idx_out, float* neigh_out, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    const char* e = getenv("CMDIAD_KNN_GRID");
    const bool grid = !(e && e[0] == '0') && N >= 2048 && G > 0 && B > 0 && K > 0 && K <= 128 && K <= N;
    if (!grid) return cmdiad_knn_group(xyz, n_valid, center, B, N, G, K, idx_out, neigh_out, stream);
    CMDIAD_REQUIRE(xyz && center, CMDIAD_ERR_ARG, "cmdiad_knn_group_ws: null pointer");
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_knn_workspace_bytes(B, N) && ((uintptr_t)workspace & 15) == 0, CMDIAD_ERR_WORKSPACE,
                   "cmdiad_knn_group_ws: workspace too small or not 16-byte aligned");
    float4* sorted = (float4*)workspace;
    int* cell_start = (int*)(sorted + (size_t)B * N);
    float* hdr = (float*)(cell_start + (size_t)B * (kGridCells + 1));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(knn_grid_build_kernel, dim3(B), dim3(1024), 0, s, xyz, n_valid, N, sorted, cell_start, hdr);
    hipLaunchKernelGGL((knn_grid_query_kernel<4>), dim3((G + 3) / 4, B), dim3(256), 0, s, xyz, n_valid, center, (const float4*)sorted,
                       (const int*)cell_start, (const float*)hdr, N, G, K, idx_out, neigh_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
