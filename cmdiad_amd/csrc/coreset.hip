// Greedy k-centre coreset selection (reference feature_extractors/features.py:372-425,
// get_coreset_idx_randomp after the sparse random projection), SURVEY 8(f) row f1.
//
// Reference semantics restated (coreset_dtype 'FP16'): z is fp16 [n,d]; every iteration computes
// dist_i = || z_i - z_last ||_2 (difference rounded to fp16, squares accumulated in fp32, result
// rounded to fp16: torch's half-precision norm on the GPU [external: "parity unpinned", the reference
// hard-codes .to("cuda") and cannot run in this container]), min_d = min(min_d, dist), next =
// argmax(min_d) with the lowest index on ties.  min_d[next] = 0 (:419) is implied: the next round's
// distance of the selected row to itself is exactly 0.
//
// One launch per iteration, no host synchronisation: round r publishes its winner with a 64-bit
// atomicMax of (value bits << 32 | ~index) into best[r]; round r+1 reads best[r].  The scan is
// HBM-bound: n*d*2 bytes per round (511 MB for the xyz bank of 'bagel': twice the Infinity Cache).
//
// Layout: besides the row-major fp16 copy (the pivot row is fetched from it) the rows are kept TRANSPOSED in pairs of
// dimensions, zT[c][row] = (z[row][2c], z[row][2c+1]) as half2 -- a thread owns four consecutive rows and walks c, so every
// load instruction of a wave is one contiguous 1 KiB piece and a row's sum of squares lives in ONE lane: no cross-lane
// reduction per row (round 1's wave-per-row form spent a 6-step butterfly per 668 bytes and reached 2.3 TB/s).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;

__device__ __forceinline__ float sq_diff_acc(unsigned a, unsigned b, float acc)
{
    const h2_t df = __builtin_bit_cast(h2_t, a) - __builtin_bit_cast(h2_t, b);   // rounded to fp16 like z_lib - last_item (v_pk_add_f16)
    const float x = (float)df[0], y = (float)df[1];
    return acc + (x * x + y * y);
}

// EARLY: partial-distance termination, exact.  The squares are non-negative and accumulated in fp32, so a row's partial sum
// never decreases, and sqrt and the fp16 rounding are monotone: once fp16(sqrt(partial)) >= min_d[row], the full distance cannot be
// below min_d[row] either and the row's minimum stays as it is.  Every eight dimension pairs the wave votes; when all of its 256
// rows are settled it stops reading (the remaining 1 KiB pieces of its rows are never fetched).  Late in a selection most rows sit
// close to some centre and far from the newest one, so most waves leave after a fraction of the dimensions.
template <bool EARLY>
__global__ __launch_bounds__(256) void coreset_round_kernel(const __half* __restrict__ z, const uint4* __restrict__ zT, int n, int n4,
                                                            int d2, __half* __restrict__ min_d,
                                                            const unsigned long long* __restrict__ best_prev,
                                                            unsigned long long* __restrict__ best_cur, int first_idx, int q_begin, int q_end)
{
    // [q_begin, q_end): the groups of four rows this launch scans -- all of them, or one rank's row shard (cmdiad_coreset_round)
    __shared__ unsigned s_piv[512];
    __shared__ unsigned long long s_key[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int last = best_prev ? (int)(0xFFFFFFFFu - (unsigned)(*best_prev & 0xFFFFFFFFull)) : first_idx;
    const unsigned* zl = reinterpret_cast<const unsigned*>(z + (size_t)last * (2 * d2));
    for (int c = threadIdx.x; c < 512; c += 256) s_piv[c] = c < d2 ? zl[c] : 0u;   // (zero beyond d2: the padded tail of the last chunk)
    __syncthreads();
    unsigned long long best = 0ull;
    const int q = q_begin + blockIdx.x * 256 + threadIdx.x;        // this thread's group of four rows
    if (q < q_end) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const uint4* src = zT + q;
        __half md[4];
        *reinterpret_cast<uint2*>(md) = *reinterpret_cast<const uint2*>(min_d + (size_t)q * 4);
        // Chunks of eight dimension pairs, DOUBLE-BUFFERED: the next chunk's eight 16-byte loads are issued before the current
        // chunk is summed (and, with EARLY, voted on), so a wave always has 8-16 KiB in flight -- with the loads of a chunk issued
        // only after the previous chunk's vote the three waves per SIMD of this grid left the memory pipe idle between chunks
        // (109 us per round at 765 184 x 334; profiles/r4_notes.md).  The sum per row runs over c in the same order as before.
        const int nchunk = (d2 + 7) / 8;
        uint4 buf[2][8];
        auto fetch = [&](int ch, uint4 (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = ch * 8 + e;
                dst[e] = c < d2 ? src[(size_t)c * n4] : uint4{0u, 0u, 0u, 0u};
            }
        };
        auto sum = [&](int ch, const uint4 (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned pv = s_piv[ch * 8 + e];
                s0 = sq_diff_acc(v[e].x, pv, s0);
                s1 = sq_diff_acc(v[e].y, pv, s1);
                s2 = sq_diff_acc(v[e].z, pv, s2);
                s3 = sq_diff_acc(v[e].w, pv, s3);
            }
        };
        fetch(0, buf[0]);
        for (int ch = 0; ch < nchunk; ch += 2) {
            if (ch + 1 < nchunk) fetch(ch + 1, buf[1]);
            sum(ch, buf[0]);
            if constexpr (EARLY) {
                const bool settled = !__hlt(__float2half(sqrtf(s0)), md[0]) && !__hlt(__float2half(sqrtf(s1)), md[1]) &&
                                     !__hlt(__float2half(sqrtf(s2)), md[2]) && !__hlt(__float2half(sqrtf(s3)), md[3]);
                if (__all(settled)) break;   // wave-uniform: every row of the wave keeps its minimum
            }
            if (ch + 1 >= nchunk) break;
            if (ch + 2 < nchunk) fetch(ch + 2, buf[0]);
            sum(ch + 1, buf[1]);
            if constexpr (EARLY) {
                const bool settled = !__hlt(__float2half(sqrtf(s0)), md[0]) && !__hlt(__float2half(sqrtf(s1)), md[1]) &&
                                     !__hlt(__float2half(sqrtf(s2)), md[2]) && !__hlt(__float2half(sqrtf(s3)), md[3]);
                if (__all(settled)) break;
            }
        }
        const float ss[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            const __half dist = __float2half(sqrtf(ss[r]));
            const __half nm = __hlt(dist, md[r]) ? dist : md[r];
            md[r] = nm;
            if (row < n) {
                const unsigned long long k = ((unsigned long long)__float_as_uint(__half2float(nm)) << 32) | (0xFFFFFFFFu - (unsigned)row);
                best = k > best ? k : best;
            }
        }
        *reinterpret_cast<uint2*>(min_d + (size_t)q * 4) = *reinterpret_cast<const uint2*>(md);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o > best ? o : best;
    }
    if (lane == 0) s_key[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = s_key[0];
        for (int w = 1; w < 4; ++w) m = s_key[w] > m ? s_key[w] : m;
        atomicMax(best_cur, m);
    }
}

// initial min distances in fp32 from fp32 z (features.py:378 runs before the .half() of :389-391)
__global__ __launch_bounds__(256) void coreset_init_kernel(const float* __restrict__ z32, int n, int n4, int d, int first_idx,
                                                           __half* __restrict__ z16, __half* __restrict__ zT,
                                                           __half* __restrict__ min_d)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* zl = z32 + (size_t)first_idx * d;
    const size_t tstride = (size_t)n4 * 8;   // halves between consecutive dimension pairs of zT: n4 groups x (4 rows x 2)
    for (int row = blockIdx.x * 4 + wave; row < n4 * 4; row += gridDim.x * 4) {
        if (row >= n) {  // padding rows of the last group: zeros, distance 0 (never the arg-max)
            for (int c = lane; c < d; c += 64) zT[(size_t)(c >> 1) * tstride + (size_t)row * 2 + (c & 1)] = __float2half(0.f);
            if (lane == 0) min_d[row] = __float2half(0.f);
            continue;
        }
        const float* zr = z32 + (size_t)row * d;
        float s = 0.0f;
        for (int c = lane; c < d; c += 64) {
            const float v = zr[c];
            const float df = v - zl[c];
            s += df * df;
            const __half h = __float2half(v);
            z16[(size_t)row * d + c] = h;
            zT[(size_t)(c >> 1) * tstride + (size_t)row * 2 + (c & 1)] = h;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) min_d[row] = __float2half(sqrtf(s));
    }
}

// coreset_dtype 'TF32' (features.py:390-391, main.py:151): torch.backends.cuda.matmul.allow_tf32 only changes matrix products, and
// the loop has none -- the branch is the same greedy selection on the UNROUNDED fp32 rows: dist = sqrt(sum (z_i - z_last)^2) in fp32
// (torch's fp32 norm; its summation order is not specified: "parity unpinned" beyond the golden's size), fp32 running minimum.
// Same layout idea: zT32[c][row] (one float per dimension, a thread owns four consecutive rows: 16-byte loads, 1 KiB per wave
// instruction), the pivot row in LDS, the exact partial-distance exit (sums of squares never decrease, sqrt is monotone).
__global__ __launch_bounds__(256) void coreset_round_f32_kernel(const float* __restrict__ z, const float4* __restrict__ zT, int n, int n4,
                                                                int d, float* __restrict__ min_d,
                                                                const unsigned long long* __restrict__ best_prev,
                                                                unsigned long long* __restrict__ best_cur, int first_idx)
{
    __shared__ float s_piv[1024];
    __shared__ unsigned long long s_key[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int last = best_prev ? (int)(0xFFFFFFFFu - (unsigned)(*best_prev & 0xFFFFFFFFull)) : first_idx;
    const float* zl = z + (size_t)last * d;
    for (int c = threadIdx.x; c < d; c += 256) s_piv[c] = zl[c];
    __syncthreads();
    unsigned long long best = 0ull;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < n4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float4* src = zT + q;
        float4 md = *reinterpret_cast<const float4*>(min_d + (size_t)q * 4);
        int c = 0;
        for (; c + 8 <= d; c += 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 v = src[(size_t)(c + e) * n4];
                const float pv = s_piv[c + e];
                const float d0 = v.x - pv, d1 = v.y - pv, d2 = v.z - pv, d3 = v.w - pv;
                s0 += d0 * d0; s1 += d1 * d1; s2 += d2 * d2; s3 += d3 * d3;
            }
            const bool settled = !(sqrtf(s0) < md.x) && !(sqrtf(s1) < md.y) && !(sqrtf(s2) < md.z) && !(sqrtf(s3) < md.w);
            if (__all(settled)) { c = d; break; }
        }
        for (; c < d; ++c) {
            const float4 v = src[(size_t)c * n4];
            const float pv = s_piv[c];
            const float d0 = v.x - pv, d1 = v.y - pv, d2 = v.z - pv, d3 = v.w - pv;
            s0 += d0 * d0; s1 += d1 * d1; s2 += d2 * d2; s3 += d3 * d3;
        }
        const float ss[4] = {s0, s1, s2, s3};
        float m4[4] = {md.x, md.y, md.z, md.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            const float dist = sqrtf(ss[r]);
            const float nm = dist < m4[r] ? dist : m4[r];
            m4[r] = nm;
            if (row < n) {
                const unsigned long long k = ((unsigned long long)__float_as_uint(nm) << 32) | (0xFFFFFFFFu - (unsigned)row);
                best = k > best ? k : best;
            }
        }
        *reinterpret_cast<float4*>(min_d + (size_t)q * 4) = float4{m4[0], m4[1], m4[2], m4[3]};
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o > best ? o : best;
    }
    if (lane == 0) s_key[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = s_key[0];
        for (int w = 1; w < 4; ++w) m = s_key[w] > m ? s_key[w] : m;
        atomicMax(best_cur, m);
    }
}

__global__ __launch_bounds__(256) void coreset_init_f32_kernel(const float* __restrict__ z32, int n, int n4, int d, int first_idx,
                                                               float* __restrict__ zT, float* __restrict__ min_d)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* zl = z32 + (size_t)first_idx * d;
    const size_t tstride = (size_t)n4 * 4;   // floats between consecutive dimensions of zT
    for (int row = blockIdx.x * 4 + wave; row < n4 * 4; row += gridDim.x * 4) {
        if (row >= n) {
            for (int c = lane; c < d; c += 64) zT[(size_t)c * tstride + row] = 0.f;
            if (lane == 0) min_d[row] = 0.f;
            continue;
        }
        const float* zr = z32 + (size_t)row * d;
        float s = 0.0f;
        for (int c = lane; c < d; c += 64) {
            const float v = zr[c];
            const float df = v - zl[c];
            s += df * df;
            zT[(size_t)c * tstride + row] = v;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) min_d[row] = sqrtf(s);
    }
}

__global__ void coreset_decode_kernel(const unsigned long long* __restrict__ best, int n_sel, int first_idx,
                                      int64_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sel) return;
    out[i] = i == 0 ? first_idx : (int64_t)(0xFFFFFFFFu - (unsigned)(best[i - 1] & 0xFFFFFFFFull));
}

}  // namespace

static size_t up256(size_t b) { return (b + 255) / 256 * 256; }

extern "C" size_t cmdiad_coreset_workspace_bytes(int n, int d, int n_select)
{
    const size_t n4 = ((size_t)n + 3) / 4;
    return up256((size_t)n * d * 2) + up256(n4 * 4 * (size_t)d * 2) + up256(n4 * 4 * 2) + (size_t)(n_select > 0 ? n_select : 1) * 8;
}

// Row-sharded selection (SURVEY 8e "fit-time sharding"): every rank holds the whole projected library (the pivot row of a round can be
// any row) but SCANS only its row range; per round one packed-key all_reduce(MAX) of 8 bytes between the ranks (the caller's: RCCL /
// gloo) turns the per-rank winners into the global one, which is the next round's pivot.  prepare = the first pass of
// cmdiad_coreset_greedy (fp16 copies, transposed layout, distances to row first_idx); round = one scan of rows [row_lo, row_hi);
// keys[r] = (fp32 bits of the winning running minimum) << 32 | ~row -- exactly what the single-device loop chains internally, so the
// picks are identical.  row_lo % 4 == 0.
extern "C" int cmdiad_coreset_prepare(const float* z32, int n, int d, int first_idx, void* workspace, size_t workspace_bytes,
                                      cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z32 && n > 0 && d > 0 && d % 2 == 0 && d <= 1024 && first_idx >= 0 && first_idx < n, CMDIAD_ERR_ARG,
                   "cmdiad_coreset_prepare: bad args (need d even, d <= 1024)");
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_coreset_workspace_bytes(n, d, 1), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_coreset_prepare: workspace too small");
    char* ws = (char*)workspace;
    const int n4 = (n + 3) / 4;
    __half* z16 = (__half*)ws;
    __half* zT = (__half*)(ws + up256((size_t)n * d * 2));
    __half* min_d = (__half*)((char*)zT + up256((size_t)n4 * 4 * d * 2));
    hipLaunchKernelGGL(coreset_init_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, z32, n, n4, d, first_idx, z16, zT, min_d);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_coreset_round(void* workspace, int n, int d, int row_lo, int row_hi, const unsigned long long* pivot_key,
                                    int first_idx, unsigned long long* best_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(workspace && best_out && n > 0 && d > 0 && d % 2 == 0 && d <= 1024 && row_lo >= 0 && row_lo <= row_hi && row_hi <= n,
                   CMDIAD_ERR_ARG, "cmdiad_coreset_round: bad args (0 <= row_lo <= row_hi <= n)");
    if (row_lo == row_hi) return CMDIAD_OK;     // an empty shard proposes nothing: its key stays 0 and loses the MAX (any row_lo)
    CMDIAD_REQUIRE(row_lo % 4 == 0, CMDIAD_ERR_ARG, "cmdiad_coreset_round: row_lo %% 4 == 0 (the scan owns groups of four rows)");
    char* ws = (char*)workspace;
    const int n4 = (n + 3) / 4;
    __half* z16 = (__half*)ws;
    __half* zT = (__half*)(ws + up256((size_t)n * d * 2));
    __half* min_d = (__half*)((char*)zT + up256((size_t)n4 * 4 * d * 2));
    const int q0 = row_lo / 4, q1 = (row_hi + 3) / 4;
    // rows of the last group beyond row_hi belong to the next rank (or are padding): the shard boundaries are 4-row aligned except
    // at n itself, where the kernel's `row < n` test applies
    CMDIAD_REQUIRE(row_hi % 4 == 0 || row_hi == n, CMDIAD_ERR_ARG, "cmdiad_coreset_round: row_hi must be a multiple of 4 or n");
    hipLaunchKernelGGL(coreset_round_kernel<true>, dim3((q1 - q0 + 255) / 256), dim3(256), 0, (hipStream_t)stream, z16, (const uint4*)zT, n, n4,
                       d / 2, min_d, pivot_key, best_out, first_idx, q0, q1);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_coreset_decode(const unsigned long long* keys, int n_select, int first_idx, int64_t* idx_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(keys && idx_out && n_select > 0, CMDIAD_ERR_ARG, "cmdiad_coreset_decode: bad args");
    hipLaunchKernelGGL(coreset_decode_kernel, dim3((n_select + 255) / 256), dim3(256), 0, (hipStream_t)stream, keys, n_select, first_idx, idx_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" size_t cmdiad_coreset_f32_workspace_bytes(int n, int d, int n_select)
{
    const size_t n4 = ((size_t)n + 3) / 4;
    return up256(n4 * 4 * (size_t)d * 4) + up256(n4 * 4 * 4) + (size_t)(n_select > 0 ? n_select : 1) * 8;
}

// coreset_dtype 'TF32': the selection on the fp32 rows (see coreset_round_f32_kernel); the pivot row is read from z32 itself.
extern "C" int cmdiad_coreset_greedy_f32(const float* z32, int n, int d, int n_select, int first_idx, int64_t* idx_out,
                                         void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z32 && idx_out && n > 0 && d > 0 && n_select > 0 && n_select <= n && d <= 1024, CMDIAD_ERR_ARG,
                   "cmdiad_coreset_greedy_f32: bad args (need d <= 1024)");
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_coreset_f32_workspace_bytes(n, d, n_select), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_coreset_greedy_f32: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int n4 = (n + 3) / 4;
    float* zT = (float*)ws;
    float* min_d = (float*)(ws + up256((size_t)n4 * 4 * d * 4));
    unsigned long long* best = (unsigned long long*)((char*)min_d + up256((size_t)n4 * 4 * 4));
    if (hipMemsetAsync(best, 0, (size_t)n_select * 8, s) != hipSuccess) {
        cmdiad_set_error("cmdiad_coreset_greedy_f32: memset failed");
        return CMDIAD_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(coreset_init_f32_kernel, dim3(2048), dim3(256), 0, s, z32, n, n4, d, first_idx, zT, min_d);
    const int grid = (n4 + 255) / 256;
    for (int r = 0; r + 1 < n_select; ++r)
        hipLaunchKernelGGL(coreset_round_f32_kernel, dim3(grid), dim3(256), 0, s, z32, (const float4*)zT, n, n4, d, min_d,
                           r == 0 ? nullptr : best + (r - 1), best + r, first_idx);
    hipLaunchKernelGGL(coreset_decode_kernel, dim3((n_select + 255) / 256), dim3(256), 0, s, best, n_select, first_idx, idx_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_coreset_greedy(const float* z32, int n, int d, int n_select, int first_idx, int64_t* idx_out,
                                     void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z32 && idx_out && n > 0 && d > 0 && n_select > 0 && n_select <= n && d % 2 == 0 && d <= 1024, CMDIAD_ERR_ARG,
                   "cmdiad_coreset_greedy: bad args (need d even, d <= 1024)");
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_coreset_workspace_bytes(n, d, n_select), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_coreset_greedy: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int n4 = (n + 3) / 4;
    __half* z16 = (__half*)ws;
    __half* zT = (__half*)(ws + up256((size_t)n * d * 2));
    __half* min_d = (__half*)((char*)zT + up256((size_t)n4 * 4 * d * 2));
    unsigned long long* best = (unsigned long long*)((char*)min_d + up256((size_t)n4 * 4 * 2));
    if (hipMemsetAsync(best, 0, (size_t)n_select * 8, s) != hipSuccess) {
        cmdiad_set_error("cmdiad_coreset_greedy: memset failed");
        return CMDIAD_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(coreset_init_kernel, dim3(2048), dim3(256), 0, s, z32, n, n4, d, first_idx, z16, zT, min_d);
    const int grid = (n4 + 255) / 256;
    const char* ee = getenv("CMDIAD_CORESET_EARLY");   // =0: every row reads all its dimensions every round (A/B runs; read per call)
    const bool early = !(ee && ee[0] == '0');
    for (int r = 0; r + 1 < n_select; ++r) {
        if (early) hipLaunchKernelGGL(coreset_round_kernel<true>, dim3(grid), dim3(256), 0, s, z16, (const uint4*)zT, n, n4, d / 2, min_d,
                                      r == 0 ? nullptr : best + (r - 1), best + r, first_idx, 0, n4);
        else hipLaunchKernelGGL(coreset_round_kernel<false>, dim3(grid), dim3(256), 0, s, z16, (const uint4*)zT, n, n4, d / 2, min_d,
                                r == 0 ? nullptr : best + (r - 1), best + r, first_idx, 0, n4);
    }
    hipLaunchKernelGGL(coreset_decode_kernel, dim3((n_select + 255) / 256), dim3(256), 0, s, best, n_select, first_idx,
                       idx_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
