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
// HBM/L2-bound: n*d*2 bytes per round (511 MB for the xyz bank of 'bagel').
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void coreset_round_kernel(const __half* __restrict__ z, int n, int d,
                                                            __half* __restrict__ min_d,
                                                            const unsigned long long* __restrict__ best_prev,
                                                            unsigned long long* __restrict__ best_cur, int first_idx)
{
    __shared__ unsigned long long s_key[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int last = best_prev ? (int)(0xFFFFFFFFu - (unsigned)(*best_prev & 0xFFFFFFFFull)) : first_idx;
    const __half* zl = z + (size_t)last * d;
    unsigned long long best = 0ull;
    for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
        const __half* zr = z + (size_t)row * d;
        float s = 0.0f;
        for (int c = lane * 2; c < d; c += 128) {
            const __half2 a = *reinterpret_cast<const __half2*>(zr + c);
            const __half2 b = *reinterpret_cast<const __half2*>(zl + c);
            const __half2 df = __hsub2(a, b);  // rounded to fp16 like the reference's z_lib - last_item
            const float2 f = __half22float2(df);
            s += f.x * f.x + f.y * f.y;
        }
        if ((d & 1) && lane == 0) {
            const float f = __half2float(__hsub(zr[d - 1], zl[d - 1]));
            s += f * f;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) {
            const __half dist = __float2half(sqrtf(s));
            const __half cur = min_d[row];
            const __half nm = __hlt(dist, cur) ? dist : cur;
            min_d[row] = nm;
            const unsigned long long k = ((unsigned long long)__float_as_uint(__half2float(nm)) << 32) | (0xFFFFFFFFu - (unsigned)row);
            best = k > best ? k : best;
        }
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
__global__ __launch_bounds__(256) void coreset_init_kernel(const float* __restrict__ z32, int n, int d, int first_idx,
                                                           __half* __restrict__ z16, __half* __restrict__ min_d)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* zl = z32 + (size_t)first_idx * d;
    for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
        const float* zr = z32 + (size_t)row * d;
        float s = 0.0f;
        for (int c = lane; c < d; c += 64) {
            const float v = zr[c];
            const float df = v - zl[c];
            s += df * df;
            z16[(size_t)row * d + c] = __float2half(v);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) min_d[row] = __float2half(sqrtf(s));
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

extern "C" size_t cmdiad_coreset_workspace_bytes(int n, int d, int n_select)
{
    size_t z16 = ((size_t)n * d * 2 + 255) / 256 * 256;
    size_t md = ((size_t)n * 2 + 255) / 256 * 256;
    return z16 + md + (size_t)(n_select > 0 ? n_select : 1) * 8;
}

extern "C" int cmdiad_coreset_greedy(const float* z32, int n, int d, int n_select, int first_idx, int64_t* idx_out,
                                     void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z32 && idx_out && n > 0 && d > 0 && n_select > 0 && n_select <= n && d % 2 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_coreset_greedy: bad args (need d even)");
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_coreset_workspace_bytes(n, d, n_select), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_coreset_greedy: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    __half* z16 = (__half*)ws;
    const size_t z16b = ((size_t)n * d * 2 + 255) / 256 * 256;
    __half* min_d = (__half*)(ws + z16b);
    unsigned long long* best = (unsigned long long*)(ws + z16b + ((size_t)n * 2 + 255) / 256 * 256);
    if (hipMemsetAsync(best, 0, (size_t)n_select * 8, s) != hipSuccess) {
        cmdiad_set_error("cmdiad_coreset_greedy: memset failed");
        return CMDIAD_ERR_LAUNCH;
    }
    const int grid = 2048;
    hipLaunchKernelGGL(coreset_init_kernel, dim3(grid), dim3(256), 0, s, z32, n, d, first_idx, z16, min_d);
    for (int r = 0; r + 1 < n_select; ++r)
        hipLaunchKernelGGL(coreset_round_kernel, dim3(grid), dim3(256), 0, s, z16, n, d, min_d,
                           r == 0 ? nullptr : best + (r - 1), best + r, first_idx);
    hipLaunchKernelGGL(coreset_decode_kernel, dim3((n_select + 255) / 256), dim3(256), 0, s, best, n_select, first_idx,
                       idx_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
