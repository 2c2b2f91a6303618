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
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCPB = 4;        // centres per block
constexpr int kCap = 1024;     // keys per centre buffer (power of two, bitonic)
constexpr int kChunk = 512;    // points per streaming step; prune when cnt > kCap - kChunk
constexpr unsigned long long kInf = ~0ull;

__device__ __forceinline__ void bitonic_sort_1024(unsigned long long* s, int tid)
{
    for (int k = 2; k <= kCap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int r = 0; r < kCap / 2 / kThreads; ++r) {
                const int t = r * kThreads + tid;          // compare-exchange id, 0..511
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const unsigned long long a = s[i], b = s[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s[i] = b; s[p] = a; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kThreads) void knn_group_kernel(const float* __restrict__ xyz,
                                                             const int32_t* __restrict__ n_valid,
                                                             const float* __restrict__ center, int N, int G, int K,
                                                             int64_t* __restrict__ idx_out,
                                                             float* __restrict__ neigh_out)
{
    __shared__ unsigned long long s_keys[kCPB][kCap];
    __shared__ unsigned long long s_tau[kCPB];
    __shared__ int s_cnt[kCPB];

    const int b = blockIdx.y;
    const int g0 = blockIdx.x * kCPB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n = n_valid ? n_valid[b] : N;
    const float* p = xyz + (size_t)b * N * 3;

    float cx[kCPB], cy[kCPB], cz[kCPB];
#pragma unroll
    for (int c = 0; c < kCPB; ++c) {
        const int g = min(g0 + c, G - 1);
        const float* cc = center + ((size_t)b * G + g) * 3;
        cx[c] = cc[0]; cy[c] = cc[1]; cz[c] = cc[2];
    }
    if (tid < kCPB) { s_cnt[tid] = 0; s_tau[tid] = kInf; }
    __syncthreads();

    // Visit the 512-point chunks in a strided (coprime) order instead of raster order: the cloud comes
    // from an organised scan, so a raster walk APPROACHES every centre monotonically and almost every
    // point would beat the running threshold; a scattered walk makes tau representative after a few
    // chunks (expected appends ~ K ln(chunks)).  The selected set and its order do not depend on it.
    const int nchunks = (n + kChunk - 1) / kChunk;
    int cstride = (int)(0.6180339887f * (float)nchunks) | 1;
    for (;; cstride += 2) {
        int a = cstride, bb = nchunks;
        while (bb) { const int t = a % bb; a = bb; bb = t; }
        if (a == 1) break;
    }
    int cidx = 0;
    for (int step = 0; step < nchunks; ++step) {
        const int base = cidx * kChunk;
        cidx += cstride;
        if (cidx >= nchunks) cidx %= nchunks;
        unsigned long long tau[kCPB];
#pragma unroll
        for (int c = 0; c < kCPB; ++c) tau[c] = s_tau[c];
#pragma unroll
        for (int r = 0; r < kChunk / kThreads; ++r) {
            const int k = base + r * kThreads + tid;
            const bool inb = k < n;
            float x = 0.f, y = 0.f, z = 0.f;
            if (inb) { x = p[k * 3 + 0]; y = p[k * 3 + 1]; z = p[k * 3 + 2]; }
#pragma unroll
            for (int c = 0; c < kCPB; ++c) {
                const float dx = x - cx[c], dy = y - cy[c], dz = z - cz[c];
                const float d = (dx * dx + dy * dy) + dz * dz;
                const unsigned long long key = pack_key(d, (unsigned)k);
                const bool pass = inb && key < tau[c];
                const unsigned long long m = __ballot(pass);
                if (m) {
                    const int leader = __ffsll((long long)m) - 1;
                    int slot = 0;
                    if (lane == leader) slot = atomicAdd(&s_cnt[c], __popcll(m));
                    slot = __shfl(slot, leader, 64);
                    if (pass) s_keys[c][slot + __popcll(m & ((1ull << lane) - 1ull))] = key;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kCPB; ++c) {
            const int cnt = s_cnt[c];                       // block-uniform
            if (cnt > kCap - kChunk) {
                for (int i = cnt + tid; i < kCap; i += kThreads) s_keys[c][i] = kInf;
                __syncthreads();
                bitonic_sort_1024(s_keys[c], tid);
                if (tid == 0) {
                    s_cnt[c] = min(cnt, K);
                    if (cnt >= K) s_tau[c] = s_keys[c][K - 1];
                }
                __syncthreads();
            }
        }
    }

    // final ordering + epilogue
#pragma unroll
    for (int c = 0; c < kCPB; ++c) {
        const int cnt = s_cnt[c];
        for (int i = cnt + tid; i < kCap; i += kThreads) s_keys[c][i] = kInf;
        __syncthreads();
        bitonic_sort_1024(s_keys[c], tid);
        const int g = g0 + c;
        if (g < G) {
            const int have = min(cnt, K);
            for (int k = tid; k < K; k += kThreads) {
                const int i = k < have ? (int)(s_keys[c][k] & 0xFFFFFFFFull) : 0;
                const size_t o = ((size_t)b * G + g) * K + k;
                if (idx_out) idx_out[o] = i;
                if (neigh_out) {
                    neigh_out[o * 3 + 0] = p[i * 3 + 0] - cx[c];
                    neigh_out[o * 3 + 1] = p[i * 3 + 1] - cy[c];
                    neigh_out[o * 3 + 2] = p[i * 3 + 2] - cz[c];
                }
            }
        }
        __syncthreads();
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
    dim3 grid((G + kCPB - 1) / kCPB, B);
    hipLaunchKernelGGL(knn_group_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, xyz, n_valid, center, N, G,
                       K, idx_out, neigh_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
