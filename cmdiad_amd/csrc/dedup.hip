// Exact removal of repeated query rows in front of the distance GEMM (features.py:186-190, 227: torch.cdist + min over the library).
//
// The 56 x 56 patch grid of a sample (multiple_features.py:216, features.py:169-184) has one row per patch whether or not any point
// of the cloud lies under it: every patch without a foreground pixel is the SAME vector -- zeros pooled, then (0 - mean) / std, or
// the hallucination network's image of that vector (multiple_features.py:351) -- and the reference computes its distance to every
// library row again for each of them (half of the 3 136 patches of a typical MVTec 3D-AD sample and of the bench's synthetic
// clouds).  The nearest-row search is a pure function of one query row, so the most repeated row of a batch is searched once:
//
//   hash_rows_kernel      tag[q] = 32-bit hash of the 16-bit query row (a wave per row)
//   pick_rep_kernel       the most frequent 24-bit tag prefix among 8 192 evenly spaced rows (two rounds of 4 096 LDS counters)
//   first_match_kernel    representative = first row carrying it
//   verify_rows_kernel    a row REPEATS the representative iff its tag and the bits of its squared norm are equal AND all D
//                         elements compare equal (a wave per candidate row) -- then everything the distance kernel reads for it is
//                         identical.  A hash collision therefore only costs the comparison, never a wrong answer.
//   compact_*_kernel      order-preserving compaction of the other rows (1 024 rows per block: counts, then scan + write):
//                         rows[slot] = q, slot[q], count
//   gather_rows_kernel    the compacted 16-bit rows and norms
//   cmdiad_l2_min_keys_counted on the compacted set (l2min.hip: device-resident row count)
//   expand_keys_kernel    keys[q] = compact_keys[slot[q]]
//
// Only the search is shared; the exact fp32 re-score and everything after it run per original row.  Rows that repeat in any other way
// (a second repeated value) are simply searched individually.
#include "common.h"

namespace {

constexpr int kPlanThreads = 1024;
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

__global__ __launch_bounds__(256) void hash_rows_kernel(const uint16_t* __restrict__ q, int Q, int D, unsigned* __restrict__ tag)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const uint16_t* r = q + (size_t)row * D;
    unsigned h = 0;
    for (int c = lane; c < D / 8; c += 64) {
        const uint4 x = *reinterpret_cast<const uint4*>(r + c * 8);
        const unsigned m = (x.x * 0x9E3779B1u) ^ (x.y * 0x85EBCA77u) ^ (x.z * 0xC2B2AE3Du) ^ (x.w * 0x27D4EB2Fu);
        h += (m ^ (m >> 15)) * (2u * c + 1u) + c * 0x165667B1u;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o);
    if (lane == 0) tag[row] = h ^ (h >> 13);
}

// The most frequent 24-bit tag prefix among up to 8 192 evenly spaced rows (all rows when Q <= 8 192): two rounds of 4 096 LDS
// counters, bits 0-11, then bits 12-23 inside the winning bucket.  A repeated row worth removing shows in any such sample; counting
// every row instead costs ~50 000 atomics on ONE counter (measured: 0.5 ms as global atomics, 0.1 ms in the LDS of one block).
// info[0] = Q (first_match_kernel lowers it to the first row carrying the tag), info[1] = the full tag of the first sampled row with
// the winning prefix, info[2] = 1 iff the prefix occurred at least twice.  One block.
constexpr int kSample = 8192;
__global__ __launch_bounds__(kPlanThreads) void pick_rep_kernel(const unsigned* __restrict__ tag, int Q, int* __restrict__ info)
{
    __shared__ int s_cnt[4096];
    __shared__ int s_best;
    const int t = threadIdx.x;
    const int ns = min(Q, kSample);
    unsigned g[kSample / kPlanThreads];
#pragma unroll
    for (int e = 0; e < kSample / kPlanThreads; ++e) {
        const int i = t + e * kPlanThreads;
        g[e] = i < ns ? tag[(long)i * Q / ns] : 0u;
    }
    int lo = 0;
    for (int round = 0; round < 2; ++round) {
        for (int i = t; i < 4096; i += kPlanThreads) s_cnt[i] = 0;
        if (t == 0) s_best = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < kSample / kPlanThreads; ++e) {
            if (t + e * kPlanThreads >= ns) continue;
            if (round == 0) atomicAdd(&s_cnt[g[e] & 4095u], 1);
            else if ((int)(g[e] & 4095u) == lo) atomicAdd(&s_cnt[(g[e] >> 12) & 4095u], 1);
        }
        __syncthreads();
        // bucket with the highest count, lowest index on ties: (count << 12) | (4095 - index)
        int best = 0;
        for (int i = t; i < 4096; i += kPlanThreads) best = max(best, (s_cnt[i] << 12) | (4095 - i));
        atomicMax(&s_best, best);
        __syncthreads();
        const int win = 4095 - (s_best & 4095), n = s_best >> 12;
        __syncthreads();
        if (round == 0) lo = win;
        else {
            // the full tag of the first sampled row carrying the winning prefix (the repeated row dominates its bucket)
            const unsigned want = (unsigned)lo | ((unsigned)win << 12);
            if (t == 0) s_best = kSample;
            __syncthreads();
#pragma unroll
            for (int e = kSample / kPlanThreads - 1; e >= 0; --e)
                if (t + e * kPlanThreads < ns && (g[e] & 0xFFFFFFu) == want) atomicMin(&s_best, t + e * kPlanThreads);
            __syncthreads();
            const int first = s_best;
            if (t == 0) { info[0] = Q; info[2] = n >= 2 && first < kSample; }
#pragma unroll
            for (int e = 0; e < kSample / kPlanThreads; ++e)
                if (t + e * kPlanThreads == first) info[1] = (int)g[e];
        }
    }
}

// info[0] = first row with the chosen tag (one atomic per block of 1 024 rows).
__global__ __launch_bounds__(kPlanThreads) void first_match_kernel(const unsigned* __restrict__ tag, int Q, int* __restrict__ info)
{
    __shared__ int s_min;
    if (!info[2]) return;
    const unsigned want = (unsigned)info[1];
    if (threadIdx.x == 0) s_min = Q;
    __syncthreads();
    const int q = blockIdx.x * kPlanThreads + threadIdx.x;
    const bool hit = q < Q && tag[q] == want;
    const unsigned long long m = __ballot(hit);
    if (m && (threadIdx.x & 63) == 0) atomicMin(&s_min, q + __builtin_ctzll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_min < Q) atomicMin(&info[0], s_min);
}

// dup[q] = 1 iff row q is a verified repeat of the representative.
__global__ __launch_bounds__(256) void verify_rows_kernel(const uint16_t* __restrict__ q, const float* __restrict__ qsq,
                                                          const unsigned* __restrict__ tag, const int* __restrict__ info, int Q, int D,
                                                          unsigned char* __restrict__ dup)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const int rep = info[0];
    bool cand = rep < Q && row != rep && tag[row] == tag[min(rep, Q - 1)] && __float_as_uint(qsq[row]) == __float_as_uint(qsq[min(rep, Q - 1)]);
    cand = __builtin_amdgcn_readfirstlane(cand);
    bool same = cand;
    if (cand) {
        const uint4* a = reinterpret_cast<const uint4*>(q + (size_t)row * D);
        const uint4* b = reinterpret_cast<const uint4*>(q + (size_t)rep * D);
        for (int c = lane; c < D / 8; c += 64) {
            const uint4 x = a[c], y = b[c];
            same = same && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
        }
        same = __all(same);
    }
    if (lane == 0) dup[row] = same ? 1 : 0;
}

// Order-preserving compaction of the rows that are not repeats, 1 024 rows per block.  Every repeat comes AFTER the representative
// (its first occurrence) and nothing before the representative is dropped, so the representative's slot is its own index.
__device__ __forceinline__ int block_exclusive_keep(bool keep, int* s_wave, int& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kPlanThreads / 64; ++w) {
        const int c = s_wave[w];
        base += w < wave ? c : 0;
        sum += c;
    }
    total = sum;
    return base + before;
}

__global__ __launch_bounds__(kPlanThreads) void compact_count_kernel(const unsigned char* __restrict__ dup, int Q, int* __restrict__ sums)
{
    __shared__ int s_wave[kPlanThreads / 64];
    const int q = blockIdx.x * kPlanThreads + threadIdx.x;
    int total;
    block_exclusive_keep(q < Q && !dup[q], s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kPlanThreads) void compact_write_kernel(const unsigned char* __restrict__ dup, const int* __restrict__ sums,
                                                                     const int* __restrict__ info, int Q, int* __restrict__ slot,
                                                                     int* __restrict__ rows, int* __restrict__ count)
{
    __shared__ int s_wave[kPlanThreads / 64];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    int part = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += kPlanThreads) part += sums[i];
    if (part) atomicAdd(&s_base, part);
    __syncthreads();
    const int base = s_base;
    const int q = blockIdx.x * kPlanThreads + threadIdx.x;
    const bool in = q < Q, keep = in && !dup[q];
    int total;
    const int pos = base + block_exclusive_keep(keep, s_wave, total);
    if (keep) {
        slot[q] = pos;
        rows[pos] = q;
    } else if (in) slot[q] = info[0];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) count[0] = base + total;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ q, const float* __restrict__ qsq,
                                                          const int* __restrict__ rows, const int* __restrict__ count, int D,
                                                          uint16_t* __restrict__ qc, float* __restrict__ qsqc)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= *count) return;
    const int src = rows[i];
    const uint4* a = reinterpret_cast<const uint4*>(q + (size_t)src * D);
    uint4* b = reinterpret_cast<uint4*>(qc + (size_t)i * D);
    for (int c = lane; c < D / 8; c += 64) b[c] = a[c];
    if (lane == 0) qsqc[i] = qsq[src];
}

__global__ __launch_bounds__(256) void expand_keys_kernel(const unsigned long long* __restrict__ kc, const int* __restrict__ slot, int Q,
                                                          unsigned long long* __restrict__ keys)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < Q) keys[q] = kc[slot[q]];
}

}  // namespace

// workspace: tag [Q] u32 | info [4] i32 | dup [Q] u8 (padded to 16) | sums [ceil(Q / 1024)] i32
__global__ __launch_bounds__(256) void expand_rows_f32_kernel(const float4* __restrict__ src, const int* __restrict__ slot, int Q, int D4,
                                                              float4* __restrict__ out)
{
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const float4* a = src + (size_t)slot[q] * D4;
    float4* b = out + (size_t)q * D4;
    for (int c = lane; c < D4; c += 64) b[c] = a[c];
}

extern "C" size_t cmdiad_rows_dedup_workspace_bytes(int Q)
{
    const size_t q = (size_t)(Q > 0 ? Q : 0);
    return q * 4 + 16 + ((q + 15) & ~(size_t)15) + (q + kPlanThreads - 1) / kPlanThreads * 4;
}

extern "C" int cmdiad_rows_dedup_plan(const uint16_t* q, const float* q_sqnorm, int Q, int D, void* workspace, int* slot, int* rows,
                                      int* count, uint16_t* q_compact, float* q_sqnorm_compact, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && q_sqnorm && workspace && slot && rows && count && q_compact && q_sqnorm_compact, CMDIAD_ERR_ARG,
                   "cmdiad_rows_dedup_plan: null pointer");
    CMDIAD_REQUIRE(Q >= 0 && D > 0 && D % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_rows_dedup_plan: need D%%8==0 (D=%d)", D);
    CMDIAD_REQUIRE(aligned16(q) && aligned16(q_compact), CMDIAD_ERR_ARG, "cmdiad_rows_dedup_plan: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    unsigned* tag = (unsigned*)workspace;
    int* info = (int*)(tag + (Q > 0 ? Q : 0));
    unsigned char* dup = (unsigned char*)(info + 4);
    const int nb = (Q + kPlanThreads - 1) / kPlanThreads;
    int* sums = (int*)(dup + (((size_t)(Q > 0 ? Q : 0) + 15) & ~(size_t)15));
    if (Q > 0) hipLaunchKernelGGL(hash_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, q, Q, D, tag);
    hipLaunchKernelGGL(pick_rep_kernel, dim3(1), dim3(kPlanThreads), 0, s, tag, Q, info);
    if (Q > 0) hipLaunchKernelGGL(first_match_kernel, dim3(nb), dim3(kPlanThreads), 0, s, tag, Q, info);
    if (Q > 0) hipLaunchKernelGGL(verify_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, q, q_sqnorm, tag, info, Q, D, dup);
    if (Q > 0) {
        hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(kPlanThreads), 0, s, dup, Q, sums);
        hipLaunchKernelGGL(compact_write_kernel, dim3(nb), dim3(kPlanThreads), 0, s, dup, sums, info, Q, slot, rows, count);
    } else if (hipMemsetAsync(count, 0, sizeof(int), s) != hipSuccess) {
        cmdiad_set_error("cmdiad_rows_dedup_plan: hipMemsetAsync failed");
        return CMDIAD_ERR_LAUNCH;
    }
    if (Q > 0) hipLaunchKernelGGL(gather_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, q, q_sqnorm, rows, count, D, q_compact, q_sqnorm_compact);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_keys_expand(const unsigned long long* keys_compact, const int* slot, int Q, unsigned long long* keys,
                                  cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(keys_compact && slot && keys, CMDIAD_ERR_ARG, "cmdiad_keys_expand: null pointer");
    if (Q <= 0) return CMDIAD_OK;
    hipLaunchKernelGGL(expand_keys_kernel, dim3((Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, keys_compact, slot, Q, keys);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_rows_expand_f32(const float* rows_compact, const int* slot, int Q, int D, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(rows_compact && slot && out, CMDIAD_ERR_ARG, "cmdiad_rows_expand_f32: null pointer");
    CMDIAD_REQUIRE(D > 0 && D % 4 == 0 && aligned16(rows_compact) && aligned16(out), CMDIAD_ERR_ARG, "cmdiad_rows_expand_f32: D%%4, alignment");
    if (Q <= 0) return CMDIAD_OK;
    hipLaunchKernelGGL(expand_rows_f32_kernel, dim3((Q + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float4*)rows_compact, slot, Q, D / 4,
                       (float4*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
