// Exact removal of repeated query rows in front of the distance GEMM (features.py:186-190, 227: torch.cdist + min over the library).
//
// The 56 x 56 patch grid of a sample (multiple_features.py:216, features.py:169-184) has one row per patch whether or not any point
// of the cloud lies under it: every patch without a foreground pixel is the SAME vector -- zeros pooled, then (0 - mean) / std -- and
// the reference computes its distance to every library row again for each of them (half of the 3 136 patches of a typical
// MVTec 3D-AD sample and of the bench's synthetic clouds).  The nearest-row search is a pure function of one query row, so those rows
// are searched once:
//
//   const_rows_kernel     tag[q] = the row's 16-bit value if all D elements of the 16-bit query row are that value, else 0
//   dedup_plan_kernel     representative = first tagged row; a row repeats it iff its tag and the bits of its squared norm equal the
//                         representative's (then its 16-bit row and norm -- everything the distance kernel reads -- are identical);
//                         order-preserving compaction of the other rows: rows[slot] = q, slot[q], count
//   gather_rows_kernel    the compacted 16-bit rows and norms
//   cmdiad_l2_min_keys_counted on the compacted set (l2min.hip: device-resident row count)
//   expand_keys_kernel    keys[q] = compact_keys[slot[q]]
//
// Only the search is shared; the exact fp32 re-score and everything after it run per original row.  Rows that repeat in any other way
// are simply searched individually.
#include "common.h"

namespace {

constexpr int kPlanThreads = 1024;
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

__global__ __launch_bounds__(256) void const_rows_kernel(const uint16_t* __restrict__ q, int Q, int D, unsigned* __restrict__ tag)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const uint16_t* r = q + (size_t)row * D;
    const unsigned v = r[0], vv = v | (v << 16);
    bool same = true;
    for (int c = lane; c < D / 8; c += 64) {
        const uint4 x = *reinterpret_cast<const uint4*>(r + c * 8);
        same = same && x.x == vv && x.y == vv && x.z == vv && x.w == vv;
    }
    const bool all = __all(same);
    if (lane == 0) tag[row] = all ? (0x80000000u | v) : 0u;
}

// One block: Q is a few hundred thousand at most (B x 3 136).
__global__ __launch_bounds__(kPlanThreads) void dedup_plan_kernel(const unsigned* __restrict__ tag, const float* __restrict__ qsq, int Q,
                                                                  int* __restrict__ slot, int* __restrict__ rows, int* __restrict__ count)
{
    __shared__ int s_rep, s_rep_slot;
    __shared__ int s_cnt[kPlanThreads];
    const int t = threadIdx.x;
    const int per = (Q + kPlanThreads - 1) / kPlanThreads;
    const int q0 = min(t * per, Q), q1 = min(q0 + per, Q);
    if (t == 0) { s_rep = Q; s_rep_slot = -1; }
    __syncthreads();
    int first = Q;
    for (int q = q0; q < q1; ++q)
        if (tag[q]) { first = q; break; }
    if (first < Q) atomicMin(&s_rep, first);
    __syncthreads();
    const int rep = s_rep;
    const unsigned rep_tag = rep < Q ? tag[rep] : 0u;
    const unsigned rep_sq = rep < Q ? __float_as_uint(qsq[rep]) : 0u;
    auto repeats = [&](int q) { return q != rep && rep_tag != 0u && tag[q] == rep_tag && __float_as_uint(qsq[q]) == rep_sq; };
    int n = 0;
    for (int q = q0; q < q1; ++q) n += repeats(q) ? 0 : 1;
    s_cnt[t] = n;
    __syncthreads();
    // inclusive scan over the 1024 per-thread counts
    for (int off = 1; off < kPlanThreads; off <<= 1) {
        const int add = t >= off ? s_cnt[t - off] : 0;
        __syncthreads();
        s_cnt[t] += add;
        __syncthreads();
    }
    int pos = s_cnt[t] - n;
    for (int q = q0; q < q1; ++q) {
        if (repeats(q)) continue;
        slot[q] = pos;
        rows[pos] = q;
        if (q == rep) s_rep_slot = pos;
        ++pos;
    }
    if (t == kPlanThreads - 1) count[0] = s_cnt[t];
    __syncthreads();
    const int rs = s_rep_slot;
    for (int q = q0; q < q1; ++q)
        if (repeats(q)) slot[q] = rs;
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

extern "C" size_t cmdiad_rows_dedup_workspace_bytes(int Q) { return (size_t)(Q > 0 ? Q : 0) * sizeof(unsigned); }

extern "C" int cmdiad_rows_dedup_plan(const uint16_t* q, const float* q_sqnorm, int Q, int D, void* workspace, int* slot, int* rows,
                                      int* count, uint16_t* q_compact, float* q_sqnorm_compact, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && q_sqnorm && workspace && slot && rows && count && q_compact && q_sqnorm_compact, CMDIAD_ERR_ARG,
                   "cmdiad_rows_dedup_plan: null pointer");
    CMDIAD_REQUIRE(Q >= 0 && D > 0 && D % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_rows_dedup_plan: need D%%8==0 (D=%d)", D);
    CMDIAD_REQUIRE(aligned16(q) && aligned16(q_compact), CMDIAD_ERR_ARG, "cmdiad_rows_dedup_plan: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    unsigned* tag = (unsigned*)workspace;
    if (Q > 0) hipLaunchKernelGGL(const_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, q, Q, D, tag);
    hipLaunchKernelGGL(dedup_plan_kernel, dim3(1), dim3(kPlanThreads), 0, s, tag, q_sqnorm, Q, slot, rows, count);
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
