// Exact removal of repeated query rows in front of the distance GEMM (features.py:186-190, 227: torch.cdist + min over the library).
//
// The 56 x 56 patch grid of a sample (multiple_features.py:216, features.py:169-184) has one row per patch whether or not any point
// of the cloud lies under it: every patch without a foreground pixel is the SAME vector -- zeros pooled, then (0 - mean) / std, or
// the hallucination network's image of that vector (multiple_features.py:596) -- and the reference computes its distance to every
// library row again for each of them (half of the 3 136 patches of a typical MVTec 3D-AD sample and of the bench's synthetic
// clouds).  The nearest-row search is a pure function of one query row, so the most repeated row of a batch is searched once:
//
//   hash_rows_kernel      tag[q] = 32-bit hash of the 16-bit query row (a wave per row)
//   pick_rep_kernel       the most frequent tag (two rounds of 4 096 LDS counters: bits 0-11, then bits 12-23 inside the winning
//                         bucket); representative = first row carrying it
//   verify_rows_kernel    a row REPEATS the representative iff its tag and the bits of its squared norm are equal AND all D
//                         elements compare equal (a wave per candidate row) -- then everything the distance kernel reads for it is
//                         identical.  A hash collision therefore only costs the comparison, never a wrong answer.
//   compact_kernel        order-preserving compaction of the other rows: rows[slot] = q, slot[q], count
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

// info[0] = representative row (Q: none), info[1] = its tag.  One block.
__global__ __launch_bounds__(kPlanThreads) void pick_rep_kernel(const unsigned* __restrict__ tag, int Q, int* __restrict__ info)
{
    __shared__ int s_cnt[4096];
    __shared__ int s_best, s_rep;
    const int t = threadIdx.x;
    int lo = 0;
    for (int round = 0; round < 2; ++round) {
        for (int i = t; i < 4096; i += kPlanThreads) s_cnt[i] = 0;
        if (t == 0) s_best = 0;
        __syncthreads();
        for (int q = t; q < Q; q += kPlanThreads) {
            const unsigned g = tag[q];
            if (round == 0) atomicAdd(&s_cnt[g & 4095u], 1);
            else if ((int)(g & 4095u) == lo) atomicAdd(&s_cnt[(g >> 12) & 4095u], 1);
        }
        __syncthreads();
        // bucket with the highest count, lowest index on ties: (count << 12) | (4095 - index)
        int best = 0;
        for (int i = t; i < 4096; i += kPlanThreads) best = max(best, (min(s_cnt[i], 0x3FFFF) << 12) | (4095 - i));
        atomicMax(&s_best, best);
        __syncthreads();
        const int win = 4095 - (s_best & 4095), n = s_best >> 12;
        __syncthreads();
        if (round == 0) lo = win;
        else {
            if (t == 0) s_rep = Q;
            __syncthreads();
            if (n >= 2) {
                const unsigned want = (unsigned)lo | ((unsigned)win << 12);
                for (int q = t; q < Q; q += kPlanThreads)
                    if ((tag[q] & 0xFFFFFFu) == want) { atomicMin(&s_rep, q); break; }
            }
            __syncthreads();
            if (t == 0) {
                info[0] = s_rep;
                info[1] = s_rep < Q ? (int)tag[s_rep] : 0;
            }
        }
    }
}

// dup[q] = 1 iff row q is a verified repeat of the representative.
__global__ __launch_bounds__(256) void verify_rows_kernel(const uint16_t* __restrict__ q, const float* __restrict__ qsq,
                                                          const unsigned* __restrict__ tag, const int* __restrict__ info, int Q, int D,
                                                          unsigned char* __restrict__ dup)
{
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const int rep = info[0];
    bool cand = rep < Q && row != rep && tag[row] == (unsigned)info[1] && __float_as_uint(qsq[row]) == __float_as_uint(qsq[min(rep, Q - 1)]);
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

// One block: Q is a few hundred thousand at most (B x 3 136).
__global__ __launch_bounds__(kPlanThreads) void compact_kernel(const unsigned char* __restrict__ dup, const int* __restrict__ info, int Q,
                                                               int* __restrict__ slot, int* __restrict__ rows, int* __restrict__ count)
{
    __shared__ int s_rep_slot;
    __shared__ int s_cnt[kPlanThreads];
    const int t = threadIdx.x;
    const int per = (Q + kPlanThreads - 1) / kPlanThreads;
    const int q0 = min(t * per, Q), q1 = min(q0 + per, Q);
    const int rep = info[0];
    if (t == 0) s_rep_slot = -1;
    int n = 0;
    for (int q = q0; q < q1; ++q) n += dup[q] ? 0 : 1;
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
        if (dup[q]) continue;
        slot[q] = pos;
        rows[pos] = q;
        if (q == rep) s_rep_slot = pos;
        ++pos;
    }
    if (t == kPlanThreads - 1) count[0] = s_cnt[t];
    __syncthreads();
    const int rs = s_rep_slot;
    for (int q = q0; q < q1; ++q)
        if (dup[q]) slot[q] = rs;
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

// workspace: tag [Q] u32 | info [4] i32 | dup [Q] u8
extern "C" size_t cmdiad_rows_dedup_workspace_bytes(int Q) { const size_t q = (size_t)(Q > 0 ? Q : 0); return q * 4 + 16 + ((q + 15) & ~(size_t)15); }

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
    if (Q > 0) hipLaunchKernelGGL(hash_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, q, Q, D, tag);
    hipLaunchKernelGGL(pick_rep_kernel, dim3(1), dim3(kPlanThreads), 0, s, tag, Q, info);
    if (Q > 0) hipLaunchKernelGGL(verify_rows_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, q, q_sqnorm, tag, info, Q, D, dup);
    hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(kPlanThreads), 0, s, dup, info, Q, slot, rows, count);
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
