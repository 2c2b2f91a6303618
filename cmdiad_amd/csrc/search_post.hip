// What surrounds the distance GEMM of l2min.hip in the patch-library scoring (reference feature_extractors/features.py:186-190,
// 225-290; multiple_features.py:976-977): the search operands (normalise + 16-bit cast + row norms), the exact fp32 re-score of the
// search's candidates, and the per-image score head / tail.  Bandwidth-bound kernels of a few microseconds each; kept out of
// l2min.hip so that the dominant kernel's source (whose hash gates bench.py's `roofline.traffic`) changes only when IT does.
#include <stdlib.h>

#include "gemm_core.h"

namespace {

using namespace gemm;

// Exact fp32 distance to the winning row: one wave per query.
__global__ __launch_bounds__(256) void l2_rescore_kernel(const float* __restrict__ q, const float* __restrict__ bank,
                                                         const unsigned long long* __restrict__ keys, int Q, int Nb,
                                                         int D, unsigned row_offset, float* __restrict__ min_val,
                                                         int64_t* __restrict__ min_idx)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= Q) return;
    const unsigned gi = (unsigned)(keys[row] & 0xFFFFFFFFull);
    if (gi < row_offset || gi >= row_offset + (unsigned)Nb) return;  // another shard owns the winner
    const float* a = q + (size_t)row * D;
    const float* b = bank + (size_t)(gi - row_offset) * D;
    float s = 0.0f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    s = wave_sum(s);
    if (lane == 0) {
        min_val[row] = sqrtf(s);
        min_idx[row] = (int64_t)gi;
    }
}

// The same for BOTH candidates of a query (best and runner-up of the 16-bit search, RowMin): the squared fp32 distances in the
// SAME summation order, the smaller one wins, of equal ones the lower row.  d2_pair (optional, [2][Q]): the squared distances of
// the candidates whose rows THIS shard owns (others untouched: the caller sums over the shards, then cmdiad_l2_choose).
// min_val / min_idx (optional): the decision, for queries whose candidates are all local or absent.
__device__ __forceinline__ float row_dist2(const float* __restrict__ a, const float* __restrict__ b, int D, int lane)
{
    float s = 0.0f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    return wave_sum(s);
}

__global__ __launch_bounds__(256) void l2_rescore2_kernel(const float* __restrict__ q, const float* __restrict__ bank,
                                                          const unsigned long long* __restrict__ keys,
                                                          const unsigned long long* __restrict__ keys2, int Q, int Nb, int D,
                                                          unsigned row_offset, float* __restrict__ d2_pair,
                                                          float* __restrict__ min_val, int64_t* __restrict__ min_idx)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= Q) return;
    const unsigned long long k1 = keys[row], k2 = keys2[row];
    const unsigned g1 = (unsigned)(k1 & 0xFFFFFFFFull), g2 = (unsigned)(k2 & 0xFFFFFFFFull);
    // "no candidate": every value field of a real key is the bit pattern of a finite d2 >= 0 (< 0x7F800000)
    const bool has1 = (unsigned)(k1 >> 32) < 0x7F800000u, has2 = (unsigned)(k2 >> 32) < 0x7F800000u;
    const bool own1 = has1 && g1 >= row_offset && g1 < row_offset + (unsigned)Nb;
    const bool own2 = has2 && g2 >= row_offset && g2 < row_offset + (unsigned)Nb;
    const float* a = q + (size_t)row * D;
    float s1 = 0.0f, s2 = 0.0f;
    if (own1) s1 = row_dist2(a, bank + (size_t)(g1 - row_offset) * D, D, lane);
    if (own2) s2 = row_dist2(a, bank + (size_t)(g2 - row_offset) * D, D, lane);
    if (lane == 0) {
        if (d2_pair) {
            if (own1) d2_pair[row] = s1;
            if (own2) d2_pair[(size_t)Q + row] = s2;
        }
        if (min_val && own1) {
            const bool second = own2 && (s2 < s1 || (s2 == s1 && g2 < g1));
            min_val[row] = sqrtf(second ? s2 : s1);
            min_idx[row] = (int64_t)(second ? g2 : g1);
        }
    }
}

// the decision alone, from squared distances summed over the shards (every candidate row is owned by exactly one of them)
__global__ void l2_choose_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ keys2,
                                 const float* __restrict__ d2_pair, int Q, float* __restrict__ min_val, int64_t* __restrict__ min_idx)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= Q) return;
    const unsigned long long k1 = keys[row], k2 = keys2[row];
    if ((unsigned)(k1 >> 32) >= 0x7F800000u) return;
    const bool has2 = (unsigned)(k2 >> 32) < 0x7F800000u;
    const unsigned g1 = (unsigned)(k1 & 0xFFFFFFFFull), g2 = (unsigned)(k2 & 0xFFFFFFFFull);
    const float s1 = d2_pair[row], s2 = d2_pair[(size_t)Q + row];
    const bool second = has2 && (s2 < s1 || (s2 == s1 && g2 < g1));
    min_val[row] = sqrtf(second ? s2 : s1);
    min_idx[row] = (int64_t)(second ? g2 : g1);
}

// (x - mean) * inv_std -> bf16 (+ optional f32 copy, + optional |row|^2 of the ROUNDED values; non-finite rows: see below).
// One wave per row.
template <bool F16>
__global__ __launch_bounds__(256) void normalize_cast_kernel(const float* __restrict__ x, size_t rows, int D, float mean,
                                                             float inv_std, uint16_t* __restrict__ out16,
                                                             float* __restrict__ out_f32, float* __restrict__ sq,
                                                             int group_rows, int group_skip)
{
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    // input rows in groups with rows to skip in front of each (the ViT's tokens [B, 1 + 784, C]: one cls row per image): output
    // row r reads input row r + (r / group_rows + 1) * group_skip -- the patch rows leave compactly, no gather copy in between
    const size_t in_row = group_rows > 0 ? row + (row / (size_t)group_rows + 1) * (size_t)group_skip : row;
    float s = 0.0f;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(x + in_row * D + c);
        v.x = (v.x - mean) * inv_std; v.y = (v.y - mean) * inv_std;
        v.z = (v.z - mean) * inv_std; v.w = (v.w - mean) * inv_std;
        if (out_f32) *reinterpret_cast<float4*>(out_f32 + row * D + c) = v;
        s += ((v.x + v.y) + (v.z + v.w)) * 0.0f;   // NaN as soon as an element is not finite (the fp16 cast below saturates, i.e. hides it)
        float r0, r1, r2, r3;
        if constexpr (F16) {
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
            const float lim = 65504.0f;  // saturate instead of overflowing to inf
            f16x4 o = {(_Float16)fminf(fmaxf(v.x, -lim), lim), (_Float16)fminf(fmaxf(v.y, -lim), lim),
                       (_Float16)fminf(fmaxf(v.z, -lim), lim), (_Float16)fminf(fmaxf(v.w, -lim), lim)};
            if (out16) *reinterpret_cast<f16x4*>(out16 + row * D + c) = o;
            r0 = (float)o[0]; r1 = (float)o[1]; r2 = (float)o[2]; r3 = (float)o[3];
        } else {
            bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
            if (out16) *reinterpret_cast<bf16x4*>(out16 + row * D + c) = o;
            r0 = bf2f(o[0]); r1 = bf2f(o[1]); r2 = bf2f(o[2]); r3 = bf2f(o[3]);
        }
        s += r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
    }
    if (sq) {
        s = wave_sum(s);
        // A row with a non-finite element (or whose squares overflow) must not reach the distance GEMM's unsigned running minimum
        // as NaN / +inf accumulators (they would sort BELOW every finite candidate, RowMin in l2min.hip): its 16-bit copy becomes
        // zeros and its squared norm +inf, so every accumulator that involves it starts at -inf and stays there -- as a library
        // row it never wins, as a query row it finds nothing.  (The fp32 copy keeps what the caller passed.)
        if (!(s < __builtin_inff())) {   // wave-uniform
            s = __builtin_inff();
            if (out16)
                for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<uint2*>(out16 + row * D + c) = uint2{0u, 0u};
        }
        if (lane == 0) sq[row] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Score head / tail (features.py:227-290), one block per image.
//   head: s_idx = argmax(min_val) (first occurrence), s_star = max; gathers m_test = patch[s_idx] and
//         m_star = bank[min_idx[s_idx]] into probe buffers for the re-weighting scan.
//   tail: m_star_knn = || m_test - bank[nn[1:]] || for the 2nd and 3rd nearest rows of m_star,
//         w = 1 - exp(s*/sqrt(D)) / sum(exp(m_star_knn/sqrt(D))), s = w * s*.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_head_kernel(const float* __restrict__ min_val, const int64_t* __restrict__ min_idx,
                                                         const float* __restrict__ patch, const float* __restrict__ bank,
                                                         int Q, int D, unsigned row_offset, int Nb,
                                                         float* __restrict__ s_star, int32_t* __restrict__ s_idx,
                                                         float* __restrict__ m_test, float* __restrict__ m_star)
{
    __shared__ unsigned long long s_key[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* mv = min_val + (size_t)b * Q;
    // max value, lowest index: key = value bits << 32 | (0xFFFFFFFF - idx); distances are non-negative
    unsigned long long best = 0ull;
    for (int i = tid; i < Q; i += 256) {
        const unsigned long long k = ((unsigned long long)__float_as_uint(fmaxf(mv[i], 0.0f)) << 32) | (0xFFFFFFFFu - (unsigned)i);
        best = k > best ? k : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = shfl_xor_u64(best, m);
        best = o > best ? o : best;
    }
    if ((tid & 63) == 0) s_key[tid >> 6] = best;
    __syncthreads();
    best = s_key[0];
    for (int w = 1; w < 4; ++w) best = s_key[w] > best ? s_key[w] : best;
    const int si = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    if (tid == 0) { s_star[b] = mv[si]; s_idx[b] = si; }
    const long long gi = min_idx[(size_t)b * Q + si] - (long long)row_offset;
    const float* pt = patch + ((size_t)b * Q + si) * D;
    for (int c = tid; c < D; c += 256) {
        m_test[(size_t)b * D + c] = pt[c];
        if (gi >= 0 && gi < Nb) m_star[(size_t)b * D + c] = bank[(size_t)gi * D + c];  // owner shard writes it
    }
}

__global__ __launch_bounds__(256) void score_tail_kernel(const float* __restrict__ s_star, const float* __restrict__ m_test,
                                                         const unsigned long long* __restrict__ top3,
                                                         const float* __restrict__ bank, int D, unsigned row_offset, int Nb,
                                                         float* __restrict__ knn_d /*[B,2]*/)
{
    __shared__ float s_part[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = 1; k < 3; ++k) {
        const long long gi = (long long)(top3[b * 3 + k] & 0xFFFFFFFFull) - (long long)row_offset;
        if (gi < 0 || gi >= Nb) continue;  // block-uniform: another shard owns this row
        const float* row = bank + (size_t)gi * D;
        float s = 0.0f;
        for (int c = tid; c < D; c += 256) { const float d = m_test[(size_t)b * D + c] - row[c]; s += d * d; }
        s = wave_sum(s);
        if ((tid & 63) == 0) s_part[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) knn_d[b * 2 + (k - 1)] = sqrtf(s_part[0] + s_part[1] + s_part[2] + s_part[3]);
        __syncthreads();
    }
}

// s = (1 - exp(s*/sqrt(D)) / (exp(k0/sqrt(D)) + exp(k1/sqrt(D)))) * s*      features.py:285-290
__global__ void score_final_kernel(const float* __restrict__ s_star, const float* __restrict__ knn_d, int B, int D,
                                   float* __restrict__ s_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float rd = sqrtf((float)D);
    const float w = 1.0f - expf(s_star[b] / rd) / (expf(knn_d[b * 2] / rd) + expf(knn_d[b * 2 + 1] / rd));
    s_out[b] = w * s_star[b];
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int cmdiad_l2_rescore(const float* q, const float* bank, const unsigned long long* keys, int Q, int Nb,
                                 int D, uint32_t row_offset, float* min_val, int64_t* min_idx, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && bank && keys && min_val && min_idx, CMDIAD_ERR_ARG, "cmdiad_l2_rescore: null pointer");
    CMDIAD_REQUIRE(D % 4 == 0 && aligned16(q) && aligned16(bank), CMDIAD_ERR_ARG, "cmdiad_l2_rescore: D%%4, alignment");
    if (Q == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(l2_rescore_kernel, dim3((Q + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, bank, keys, Q, Nb, D,
                       row_offset, min_val, min_idx);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_l2_rescore2(const float* q, const float* bank, const unsigned long long* keys, const unsigned long long* keys2,
                                  int Q, int Nb, int D, uint32_t row_offset, float* d2_pair, float* min_val, int64_t* min_idx,
                                  cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && bank && keys && keys2 && (d2_pair || (min_val && min_idx)) && (!min_val == !min_idx), CMDIAD_ERR_ARG,
                   "cmdiad_l2_rescore2: null pointer");
    CMDIAD_REQUIRE(D % 4 == 0 && aligned16(q) && aligned16(bank), CMDIAD_ERR_ARG, "cmdiad_l2_rescore2: D%%4, alignment");
    if (Q == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(l2_rescore2_kernel, dim3((Q + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, bank, keys, keys2, Q, Nb, D,
                       row_offset, d2_pair, min_val, min_idx);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_l2_choose(const unsigned long long* keys, const unsigned long long* keys2, const float* d2_pair, int Q,
                                float* min_val, int64_t* min_idx, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(keys && keys2 && d2_pair && min_val && min_idx, CMDIAD_ERR_ARG, "cmdiad_l2_choose: null pointer");
    if (Q == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(l2_choose_kernel, dim3((Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, keys, keys2, d2_pair, Q, min_val,
                       min_idx);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_normalize_cast(const float* x, size_t rows, int D, float mean, float inv_std, uint16_t* out_bf16,
                                     float* out_f32, float* row_sqnorm, int out_dtype, cmdiad_stream_t stream)
{
    return cmdiad_normalize_cast_rows(x, rows, D, 0, 0, mean, inv_std, out_bf16, out_f32, row_sqnorm, out_dtype, stream);
}

extern "C" int cmdiad_normalize_cast_rows(const float* x, size_t rows, int D, int group_rows, int group_skip, float mean,
                                          float inv_std, uint16_t* out_bf16, float* out_f32, float* row_sqnorm, int out_dtype,
                                          cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x, CMDIAD_ERR_ARG, "cmdiad_normalize_cast: null input");
    CMDIAD_REQUIRE(group_rows >= 0 && group_skip >= 0 && (group_rows > 0 || group_skip == 0), CMDIAD_ERR_ARG,
                   "cmdiad_normalize_cast_rows: group_rows=%d group_skip=%d", group_rows, group_skip);
    CMDIAD_REQUIRE(D % 4 == 0 && aligned16(x) && (!out_f32 || aligned16(out_f32)) &&
                       (!out_bf16 || ((uintptr_t)out_bf16 & 7) == 0),
                   CMDIAD_ERR_ARG, "cmdiad_normalize_cast: D%%4==0 and aligned buffers");
    if (rows == 0) return CMDIAD_OK;
    if (out_dtype == CMDIAD_DT_F16)
        hipLaunchKernelGGL(normalize_cast_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                           rows, D, mean, inv_std, out_bf16, out_f32, row_sqnorm, group_rows, group_skip);
    else
        hipLaunchKernelGGL(normalize_cast_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                           rows, D, mean, inv_std, out_bf16, out_f32, row_sqnorm, group_rows, group_skip);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_score_head(const float* min_val, const int64_t* min_idx, const float* patch, const float* bank,
                                 int B, int Q, int D, int Nb, uint32_t row_offset, float* s_star, int32_t* s_idx,
                                 float* m_test, float* m_star, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(min_val && min_idx && patch && bank && s_star && s_idx && m_test && m_star, CMDIAD_ERR_ARG,
                   "cmdiad_score_head: null pointer");
    if (B == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(score_head_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, min_val, min_idx, patch, bank, Q, D,
                       row_offset, Nb, s_star, s_idx, m_test, m_star);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_score_tail(const float* s_star, const float* m_test, const unsigned long long* top3,
                                 const float* bank, int B, int D, int Nb, uint32_t row_offset, float* knn_d,
                                 cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(s_star && m_test && top3 && bank && knn_d, CMDIAD_ERR_ARG, "cmdiad_score_tail: null pointer");
    if (B == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(score_tail_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, s_star, m_test, top3, bank, D,
                       row_offset, Nb, knn_d);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_score_final(const float* s_star, const float* knn_d, int B, int D, float* s_out,
                                  cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(s_star && knn_d && s_out, CMDIAD_ERR_ARG, "cmdiad_score_final: null pointer");
    if (B == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(score_final_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, s_star, knn_d, B, D, s_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
