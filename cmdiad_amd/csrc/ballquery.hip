// Ball query and index gathers of the pointnet2_ops API.  The reference imports
// pointnet2_ops.pointnet2_utils (models/models.py:5) but only calls furthest_point_sample / gather_operation; ball_query,
// grouping_operation and QueryAndGroup are provided so that code written against that wheel finds the whole surface.
//
// cmdiad_ball_query: for every query point, the first `nsample` cloud points (in index order) with
// dx*dx + dy*dy + dz*dz < radius^2; the first hit pre-fills every slot; no hit leaves zeros (the upstream CUDA kernel's
// behaviour, restated in oracle orc_ball_query; built with -ffp-contract=off, bit-exact against it).
// One wave per query: 64 points per step, ballot + prefix popcount give every hit its slot in index order, the wave
// stops as soon as nsample hits are placed.
// cmdiad_gather_points: out[b,c,j] = feat[b,c,idx[b,j]] (gather_operation with j over M, grouping_operation with j over
// M * nsample).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ n_valid,
                                                         const float* __restrict__ new_xyz, int N, int M, float r2,
                                                         int nsample, int32_t* __restrict__ idx)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= M) return;
    const int n = n_valid ? n_valid[b] : N;
    const float* q = new_xyz + ((size_t)b * M + j) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    int32_t* out = idx + ((size_t)b * M + j) * nsample;
    const float* cloud = xyz + (size_t)b * N * 3;
    int cnt = 0;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) {
            const float dx = qx - cloud[k * 3], dy = qy - cloud[k * 3 + 1], dz = qz - cloud[k * 3 + 2];
            hit = (dx * dx + dy * dy) + dz * dz < r2;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0) continue;
        if (cnt == 0) {  // first hit of this query: pre-fill every slot with it
            const int first = base + __ffsll((long long)mask) - 1;
            for (int l = lane; l < nsample; l += 64) out[l] = first;
        }
        const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && slot < nsample) out[slot] = k;
        cnt += __popcll(mask);
    }
    if (cnt == 0)
        for (int l = lane; l < nsample; l += 64) out[l] = 0;
}

__global__ __launch_bounds__(256) void gather_points_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                            int C, int N, int J, float* __restrict__ out)
{
    const int b = blockIdx.z, c = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const int i = idx[(size_t)b * J + j];
    out[((size_t)b * C + c) * J + j] = feat[((size_t)b * C + c) * N + i];
}

}  // namespace

extern "C" int cmdiad_ball_query(const float* xyz, const int32_t* n_valid, const float* new_xyz, int B, int N, int M,
                                 float radius, int nsample, int32_t* idx_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(xyz && new_xyz && idx_out, CMDIAD_ERR_ARG, "cmdiad_ball_query: null pointer");
    CMDIAD_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample > 0 && radius > 0.0f, CMDIAD_ERR_ARG,
                   "cmdiad_ball_query: bad sizes B=%d N=%d M=%d nsample=%d radius=%g", B, N, M, nsample, (double)radius);
    if (B == 0 || M == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(ball_query_kernel, dim3((M + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, xyz, n_valid, new_xyz, N, M,
                       radius * radius, nsample, idx_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_gather_points(const float* feat, const int32_t* idx, int B, int C, int N, int J, float* out,
                                    cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(feat && idx && out, CMDIAD_ERR_ARG, "cmdiad_gather_points: null pointer");
    CMDIAD_REQUIRE(B >= 0 && C > 0 && N > 0 && J >= 0 && C <= 65535 && B <= 65535, CMDIAD_ERR_ARG, "cmdiad_gather_points: bad sizes");
    if (B == 0 || J == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(gather_points_kernel, dim3((J + 255) / 256, C, B), dim3(256), 0, (hipStream_t)stream, feat, idx, C, N, J, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
