// Training-side kernels of the feature-to-feature distillation network (reference
// models/hallucination_network.py:47-69 losses; hallucination_network_pretrain.py:102-159 update step;
// torch.optim.Adam as constructed at :261).  The GEMMs (forward, dX, dW with split-K) are
// cmdiad_gemm_bf16; this file holds the bandwidth-bound pieces around them:
//   * fused loss head: y = GELU(z3), per-row loss (l2 / cos_dist / smooth_l1), dL/dz3 (through the GELU) as bf16
//   * deterministic reductions: slab sum (split-K partials, column partials), column sums for bias grads,
//     LayerNorm parameter gradients
//   * fused Adam update that also refreshes the bf16 GEMM operands (W and W^T)
#include "common.h"

namespace {

__device__ __forceinline__ float gelu_grad(float x) { return gelu_erf_grad(x); }

// mode 0: l2 (sum_rows ||y-t||), 1: cos_dist (sum_rows 1-cos), 2: smooth_l1 (sum of elements, beta 1).
// One wave per row, D % 4 == 0.  scale = grad_scale / B is folded into dz.
__global__ __launch_bounds__(256) void loss_head_kernel(const float* __restrict__ z3, const float* __restrict__ target, int M,
                                                        int D, int mode, float inv_b, float* __restrict__ row_loss,
                                                        bf16_t* __restrict__ dz3, float* __restrict__ y_out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    // output activation (mode >> 8): 0 = GELU (the MLP head's closing activation), 1 = none, 2 = sigmoid of BOTH the output and
    // the target (the convolutional head with sigmoid=True, hallucination_network.py:136-140)
    const int out_act = mode >> 8;
    mode &= 255;
    auto act = [&](float v) { return out_act == 0 ? gelu_erf(v) : out_act == 1 ? v : 1.0f / (1.0f + __expf(-v)); };
    auto tgt = [&](float v) { return out_act == 2 ? 1.0f / (1.0f + __expf(-v)) : v; };
    const float* z = z3 + (size_t)row * D;
    const float* t = target + (size_t)row * D;
    float a = 0.f, b = 0.f, c = 0.f;  // l2: a = sum d^2 ; cos: a = y.t, b = y.y, c = t.t ; smooth: a = sum elementwise
    for (int i = lane * 4; i < D; i += 256) {
        const float4 zz = *reinterpret_cast<const float4*>(z + i);
        const float4 tt = *reinterpret_cast<const float4*>(t + i);
        const float y[4] = {act(zz.x), act(zz.y), act(zz.z), act(zz.w)};
        const float tv[4] = {tgt(tt.x), tgt(tt.y), tgt(tt.z), tgt(tt.w)};
        if (y_out) *reinterpret_cast<float4*>(y_out + (size_t)row * D + i) = make_float4(y[0], y[1], y[2], y[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = y[e] - tv[e];
            if (mode == 0) a += d * d;
            else if (mode == 1) { a += y[e] * tv[e]; b += y[e] * y[e]; c += tv[e] * tv[e]; }
            else { const float ad = fabsf(d); a += ad < 1.0f ? 0.5f * d * d : ad - 0.5f; }
        }
    }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    float loss, k0 = 0.f, k1 = 0.f;  // dL/dy = k0 * (y - t) [l2]  |  k0 * t + k1 * y [cos]
    if (mode == 0) {
        const float nrm = sqrtf(a);
        loss = nrm;
        k0 = nrm > 0.f ? inv_b / nrm : 0.f;
    } else if (mode == 1) {
        const float ny = fmaxf(sqrtf(b), 1e-8f), nt = fmaxf(sqrtf(c), 1e-8f);
        const float cs = a / (ny * nt);
        loss = 1.0f - cs;
        k0 = -inv_b / (ny * nt);
        k1 = inv_b * cs / (ny * ny);
    } else loss = a;
    if (lane == 0) row_loss[row] = loss;
    if (!dz3) return;
    for (int i = lane * 4; i < D; i += 256) {
        const float4 zz = *reinterpret_cast<const float4*>(z + i);
        const float4 tt = *reinterpret_cast<const float4*>(t + i);
        const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, tv[4] = {tgt(tt.x), tgt(tt.y), tgt(tt.z), tgt(tt.w)};
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = act(zv[e]);
            float dy;
            if (mode == 0) dy = k0 * (y - tv[e]);
            else if (mode == 1) dy = k0 * tv[e] + k1 * y;
            else { const float d = y - tv[e]; dy = inv_b * (fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f)); }
            o[e] = f2bf(dy * (out_act == 0 ? gelu_grad(zv[e]) : out_act == 1 ? 1.0f : y * (1.0f - y)));
        }
        *reinterpret_cast<bf16x4*>(dz3 + (size_t)row * D + i) = o;
    }
}

// out[i] = scale * sum_s slabs[s][i]  (fixed order: bit-reproducible).  n % 4 == 0.  Block = 64 float4 columns x 4 slab lanes (lane
// l sums slabs l, l + 4, ...; the four lanes are combined in a fixed order): with one thread per column the 64 slabs of a weight
// gradient were 64 dependent loads -- 17 us per launch, 109 launches in a training step of the HRNet trunk.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int S, size_t n4, size_t stride,
                                                           float scale, float* __restrict__ out)
{
    __shared__ float4 s_p[4][64];
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4)
        for (int s = sl; s < S; s += 4) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)s * stride + i * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    s_p[sl][col] = acc;
    __syncthreads();
    if (sl == 0 && i < n4) {
        const float4 a = s_p[0][col], b = s_p[1][col], c = s_p[2][col], d = s_p[3][col];
        float4 r;
        r.x = ((a.x + b.x) + (c.x + d.x)) * scale; r.y = ((a.y + b.y) + (c.y + d.y)) * scale;
        r.z = ((a.z + b.z) + (c.z + d.z)) * scale; r.w = ((a.w + b.w) + (c.w + d.w)) * scale;
        *reinterpret_cast<float4*>(out + i * 4) = r;
    }
}

// sum of a vector, single block, fixed order.
__global__ __launch_bounds__(1024) void sum_vector_kernel(const float* __restrict__ x, size_t n, float scale,
                                                          float* __restrict__ out)
{
    __shared__ float s_part[16];
    float s = 0.0f;
    for (size_t i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += s_part[w];
        out[0] = t * scale;
    }
}

// Column partial sums of a bf16 matrix [M,N]: block (x = 64-column group, y = row chunk) -> partial[y][n].
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, int M, int N, int rows_per_chunk,
                                                          float* __restrict__ partial)
{
    __shared__ float s_acc[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, M);
    float s = 0.0f;
    if (c < N)
        for (int r = r0 + (threadIdx.x >> 6); r < r1; r += 4) s += bf2f(x[(size_t)r * N + c]);
    s_acc[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && c < N)
        partial[(size_t)blockIdx.y * N + c] = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

// LayerNorm parameter gradients: partial_g[y][c] = sum_r dh[r][c] * xhat[r][c], partial_b[y][c] = sum_r dh[r][c]
// with xhat recomputed from x and the saved row statistics.
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int M, int C, int rows_per_chunk, float* __restrict__ partial_g,
                                                            float* __restrict__ partial_b)
{
    __shared__ float s_g[4][64], s_b[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, M);
    float g = 0.f, b = 0.f;
    if (c < C)
        for (int r = r0 + (threadIdx.x >> 6); r < r1; r += 4) {
            const float d = dh[(size_t)r * C + c];
            g += d * (x[(size_t)r * C + c] - mean[r]) * rstd[r];
            b += d;
        }
    s_g[threadIdx.x >> 6][threadIdx.x & 63] = g;
    s_b[threadIdx.x >> 6][threadIdx.x & 63] = b;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        const int l = threadIdx.x;
        partial_g[(size_t)blockIdx.y * C + c] = (s_g[0][l] + s_g[1][l]) + (s_g[2][l] + s_g[3][l]);
        partial_b[(size_t)blockIdx.y * C + c] = (s_b[0][l] + s_b[1][l]) + (s_b[2][l] + s_b[3][l]);
    }
}

// torch.optim.Adam (no weight decay, amsgrad off): one thread per element.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale, bf16_t* __restrict__ p_bf16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p[i] - (lr / bc1) * (mi / denom);
    p[i] = pi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int cmdiad_loss_head(const float* z3, const float* target, int M, int D, int mode, float inv_b,
                                float* row_loss, uint16_t* dz3, float* y_out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z3 && target && row_loss && M > 0 && D % 4 == 0 && mode >= 0 && (mode & 255) <= 2 && (mode >> 8) <= 2, CMDIAD_ERR_ARG,
                   "cmdiad_loss_head: bad args (mode %d)", mode);
    hipLaunchKernelGGL(loss_head_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, z3, target, M, D, mode, inv_b,
                       row_loss, (bf16_t*)dz3, y_out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_reduce_slabs(const float* slabs, int S, size_t n, size_t stride, float scale, float* out,
                                   cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(slabs && out && S > 0 && n % 4 == 0 && stride % 4 == 0, CMDIAD_ERR_ARG, "cmdiad_reduce_slabs: bad args");
    if (n == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream, slabs, S, n / 4, stride,
                       scale, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_sum_vector(const float* x, size_t n, float scale, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && out, CMDIAD_ERR_ARG, "cmdiad_sum_vector: null pointer");
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, scale, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_colsum_bf16(const uint16_t* x, int M, int N, int chunks, float* partial, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && partial && M > 0 && N > 0 && chunks > 0, CMDIAD_ERR_ARG, "cmdiad_colsum_bf16: bad args");
    const int rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, M, N,
                       rpc, partial);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_ln_param_grad(const float* dh, const float* x, const float* mean, const float* rstd, int M, int C,
                                    int chunks, float* partial_g, float* partial_b, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(dh && x && mean && rstd && partial_g && partial_b && chunks > 0, CMDIAD_ERR_ARG,
                   "cmdiad_ln_param_grad: bad args");
    const int rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, dh, x, mean, rstd, M,
                       C, rpc, partial_g, partial_b);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                                float eps, int step, float grad_scale, uint16_t* p_bf16, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(p && g && m && v && step >= 1, CMDIAD_ERR_ARG, "cmdiad_adam_step: bad args");
    if (n == 0) return CMDIAD_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       bc1, bc2s, grad_scale, (bf16_t*)p_bf16);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
