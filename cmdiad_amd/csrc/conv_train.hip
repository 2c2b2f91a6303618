// Training-side kernels of the convolutional feature-to-feature head (reference models/hallucination_network.py:72-147:
// per direction conv3x3 -> BatchNorm2d -> ReLU three times, then conv3x3; trained by hallucination_network_pretrain.py:106-147
// with the module in train() mode, i.e. BatchNorm on the statistics of the batch).  The convolutions -- forward, data gradient
// (the same kernel on flipped, transposed weights) -- are cmdiad_conv2d_nhwc_bf16; the weight gradients are nine
// cmdiad_gemm_tn_bf16 products over zero-bordered copies (one per filter tap: the tap's shift is a row offset there); this file
// holds the bandwidth-bound pieces between them, on NHWC activations flattened to [M = B*H*W, C]:
//   * BatchNorm (batch statistics) + ReLU forward from the fp32 convolution output to the next convolution's bf16 operand
//   * BatchNorm + ReLU backward: column sums of g and g * xhat (g = dY where the ReLU was open), then
//     dz = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) as the previous convolution's bf16 output gradient
//   * the zero-bordered copy [B,H,W,C] -> [B,H+2,W+2,C]
// and, for the feature-to-input convolutional head (hallucination_network.py:185-220: conv, bicubic x4, conv + ReLU twice, conv),
//   * ReLU backward from the saved bf16 output, and the adjoint of the bicubic upsampling (two one-axis gathering passes)
#include "common.h"

namespace {

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// y = z * scale[c] + shift[c] (+ residual) (then ReLU if relu) as bf16 and / or f32; scale = gamma * rstd, shift = beta - mean * scale.
// C % 8 == 0.  (Bottleneck.forward, hrnet.py:23-43: bn3 has no ReLU of its own -- the residual is added first.)
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ residual, int relu,
                                                          size_t n8, int C8, bf16_t* __restrict__ y, float* __restrict__ y32)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c = (int)(i % C8) * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(z + i * 8), a1 = *reinterpret_cast<const float4*>(z + i * 8 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(shift + c), t1 = *reinterpret_cast<const float4*>(shift + c + 4);
    float v[8] = {fmaf(a0.x, s0.x, t0.x), fmaf(a0.y, s0.y, t0.y), fmaf(a0.z, s0.z, t0.z), fmaf(a0.w, s0.w, t0.w),
                  fmaf(a1.x, s1.x, t1.x), fmaf(a1.y, s1.y, t1.y), fmaf(a1.z, s1.z, t1.z), fmaf(a1.w, s1.w, t1.w)};
    if (residual) {
        const float4 r0 = *reinterpret_cast<const float4*>(residual + i * 8), r1 = *reinterpret_cast<const float4*>(residual + i * 8 + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    if (y) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x8*>(y + i * 8) = o;
    }
    if (y32) {
        *reinterpret_cast<float4*>(y32 + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(y32 + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// BatchNorm constants of one layer from the double-precision column sums of cmdiad_col_moments: mean, biased variance (as float64,
// for the running-statistics update), and scale = gamma * rstd, shift = beta - mean * scale, mean, rstd as f32.
__global__ __launch_bounds__(256) void bn_affine_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, double rows,
                                                        double eps, int C, double* __restrict__ mean64, double* __restrict__ var64,
                                                        float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean,
                                                        float* __restrict__ rstd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double m = sum[c] / rows;
    const double v = sumsq[c] / rows - m * m;
    const double inv = 1.0 / sqrt(v + eps);
    const float sc = (float)((double)gamma[c] * inv);
    mean64[c] = m; var64[c] = v;
    scale[c] = sc;
    shift[c] = (float)((double)beta[c] - m * (double)sc);
    mean[c] = (float)m;
    rstd[c] = (float)inv;
}

// Column partial sums over a chunk of rows: p1[chunk][c] = sum g, p2[chunk][c] = sum g * xhat with
// g = dy where z * scale + shift > 0 (the ReLU was open; masked) else 0, or g = dy (not masked), xhat = (z - mean) * rstd.
// Block = 64 columns x 4 row lanes (a 128-channel layer still fills two column blocks per chunk), fixed row order per lane and a
// fixed lane order at the end: bit-reproducible.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, int masked,
                                                            size_t M, int C, size_t rows_per_chunk, float* __restrict__ p1,
                                                            float* __restrict__ p2)
{
    __shared__ float s_a[4][64], s_b[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const size_t r0 = (size_t)blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, M);
    float s1 = 0.0f, s2 = 0.0f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
        size_t r = r0 + rl;
        for (; r + 12 < r1; r += 16) {   // four rows of this lane in flight (one row per iteration was a memory round trip per row)
            float zz[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { zz[u] = z[(r + 4 * u) * C + c]; dd[u] = dy[(r + 4 * u) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float g = (!masked || fmaf(zz[u], sc, sh) > 0.0f) ? dd[u] : 0.0f;   // masked: the layer's own ReLU
                s1 += g;
                s2 = fmaf(g, (zz[u] - mu) * rs, s2);
            }
        }
        for (; r < r1; r += 4) {
            const float zz = z[r * C + c];
            const float g = (!masked || fmaf(zz, sc, sh) > 0.0f) ? dy[r * C + c] : 0.0f;
            s1 += g;
            s2 = fmaf(g, (zz - mu) * rs, s2);
        }
    }
    s_a[rl][threadIdx.x & 63] = s1;
    s_b[rl][threadIdx.x & 63] = s2;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        const int l = threadIdx.x;
        p1[(size_t)blockIdx.y * C + c] = (s_a[0][l] + s_a[1][l]) + (s_a[2][l] + s_a[3][l]);
        p2[(size_t)blockIdx.y * C + c] = (s_b[0][l] + s_b[1][l]) + (s_b[2][l] + s_b[3][l]);
    }
}

// dbeta[c] = sum_s p1[s][c], dgamma[c] = sum_s p2[s][c] over S chunk partials: 64 columns x 4 slab lanes per block, fixed order.
__global__ __launch_bounds__(256) void bn_partials_sum_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int S, int C,
                                                              float* __restrict__ dbeta, float* __restrict__ dgamma)
{
    __shared__ float s_a[4][64], s_b[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float a = 0.0f, b = 0.0f;
    if (c < C)
        for (int s = sl; s < S; s += 4) { a += p1[(size_t)s * C + c]; b += p2[(size_t)s * C + c]; }
    s_a[sl][threadIdx.x & 63] = a;
    s_b[sl][threadIdx.x & 63] = b;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        const int l = threadIdx.x;
        dbeta[c] = (s_a[0][l] + s_a[1][l]) + (s_a[2][l] + s_a[3][l]);
        dgamma[c] = (s_b[0][l] + s_b[1][l]) + (s_b[2][l] + s_b[3][l]);
    }
}

// dz = scale * (g - dbeta / M - xhat * dgamma / M) as bf16 (scale = gamma * rstd; dbeta = sum g, dgamma = sum g * xhat)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                           int masked, float inv_m, size_t n4, int C4, bf16_t* __restrict__ dz)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const float4 g4 = *reinterpret_cast<const float4*>(dy + i * 4), z4 = *reinterpret_cast<const float4*>(z + i * 4);
    const float4 sc4 = *reinterpret_cast<const float4*>(scale + c), sh4 = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu4 = *reinterpret_cast<const float4*>(mean + c), rs4 = *reinterpret_cast<const float4*>(rstd + c);
    const float4 db4 = *reinterpret_cast<const float4*>(dbeta + c), dg4 = *reinterpret_cast<const float4*>(dgamma + c);
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
    const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
    const float rsv[4] = {rs4.x, rs4.y, rs4.z, rs4.w}, dbv[4] = {db4.x, db4.y, db4.z, db4.w}, dgv[4] = {dg4.x, dg4.y, dg4.z, dg4.w};
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g = (!masked || fmaf(zv[e], scv[e], shv[e]) > 0.0f) ? gv[e] : 0.0f;
        const float xh = (zv[e] - muv[e]) * rsv[e];
        o[e] = f2bf(scv[e] * (g - dbv[e] * inv_m - xh * dgv[e] * inv_m));
    }
    *reinterpret_cast<bf16x4*>(dz + i * 4) = o;
}

// interior copy of x [B,H,W,C] into out [B,H+2,W+2,C] (the border stays as the caller zeroed it).  C % 8 == 0.
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C8,
                                                       bf16_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * H * W * C8;
    if (i >= total) return;
    const int c = (int)(i % C8);
    size_t p = i / C8;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    const size_t o = (((size_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C8 + c;
    reinterpret_cast<bf16x8*>(out)[o] = reinterpret_cast<const bf16x8*>(x)[i];
}

// dz = dx where the saved ReLU output y is positive, else 0 (bias + ReLU layers: y > 0 <=> the pre-activation was), as bf16.
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dx, const bf16_t* __restrict__ y, size_t n4,
                                                       bf16_t* __restrict__ dz, float* __restrict__ dz32)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = *reinterpret_cast<const float4*>(dx + i * 4);
    const bf16x4 yy = *reinterpret_cast<const bf16x4*>(y + i * 4);
    const float4 m = make_float4(bf2f(yy[0]) > 0.f ? g.x : 0.f, bf2f(yy[1]) > 0.f ? g.y : 0.f, bf2f(yy[2]) > 0.f ? g.z : 0.f,
                                 bf2f(yy[3]) > 0.f ? g.w : 0.f);
    if (dz) *reinterpret_cast<bf16x4*>(dz + i * 4) = bf16x4{f2bf(m.x), f2bf(m.y), f2bf(m.z), f2bf(m.w)};
    if (dz32) *reinterpret_cast<float4*>(dz32 + i * 4) = m;
}

// torch's cubic convolution weights (A = -0.75), as in conv.hip
__device__ __forceinline__ void cubic_weights_t(float t, float (&w)[4])
{
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// Adjoint of bicubic interpolation (align_corners=False, border taps clamped: F.interpolate(mode='bicubic') backward) along ONE
// axis: g [outer][L_out][inner] -> out [outer][L_in][inner],  out[l] = sum over the output positions L whose four taps
// (clamped) include l of weight * g[L].  Gathering form (no atomics; fixed order: bit-reproducible); the 2-D adjoint is two passes.
__global__ __launch_bounds__(256) void cubic_adjoint_axis_kernel(const float* __restrict__ g, size_t outer, int L_out, int L_in,
                                                                 size_t inner4, float* __restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= outer * L_in * inner4) return;
    const size_t in4 = t % inner4;
    const int l = (int)((t / inner4) % L_in);
    const size_t o = t / (inner4 * L_in);
    const float scale = (float)L_in / (float)L_out;
    // output positions that can touch l: src = scale * (L + 0.5) - 0.5 in [l - 2, l + 2) (+ the clamped border taps, which lie inside)
    int L0 = (int)floorf(((float)l - 2.0f + 0.5f) / scale - 0.5f) - 1, L1 = (int)ceilf(((float)l + 2.0f + 0.5f) / scale - 0.5f) + 1;
    L0 = max(L0, 0); L1 = min(L1, L_out - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* src = reinterpret_cast<const float4*>(g) + (o * L_out) * inner4 + in4;
    for (int L = L0; L <= L1; ++L) {
        const float f = scale * ((float)L + 0.5f) - 0.5f;
        const float ff = floorf(f);
        float w[4];
        cubic_weights_t(f - ff, w);
        const int i0 = (int)ff - 1;
        float c = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (min(max(i0 + a, 0), L_in - 1) == l) c += w[a];
        if (c != 0.0f) {
            const float4 v = src[(size_t)L * inner4];
            acc.x = fmaf(c, v.x, acc.x); acc.y = fmaf(c, v.y, acc.y); acc.z = fmaf(c, v.z, acc.z); acc.w = fmaf(c, v.w, acc.w);
        }
    }
    reinterpret_cast<float4*>(out)[(o * L_in + l) * inner4 + in4] = acc;
}

}  // namespace

extern "C" int cmdiad_relu_bwd_bf16(const float* dx, const uint16_t* y, size_t n, uint16_t* dz, float* dz_f32, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(dx && y && (dz || dz_f32) && n % 4 == 0, CMDIAD_ERR_ARG, "cmdiad_relu_bwd_bf16: n%%4==0 and an output required");
    CMDIAD_REQUIRE((((uintptr_t)dx | (uintptr_t)dz_f32) & 15) == 0 && (((uintptr_t)y | (uintptr_t)dz) & 7) == 0, CMDIAD_ERR_ARG,
                   "cmdiad_relu_bwd_bf16: alignment");
    if (n == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dx, (const bf16_t*)y, n / 4, (bf16_t*)dz,
                       dz_f32);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_upsample_bicubic_bwd(const float* grad_out, int B, int H, int W, int C, int h, int w, float* tmp, float* grad_in,
                                           cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(grad_out && tmp && grad_in && B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_upsample_bicubic_bwd: C%%4==0 required");
    CMDIAD_REQUIRE((((uintptr_t)grad_out | (uintptr_t)tmp | (uintptr_t)grad_in) & 15) == 0, CMDIAD_ERR_ARG, "cmdiad_upsample_bicubic_bwd: alignment");
    hipStream_t s = (hipStream_t)stream;
    // pass 1 along x: [B*H][W][C] -> tmp [B*H][w][C];  pass 2 along y: [B][H][w*C] -> grad_in [B][h][w*C]
    const size_t n1 = (size_t)B * H * w * (C / 4), n2 = (size_t)B * h * w * (C / 4);
    hipLaunchKernelGGL(cubic_adjoint_axis_kernel, dim3(blocks_for(n1)), dim3(256), 0, s, grad_out, (size_t)B * H, W, w, (size_t)(C / 4), tmp);
    hipLaunchKernelGGL(cubic_adjoint_axis_kernel, dim3(blocks_for(n2)), dim3(256), 0, s, tmp, (size_t)B, H, h, (size_t)w * (C / 4), grad_in);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_affine(const double* sum, const double* sumsq, const float* gamma, const float* beta, size_t rows, double eps,
                                int C, double* mean64, double* var64, float* scale, float* shift, float* mean, float* rstd,
                                cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(sum && sumsq && gamma && beta && mean64 && var64 && scale && shift && mean && rstd && rows > 0 && C > 0, CMDIAD_ERR_ARG,
                   "cmdiad_bn_affine: bad args");
    hipLaunchKernelGGL(bn_affine_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sum, sumsq, gamma, beta,
                       (double)rows, eps, C, mean64, var64, scale, shift, mean, rstd);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_relu_fwd(const float* z, const float* scale, const float* shift, const float* residual, int relu, size_t M,
                                  int C, uint16_t* y, float* y_f32, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z && scale && shift && (y || y_f32) && C > 0 && C % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_bn_relu_fwd: C%%8==0 required (C=%d)", C);
    CMDIAD_REQUIRE((((uintptr_t)z | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)y | (uintptr_t)y_f32 | (uintptr_t)residual) & 15) == 0,
                   CMDIAD_ERR_ARG, "cmdiad_bn_relu_fwd: alignment");
    if (M == 0) return CMDIAD_OK;
    const size_t n8 = M * (size_t)(C / 8);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(blocks_for(n8)), dim3(256), 0, (hipStream_t)stream, z, scale, shift, residual, relu, n8,
                       C / 8, (bf16_t*)y, y_f32);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_relu_bwd_reduce(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                                         const float* rstd, int masked, size_t M, int C, int chunks, float* part_dbeta,
                                         float* part_dgamma, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(dy && z && scale && shift && mean && rstd && part_dbeta && part_dgamma && M > 0 && C > 0 && chunks > 0, CMDIAD_ERR_ARG,
                   "cmdiad_bn_relu_bwd_reduce: bad args");
    const size_t rows = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, dy, z,
                       scale, shift, mean, rstd, masked, M, C, rows, part_dbeta, part_dgamma);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_partials_sum(const float* part_dbeta, const float* part_dgamma, int chunks, int C, float* dbeta, float* dgamma,
                                      cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(part_dbeta && part_dgamma && dbeta && dgamma && chunks > 0 && C > 0, CMDIAD_ERR_ARG, "cmdiad_bn_partials_sum: bad args");
    hipLaunchKernelGGL(bn_partials_sum_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, part_dbeta, part_dgamma,
                       chunks, C, dbeta, dgamma);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_relu_bwd_apply(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, const float* dbeta, const float* dgamma, int masked, size_t M, int C,
                                        uint16_t* dz, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(dy && z && scale && shift && mean && rstd && dbeta && dgamma && dz && M > 0 && C > 0 && C % 4 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_bn_relu_bwd_apply: C%%4==0 required (C=%d)", C);
    CMDIAD_REQUIRE((((uintptr_t)dy | (uintptr_t)z | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)dbeta |
                     (uintptr_t)dgamma) & 15) == 0 && ((uintptr_t)dz & 7) == 0, CMDIAD_ERR_ARG, "cmdiad_bn_relu_bwd_apply: alignment");
    const size_t n4 = M * (size_t)(C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, dy, z, scale, shift, mean, rstd,
                       dbeta, dgamma, masked, 1.0f / (float)M, n4, C / 4, (bf16_t*)dz);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_pad_nhwc_bf16(const uint16_t* x, int B, int H, int W, int C, uint16_t* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_pad_nhwc_bf16: C%%8==0 required");
    CMDIAD_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 15) == 0, CMDIAD_ERR_ARG, "cmdiad_pad_nhwc_bf16: alignment");
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(pad_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, B, H, W, C / 8,
                       (bf16_t*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
