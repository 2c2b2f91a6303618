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
#include "common.h"

namespace {

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// y = relu(z * scale[c] + shift[c]) as bf16; scale = gamma * rstd, shift = beta - mean * scale.  C % 8 == 0.
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, size_t n8, int C8, bf16_t* __restrict__ y)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c = (int)(i % C8) * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(z + i * 8), a1 = *reinterpret_cast<const float4*>(z + i * 8 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(shift + c), t1 = *reinterpret_cast<const float4*>(shift + c + 4);
    const float v[8] = {fmaf(a0.x, s0.x, t0.x), fmaf(a0.y, s0.y, t0.y), fmaf(a0.z, s0.z, t0.z), fmaf(a0.w, s0.w, t0.w),
                        fmaf(a1.x, s1.x, t1.x), fmaf(a1.y, s1.y, t1.y), fmaf(a1.z, s1.z, t1.z), fmaf(a1.w, s1.w, t1.w)};
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaxf(v[e], 0.0f));
    *reinterpret_cast<bf16x8*>(y + i * 8) = o;
}

// Column partial sums over a chunk of rows: p1[chunk][c] = sum g, p2[chunk][c] = sum g * xhat with
// g = dy where z * scale + shift > 0 (the ReLU was open) else 0, xhat = (z - mean) * rstd.  One thread per column, consecutive
// threads on consecutive columns (coalesced rows), fixed row order: bit-reproducible.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, size_t M,
                                                            int C, size_t rows_per_chunk, float* __restrict__ p1, float* __restrict__ p2)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const size_t r0 = (size_t)blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, M);
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    float s1 = 0.0f, s2 = 0.0f;
    for (size_t r = r0; r < r1; ++r) {
        const float zz = z[r * C + c];
        const float g = fmaf(zz, sc, sh) > 0.0f ? dy[r * C + c] : 0.0f;
        s1 += g;
        s2 = fmaf(g, (zz - mu) * rs, s2);
    }
    p1[(size_t)blockIdx.y * C + c] = s1;
    p2[(size_t)blockIdx.y * C + c] = s2;
}

// dz = scale * (g - dbeta / M - xhat * dgamma / M) as bf16 (scale = gamma * rstd; dbeta = sum g, dgamma = sum g * xhat)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                           float inv_m, size_t n4, int C4, bf16_t* __restrict__ dz)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % C4) * 4;
    const float4 g4 = *reinterpret_cast<const float4*>(dy + i * 4), z4 = *reinterpret_cast<const float4*>(z + i * 4);
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float sc = scale[c + e];
        const float g = fmaf(zv[e], sc, shift[c + e]) > 0.0f ? gv[e] : 0.0f;
        const float xh = (zv[e] - mean[c + e]) * rstd[c + e];
        o[e] = f2bf(sc * (g - dbeta[c + e] * inv_m - xh * dgamma[c + e] * inv_m));
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

}  // namespace

extern "C" int cmdiad_bn_relu_fwd(const float* z, const float* scale, const float* shift, size_t M, int C, uint16_t* y,
                                  cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(z && scale && shift && y && C > 0 && C % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_bn_relu_fwd: C%%8==0 required (C=%d)", C);
    CMDIAD_REQUIRE((((uintptr_t)z | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)y) & 15) == 0, CMDIAD_ERR_ARG, "cmdiad_bn_relu_fwd: alignment");
    if (M == 0) return CMDIAD_OK;
    const size_t n8 = M * (size_t)(C / 8);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(blocks_for(n8)), dim3(256), 0, (hipStream_t)stream, z, scale, shift, n8, C / 8, (bf16_t*)y);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_relu_bwd_reduce(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                                         const float* rstd, size_t M, int C, int chunks, float* part_dbeta, float* part_dgamma,
                                         cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(dy && z && scale && shift && mean && rstd && part_dbeta && part_dgamma && M > 0 && C > 0 && chunks > 0, CMDIAD_ERR_ARG,
                   "cmdiad_bn_relu_bwd_reduce: bad args");
    const size_t rows = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, dy, z,
                       scale, shift, mean, rstd, M, C, rows, part_dbeta, part_dgamma);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bn_relu_bwd_apply(const float* dy, const float* z, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, const float* dbeta, const float* dgamma, size_t M, int C, uint16_t* dz,
                                        cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(dy && z && scale && shift && mean && rstd && dbeta && dgamma && dz && M > 0 && C > 0 && C % 4 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_bn_relu_bwd_apply: C%%4==0 required (C=%d)", C);
    CMDIAD_REQUIRE((((uintptr_t)dy | (uintptr_t)z) & 15) == 0 && ((uintptr_t)dz & 7) == 0, CMDIAD_ERR_ARG, "cmdiad_bn_relu_bwd_apply: alignment");
    const size_t n4 = M * (size_t)(C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, dy, z, scale, shift, mean, rstd,
                       dbeta, dgamma, 1.0f / (float)M, n4, C / 4, (bf16_t*)dz);
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
