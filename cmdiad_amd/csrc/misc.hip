// Bandwidth-bound helper kernels: LayerNorm (+ optional positional add), ViT patch im2col and token
// assembly, bilinear up-sampling of the score map, casts.  One wave per row where a row reduction
// is needed; 8/16-byte vector accesses everywhere (guide G13).
#include <algorithm>
#include "common.h"

namespace {

// y = LN(x (+ add)); optionally x <- x + add in place (Point-MAE: models/models.py:240 `block(x + pos)`).
// C % 128 == 0, C <= 1024: a lane holds C/128 float2 pairs at columns 2*(lane + 64 e).
template <int PAIRS>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const float* __restrict__ add,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int M, bf16_t* __restrict__ out_bf16,
                                                        float* __restrict__ out_f32, int ldo32,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out)
{
    constexpr int C = PAIRS * 128;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float2 v[PAIRS];
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < PAIRS; ++e) {
        const int c = 2 * (lane + 64 * e);
        v[e] = *reinterpret_cast<const float2*>(x + (size_t)row * C + c);
        if (add) {
            const float2 a = *reinterpret_cast<const float2*>(add + (size_t)row * C + c);
            v[e].x += a.x; v[e].y += a.y;
            *reinterpret_cast<float2*>(x + (size_t)row * C + c) = v[e];
        }
        s += v[e].x + v[e].y;
    }
    s = wave_sum(s);
    const float mean = s / C;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < PAIRS; ++e) {
        const float a = v[e].x - mean, b = v[e].y - mean;
        q += a * a + b * b;
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / C + eps);
    if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int e = 0; e < PAIRS; ++e) {
        const int c = 2 * (lane + 64 * e);
        const float2 g = *reinterpret_cast<const float2*>(gamma + c);
        const float2 bb = *reinterpret_cast<const float2*>(beta + c);
        const float y0 = (v[e].x - mean) * rstd * g.x + bb.x;
        const float y1 = (v[e].y - mean) * rstd * g.y + bb.y;
        if (out_bf16) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 o = {f2bf(y0), f2bf(y1)};
            *reinterpret_cast<bf16x2*>(out_bf16 + (size_t)row * C + c) = o;
        }
        if (out_f32) *reinterpret_cast<float2*>(out_f32 + (size_t)row * ldo32 + c) = make_float2(y0, y1);
    }
}

// The same for C % 256 == 0 (ViT-B: 768) with 16-byte accesses: a lane holds C/256 float4 at columns 4*(lane + 64 e) -- one
// wave-instruction moves 1 KiB instead of 512 B, and the bf16 row goes out in 8-byte stores (24.5 -> see profiles/r5_notes.md
// section 18).  Same two-pass statistics; the partial sums a lane forms differ from the float2 form, so the two agree to fp32
// rounding, not bitwise (every caller of a given C always takes the same one).
template <int QUADS>
__global__ __launch_bounds__(256) void layernorm4_kernel(float* __restrict__ x, const float* __restrict__ add,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int M, bf16_t* __restrict__ out_bf16,
                                                         float* __restrict__ out_f32, int ldo32,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out)
{
    constexpr int C = QUADS * 256;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float4 v[QUADS];
    float s = 0.0f;
#pragma unroll
    for (int e = 0; e < QUADS; ++e) {
        const int c = 4 * (lane + 64 * e);
        v[e] = *reinterpret_cast<const float4*>(x + (size_t)row * C + c);
        if (add) {
            const float4 a = *reinterpret_cast<const float4*>(add + (size_t)row * C + c);
            v[e].x += a.x; v[e].y += a.y; v[e].z += a.z; v[e].w += a.w;
            *reinterpret_cast<float4*>(x + (size_t)row * C + c) = v[e];
        }
        s += (v[e].x + v[e].y) + (v[e].z + v[e].w);
    }
    s = wave_sum(s);
    const float mean = s / C;
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < QUADS; ++e) {
        const float a = v[e].x - mean, b = v[e].y - mean, c2 = v[e].z - mean, d = v[e].w - mean;
        q += (a * a + b * b) + (c2 * c2 + d * d);
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / C + eps);
    if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int e = 0; e < QUADS; ++e) {
        const int c = 4 * (lane + 64 * e);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bb = *reinterpret_cast<const float4*>(beta + c);
        const float y0 = (v[e].x - mean) * rstd * g.x + bb.x, y1 = (v[e].y - mean) * rstd * g.y + bb.y;
        const float y2 = (v[e].z - mean) * rstd * g.z + bb.z, y3 = (v[e].w - mean) * rstd * g.w + bb.w;
        if (out_bf16) {
            bf16x4 o = {f2bf(y0), f2bf(y1), f2bf(y2), f2bf(y3)};
            *reinterpret_cast<bf16x4*>(out_bf16 + (size_t)row * C + c) = o;
        }
        if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * ldo32 + c) = make_float4(y0, y1, y2, y3);
    }
}

// rgb [B,3,S,S] f32 -> patches [B*(S/8)^2, 192] bf16, k = c*64 + dy*8 + dx (conv weight [768,3,8,8] flattened).
// One thread per (patch, c, dy): reads 8 contiguous floats, writes 8 contiguous bf16 (16 B).
__global__ __launch_bounds__(256) void im2col_patch8_kernel(const float* __restrict__ rgb, int B, int S,
                                                            bf16_t* __restrict__ patches)
{
    const int P = S / 8;
    const size_t total = (size_t)B * P * P * 24;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cd = (int)(i % 24);
    const size_t pi = i / 24;
    const int c = cd / 8, dy = cd % 8;
    const int px = (int)(pi % P), py = (int)((pi / P) % P), b = (int)(pi / ((size_t)P * P));
    const float* src = rgb + (((size_t)b * 3 + c) * S + (py * 8 + dy)) * S + px * 8;
    const float4 a = *reinterpret_cast<const float4*>(src), d = *reinterpret_cast<const float4*>(src + 4);
    bf16x8 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w), f2bf(d.x), f2bf(d.y), f2bf(d.z), f2bf(d.w)};
    *reinterpret_cast<bf16x8*>(patches + pi * 192 + c * 64 + dy * 8) = o;
}

// 3 x 3 im2col with padding 1 (the stem convolution of the HRNet trunk as a GEMM, hrnet.py:152: Conv2d(3, 64, 3, stride 2, padding 1)):
// img [B,C,H,W] f32 -> cols [B*Ho*Wo, ld] bf16 with column k = c * 9 + ky * 3 + kx (torch.nn.functional.unfold's order = the
// flattened conv weight [N, C, 3, 3]), columns C * 9 .. ld - 1 zero.  One thread per (output pixel, 8 columns).
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ img, int B, int C, int H, int W, int stride,
                                                        int Ho, int Wo, int ld, bf16_t* __restrict__ cols)
{
    const int chunks = ld / 8;
    const size_t total = (size_t)B * Ho * Wo * chunks;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % chunks);
    const size_t pix = i / chunks;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = ch * 8 + e;
        float v = 0.0f;
        if (k < C * 9) {
            const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
            const int y = oy * stride + ky - 1, x = ox * stride + kx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((size_t)b * C + c) * H + y) * W + x];
        }
        o[e] = f2bf(v);
    }
    *reinterpret_cast<bf16x8*>(cols + pix * ld + ch * 8) = o;
}

// tokens[b][0] = cls + pos[0]; tokens[b][1+i] = patch_out[b][i] + pos[1+i]
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, int B, int P, int C,
                                                           float* __restrict__ tokens)
{
    const int c4 = C / 4;
    const size_t total = (size_t)B * (P + 1) * c4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    const size_t row = i / c4;
    const int t = (int)(row % (P + 1));
    const size_t b = row / (P + 1);
    const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)t * C + c);
    float4 v = t == 0 ? *reinterpret_cast<const float4*>(cls + c)
                      : *reinterpret_cast<const float4*>(patch_out + (b * P + (t - 1)) * C + c);
    v.x += pe.x; v.y += pe.y; v.z += pe.z; v.w += pe.w;
    *reinterpret_cast<float4*>(tokens + row * C + c) = v;
}

// ATen upsample_bilinear2d, align_corners=False: src = max((dst + 0.5) * h/H - 0.5, 0).
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ in, int B, int h, int H,
                                                          float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * H * H) return;
    const int ox = (int)(i % H), oy = (int)((i / H) % H);
    const size_t b = i / ((size_t)H * H);
    const float scale = (float)h / (float)H;
    float sy = scale * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = scale * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < h - 1 ? 1 : 0);
    const float ly = sy - (float)y0, hy = 1.0f - ly, lx = sx - (float)x0, hx = 1.0f - lx;
    const float* p = in + b * h * h;
    out[i] = hy * (hx * p[y0 * h + x0] + lx * p[y0 * h + x1]) + ly * (hx * p[y1 * h + x0] + lx * p[y1 * h + x1]);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, size_t n4, bf16_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    bf16x4 o = {f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
    *reinterpret_cast<bf16x4*>(out + i * 4) = o;
}

// out[c][r] = in[r][c] for bf16 matrices (training path: operands of the dW / dX GEMMs). 64x64 LDS tile.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int rows, int cols,
                                                             bf16_t* __restrict__ out)
{
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(size_t)(r0 + r) * cols + c0 + c] : (bf16_t)0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < cols && r0 + r < rows) out[(size_t)(c0 + c) * rows + r0 + r] = tile[r][c];
    }
}

// out[m][n] = act(w[n].xyz . x[m] + w[n].w) for K = 3 inputs (Point-MAE pos_embed first layer,
// models/models.py:268-272).  wb [N] float4 = {w_x, w_y, w_z, bias}.  One thread per 8 outputs.
__global__ __launch_bounds__(256) void linear3_kernel(const float* __restrict__ x, const float4* __restrict__ wb, size_t M,
                                                      int N, int act, bf16_t* __restrict__ out)
{
    const int n8 = N / 8;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * n8) return;
    const size_t m = i / n8;
    const int n0 = (int)(i % n8) * 8;
    const float a = x[m * 3], b = x[m * 3 + 1], c = x[m * 3 + 2];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float4 w = wb[n0 + e];
        float v = w.x * a + w.y * b + w.z * c + w.w;
        v = act == CMDIAD_ACT_GELU ? gelu_erf(v) : (act == CMDIAD_ACT_RELU ? fmaxf(v, 0.0f) : v);
        o[e] = f2bf(v);
    }
    *reinterpret_cast<bf16x8*>(out + m * N + n0) = o;
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace


// ------------------------------------------------------------------------------------------------
// Batch-statistics BatchNorm support (the reference never calls .eval() on the extractor, models/models.py:189,195 run
// in training mode under no_grad: SURVEY F1).  Per-column first / second moments in double precision:
//   col_moments:  x [rows, C] f32 (row pitch ld)  ->  sum[c] += x[r][c],  sumsq[c] += x[r][c]^2
//   moments3:     x [rows, 3] f32                 ->  out[0..2] += (x, y, z), out[3..8] += (xx, xy, xz, yy, yz, zz)
// (conv1 is linear in the three coordinates, so the mean / variance of each of its 128 outputs follow from these nine sums.)
// ------------------------------------------------------------------------------------------------
// Block = 64 columns x 4 row lanes over rows_per_block rows; a lane keeps four of its rows in flight (one row per iteration was
// one memory round trip per row: 62 us for 25 088 x 128 values); lanes combined in a fixed order, one atomic per column and block.
static __global__ __launch_bounds__(256) void col_moments_kernel(const float* __restrict__ x, size_t rows, int C, int ld,
                                                          int rows_per_block, double* __restrict__ sum, double* __restrict__ sumsq)
{
    __shared__ double s_s[4][64], s_q[4][64];
    const size_t r0 = (size_t)blockIdx.y * rows_per_block;
    const size_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    double s = 0.0, q = 0.0;
    if (c < C) {
        size_t r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            const float v0 = x[r * ld + c], v1 = x[(r + 4) * ld + c], v2 = x[(r + 8) * ld + c], v3 = x[(r + 12) * ld + c];
            s += (double)v0; q += (double)v0 * (double)v0;
            s += (double)v1; q += (double)v1 * (double)v1;
            s += (double)v2; q += (double)v2 * (double)v2;
            s += (double)v3; q += (double)v3 * (double)v3;
        }
        for (; r < r1; r += 4) {
            const double v = (double)x[r * ld + c];
            s += v;
            q += v * v;
        }
    }
    s_s[rl][col] = s;
    s_q[rl][col] = q;
    __syncthreads();
    if (rl == 0 && c < C) {
        atomicAdd(sum + c, (s_s[0][col] + s_s[1][col]) + (s_s[2][col] + s_s[3][col]));
        atomicAdd(sumsq + c, (s_q[0][col] + s_q[1][col]) + (s_q[2][col] + s_q[3][col]));
    }
}

static __global__ __launch_bounds__(256) void moments3_kernel(const float* __restrict__ x, size_t rows, double* __restrict__ out)
{
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const double px = x[r * 3], py = x[r * 3 + 1], pz = x[r * 3 + 2];
        a[0] += px; a[1] += py; a[2] += pz;
        a[3] += px * px; a[4] += px * py; a[5] += px * pz; a[6] += py * py; a[7] += py * pz; a[8] += pz * pz;
    }
    __shared__ double sh[4][9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double v = a[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);   // (double: once per launch)
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) atomicAdd(out + threadIdx.x, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

extern "C" int cmdiad_layernorm(float* x, const float* add, const float* gamma, const float* beta, float eps, int M,
                                int C, uint16_t* out_bf16, float* out_f32, int ldo32, float* mean_out, float* rstd_out,
                                cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && gamma && beta && (out_bf16 || out_f32), CMDIAD_ERR_ARG, "cmdiad_layernorm: null pointer");
    CMDIAD_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr), CMDIAD_ERR_ARG, "cmdiad_layernorm: mean_out and rstd_out go together");
    CMDIAD_REQUIRE(M >= 0 && C % 128 == 0 && C >= 128 && C <= 1024, CMDIAD_ERR_ARG,
                   "cmdiad_layernorm: need C%%128==0, 128<=C<=1024 (C=%d)", C);
    CMDIAD_REQUIRE((((uintptr_t)x | (uintptr_t)add | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out_f32) & 7) == 0 &&
                       ((uintptr_t)out_bf16 & 3) == 0 && (!out_f32 || ldo32 % 2 == 0),
                   CMDIAD_ERR_ARG, "cmdiad_layernorm: alignment");
    if (M == 0) return CMDIAD_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
    // 16-byte form: C % 256 == 0 and every pointer it touches 16-byte (bf16 row: 8-byte) aligned; CMDIAD_LN_WIDE=0: the float2 form (A/B)
    static const bool wide_ok = !(getenv("CMDIAD_LN_WIDE") && getenv("CMDIAD_LN_WIDE")[0] == '0');
    const bool wide = wide_ok && C % 256 == 0 && C <= 1024 && (((uintptr_t)x | (uintptr_t)add | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out_f32) & 15) == 0 &&
                      ((uintptr_t)out_bf16 & 7) == 0 && (!out_f32 || ldo32 % 4 == 0);
    if (wide) {
#define LN4_CASE(Q) case Q: hipLaunchKernelGGL(layernorm4_kernel<Q>, grid, block, 0, s, x, add, gamma, beta, eps, M, (bf16_t*)out_bf16, out_f32, ldo32, mean_out, rstd_out); break;
        switch (C / 256) { LN4_CASE(1) LN4_CASE(2) LN4_CASE(3) LN4_CASE(4) }
#undef LN4_CASE
        CMDIAD_CHECK_LAUNCH();
        return CMDIAD_OK;
    }
#define LN_CASE(P) case P: hipLaunchKernelGGL(layernorm_kernel<P>, grid, block, 0, s, x, add, gamma, beta, eps, M, (bf16_t*)out_bf16, out_f32, ldo32, mean_out, rstd_out); break;
    switch (C / 128) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(7) LN_CASE(8)
    }
#undef LN_CASE
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_im2col_patch8(const float* rgb, int B, int S, uint16_t* patches, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(rgb && patches && B > 0 && S > 0 && S % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_im2col_patch8: bad args");
    CMDIAD_REQUIRE((((uintptr_t)rgb | (uintptr_t)patches) & 15) == 0, CMDIAD_ERR_ARG, "cmdiad_im2col_patch8: alignment");
    const size_t total = (size_t)B * (S / 8) * (S / 8) * 24;
    hipLaunchKernelGGL(im2col_patch8_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, rgb, B, S,
                       (bf16_t*)patches);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_im2col3x3_bf16(const float* img, int B, int C, int H, int W, int stride, int ld, uint16_t* cols, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(img && cols && B > 0 && C > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && ld % 8 == 0 && ld >= C * 9 &&
                       ((uintptr_t)cols & 15) == 0, CMDIAD_ERR_ARG, "cmdiad_im2col3x3_bf16: bad args (need ld %% 8 == 0, ld >= 9 C, stride 1 | 2)");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t total = (size_t)B * Ho * Wo * (ld / 8);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, img, B, C, H, W, stride, Ho, Wo, ld,
                       (bf16_t*)cols);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_vit_assemble(const float* patch_out, const float* cls, const float* pos, int B, int P, int C,
                                   float* tokens, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(patch_out && cls && pos && tokens && B > 0 && P > 0 && C % 4 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_vit_assemble: bad args");
    const size_t total = (size_t)B * (P + 1) * (C / 4);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, patch_out, cls, pos,
                       B, P, C, tokens);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_bilinear_up(const float* in, int B, int h, int H, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(in && out && B > 0 && h > 0 && H > 0, CMDIAD_ERR_ARG, "cmdiad_bilinear_up: bad args");
    hipLaunchKernelGGL(bilinear_up_kernel, dim3(blocks_for((size_t)B * H * H)), dim3(256), 0, (hipStream_t)stream, in, B, h,
                       H, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_linear3(const float* x, const float* wb, size_t M, int N, int act, uint16_t* out,
                              cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && wb && out && N % 8 == 0, CMDIAD_ERR_ARG, "cmdiad_linear3: N%%8==0 required");
    CMDIAD_REQUIRE((((uintptr_t)wb | (uintptr_t)out) & 15) == 0, CMDIAD_ERR_ARG, "cmdiad_linear3: alignment");
    if (M == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(linear3_kernel, dim3(blocks_for(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, x,
                       (const float4*)wb, M, N, act, (bf16_t*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_cast_bf16(const float* x, size_t n, uint16_t* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && out && n % 4 == 0, CMDIAD_ERR_ARG, "cmdiad_cast_bf16: n%%4==0 required");
    if (n == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, n / 4, (bf16_t*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_transpose_bf16(const uint16_t* in, int rows, int cols, uint16_t* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(in && out && rows > 0 && cols > 0, CMDIAD_ERR_ARG, "cmdiad_transpose_bf16: bad args");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, rows, cols, (bf16_t*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_col_moments(const float* x, size_t rows, int C, int ld, double* sum, double* sumsq, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && sum && sumsq && C > 0 && ld >= C, CMDIAD_ERR_ARG, "cmdiad_col_moments: null pointer / C / ld");
    if (rows == 0) return CMDIAD_OK;
    const int per = 256;
    hipLaunchKernelGGL(col_moments_kernel, dim3((C + 63) / 64, (unsigned)((rows + per - 1) / per)), dim3(256), 0, (hipStream_t)stream,
                       x, rows, C, ld, per, sum, sumsq);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_moments3(const float* xyz, size_t rows, double* out9, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(xyz && out9, CMDIAD_ERR_ARG, "cmdiad_moments3: null pointer");
    if (rows == 0) return CMDIAD_OK;
    const unsigned blocks = (unsigned)((rows + 255) / 256 < 1024 ? (rows + 255) / 256 : 1024);
    hipLaunchKernelGGL(moments3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, xyz, rows, out9);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
