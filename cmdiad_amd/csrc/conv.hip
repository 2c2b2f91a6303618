// Convolution heads of the distillation networks: 3x3 / 1x1 convolutions on NHWC bf16 activations
// as an IMPLICIT GEMM on the gemm_core pipeline, the 3 -> 64 stem convolution of the HRNet head as a direct kernel, and
// torch's bicubic 56 -> 224 upsampling.
//
//   reference: models/hallucination_network.py:72-143 (HallucinationCrossModalityConv: 4 x [3x3, 768 -> 768] + BN + ReLU),
//              models/hallucination_network.py:185-220 (HallucinationFeatureToInputConv), models/hrnet.py:8-43,146-290.
//
// Token matrices [B, 3136, C] ARE NHWC images [B, 56, 56, C] (hallucination_network.py:6-15 only reshapes), so a
// convolution is out[m, o] = sum_{tap, c} x[pixel(m) + tap][c] . w[o][tap][c]: the A operand of the GEMM is gathered
// per K-step from the shifted pixel -- K-step kt covers channels c0..c0+63 of ONE tap (C % 64 == 0), so the LDS-DMA
// source of a tile row is just another 128-byte run; rows whose tap falls into the zero padding read a zero line.
// Weights are packed by the host as [Cout][tap][C] bf16 (K-contiguous, like nn.Linear), BatchNorm folded.
#include <mutex>
#include <set>

#include "gemm_core.h"

namespace {

using namespace gemm;

__device__ __attribute__((aligned(128))) bf16_t g_zero_line[64];  // zero-initialised: the padding pixels' source

struct ConvGeom {
    int B, H, W, C;      // input  [B,H,W,C]
    int Ho, Wo;          // output [B,Ho,Wo,N]
    int ks, stride, pad; // 3/1/1, 3/2/1 or 1/1/0
    int kt_per_tap, magic;  // C/64; tap = (kt * magic) >> 16 (checked on the host for every kt of the launch)
};

// Per-thread A stager: the tile rows a thread feeds are fixed for the whole block, so their pixel coordinates are
// decoded once (two integer divisions per row) and every K-step only adds the tap offset and tests the borders.
template <int PIECES>
struct ConvTile {
    const bf16_t* x;
    ConvGeom g;
    int pix[PIECES];  // pixel index (b*H + y*stride)*W + x*stride of the window centre (before -pad)
    int yx[PIECES];   // (y*stride) << 16 | (x*stride)

    __device__ __forceinline__ void init(int m0, int M, int tid, int rows_per_wave)
    {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int m = min(m0 + wave * rows_per_wave + j * 8 + (lane >> 3), M - 1);
            const int b = m / (g.Ho * g.Wo), rem = m - b * (g.Ho * g.Wo);
            const int y = rem / g.Wo, xo = rem - y * g.Wo;
            pix[j] = (b * g.H + y * g.stride) * g.W + xo * g.stride;
            yx[j] = ((y * g.stride) << 16) | (xo * g.stride);
        }
    }

    template <int ROWS, int WAVES>
    __device__ __forceinline__ void stage(char* tile, int, int k0, int tid) const
    {
        static_assert(ROWS / WAVES / 8 == PIECES, "tile rows per wave");
        const int lane = tid & 63, wave = tid >> 6;
        const int kt = k0 >> 6;
        const int tap = (kt * g.magic) >> 16;
        const int c0 = (kt - tap * g.kt_per_tap) << 6;
        const int ty = (tap * 11) >> 5;  // tap / 3 for tap < 9
        const int dy = ty - g.pad, dx = tap - ty * 3 - g.pad;
        const int shift = dy * g.W + dx;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int r = wave * (ROWS / WAVES) + j * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ (r & 7);
            const int yy = (yx[j] >> 16) + dy, xx = (yx[j] & 0xffff) + dx;
            const bool inside = (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
            const bf16_t* src = inside ? x + (size_t)(pix[j] + shift) * g.C + c0 + chunk * 8 : g_zero_line + chunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + (wave * (ROWS / WAVES) + j * 8) * (BK * 2)),
                                             16, 0, 0);
        }
    }
};

struct ConvParams {
    const bf16_t* x;
    ConvGeom g;
    int M, N, K;
    const float* bias;
    const float* residual; int ldr;
    float* out_f32; int ldo32;
    bf16_t* out_bf16; int ldo16;
};

// ACT: CMDIAD_ACT_NONE, CMDIAD_ACT_RELU (before the residual, as cmdiad_gemm_bf16), CMDIAD_ACT_RELU_POST (after it: the
// Bottleneck's relu(bn3(conv3) + residual), hrnet.py:39-41)
template <class S, int ACT>
__global__ __launch_bounds__(S::THREADS, S::WAVES_PER_SIMD) void conv_igemm_kernel(GlobalTile W, ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int n_tiles_n = (p.N + S::BN - 1) / S::BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / n_tiles_n) * S::BM, nt = wg % n_tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / S::WN, wc = wave % S::WN;

    ConvTile<S::BM / S::WAVES / 8> A;
    A.x = p.x;
    A.g = p.g;
    A.init(m0, p.M, threadIdx.x, S::BM / S::WAVES);

    run<S, true>(A, W, m0, nt, 1, p.K / BK, lds, [&](auto& acc, int ntile, char*) {
#pragma unroll
        for (int i = 0; i < S::MI; ++i) {
            const int m = m0 + wr * (S::MI * 16) + i * 16 + (lane & 15);
            if (m >= p.M) continue;
            const float* res = p.residual ? p.residual + (size_t)m * p.ldr : nullptr;
            float* o32 = p.out_f32 ? p.out_f32 + (size_t)m * p.ldo32 : nullptr;
            bf16_t* o16 = p.out_bf16 ? p.out_bf16 + (size_t)m * p.ldo16 : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ntile * S::BN + wc * 64 + j * 16 + (lane >> 4) * 4;
                if (n >= p.N) continue;
                f32x4 v = acc[i][j];
                if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if constexpr (ACT == CMDIAD_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
                if (res) { const float4 b = *reinterpret_cast<const float4*>(res + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if constexpr (ACT == CMDIAD_ACT_RELU_POST) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
                if (o32) *reinterpret_cast<f32x4*>(o32 + n) = v;
                if (o16) {
                    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    *reinterpret_cast<bf16x4*>(o16 + n) = o;
                }
            }
        }
    });
}

template <class Kern>
int launch_conv(Kern kernel, unsigned blocks, hipStream_t s, const GlobalTile& W, const ConvParams& p)
{
    static std::mutex mu;
    static std::set<const void*> configured;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!configured.count((const void*)kernel)) {
            if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S128::LDS_BYTES) != hipSuccess) {
                cmdiad_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d) failed", S128::LDS_BYTES);
                return CMDIAD_ERR_LAUNCH;
            }
            configured.insert((const void*)kernel);
        }
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(S128::THREADS), S128::LDS_BYTES, s, W, p);
    return CMDIAD_OK;
}

// ------------------------------------------------------------------------------------------------
// Stem: 3x3 convolution of an f32 NCHW image with Cin <= 4 input planes (hrnet.py:150: 3 -> 64, stride 2), BatchNorm
// folded, ReLU, bf16 NHWC output.  27 multiply-adds per output value: a direct kernel, one thread per (pixel, 8 channels).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int Cin, int H, int W,
                                                        int Ho, int Wo, int Cout, int stride, bf16_t* __restrict__ out)
{
    extern __shared__ float s_w[];  // [Cout][Cin*9] then bias [Cout]
    for (int i = threadIdx.x; i < Cout * Cin * 9 + Cout; i += blockDim.x) s_w[i] = i < Cout * Cin * 9 ? w[i] : bias[i - Cout * Cin * 9];
    __syncthreads();
    const int groups = Cout / 8;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * Ho * Wo * groups) return;
    const int cg = (int)(t % groups);
    const long pixel = t / groups;
    const int xo = (int)(pixel % Wo), yo = (int)((pixel / Wo) % Ho), b = (int)(pixel / ((long)Wo * Ho));
    float in[4 * 9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = yo * stride + ky - 1, xx = xo * stride + kx - 1;
                in[c * 9 + ky * 3 + kx] = (c < Cin && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                                              ? x[(((size_t)b * Cin + c) * H + yy) * W + xx] : 0.0f;
            }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = cg * 8 + e;
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < Cin) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += in[c * 9 + k] * s_w[(co * Cin + c) * 9 + k];
            }
        o[e] = f2bf(fmaxf(acc + s_w[Cout * Cin * 9 + co], 0.0f));
    }
    *reinterpret_cast<bf16x8*>(out + (size_t)pixel * Cout + cg * 8) = o;
}

// ------------------------------------------------------------------------------------------------
// torch.nn.functional.interpolate(mode='bicubic', align_corners=False) (hallucination_network.py:171,204): cubic
// convolution with A = -0.75 on the 4 x 4 neighbourhood around src = (dst + 0.5) * in/out - 0.5, indices clamped to
// the image.  Input f32 NHWC [B,h,w,ldi] (C channels used); output bf16 NHWC [B,H,W,ldo] (the next convolution's
// operand) or f32 NCHW [B,C,H,W] (the hallucinated image / point map).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_weights(float t, float (&w)[4])
{
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, x2 = 1.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

template <bool NCHW_OUT>
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ in, int B, int h, int w, int C, int ldi,
                                                      int H, int W, bf16_t* __restrict__ out16, int ldo, float* __restrict__ out32)
{
    const int cgroups = (C + 3) / 4;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * H * W * cgroups) return;
    const int cg = (int)(t % cgroups);
    const long pixel = t / cgroups;
    const int X = (int)(pixel % W), Y = (int)((pixel / W) % H), b = (int)(pixel / ((long)W * H));
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const float fy = sy * ((float)Y + 0.5f) - 0.5f, fx = sx * ((float)X + 0.5f) - 0.5f;
    const float yf = floorf(fy), xf = floorf(fx);
    float wy[4], wx[4];
    cubic_weights(fy - yf, wy);
    cubic_weights(fx - xf, wx);
    const int iy = (int)yf, ix = (int)xf;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int c = cg * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), h - 1);
        float row[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xx = min(max(ix - 1 + e, 0), w - 1);
            const float* px = in + (((size_t)b * h + yy) * w + xx) * ldi + c;
            if (c + 3 < C) {
                const float4 v = *reinterpret_cast<const float4*>(px);
                row[0] += v.x * wx[e]; row[1] += v.y * wx[e]; row[2] += v.z * wx[e]; row[3] += v.w * wx[e];
            } else {
                for (int q = 0; q < 4 && c + q < C; ++q) row[q] += px[q] * wx[e];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += row[q] * wy[a];
    }
    if constexpr (NCHW_OUT) {
        for (int q = 0; q < 4 && c + q < C; ++q) out32[(((size_t)b * C + c + q) * H + Y) * W + X] = acc[q];
    } else {
        if (c + 3 < C) {
            bf16x4 o = {f2bf(acc[0]), f2bf(acc[1]), f2bf(acc[2]), f2bf(acc[3])};
            *reinterpret_cast<bf16x4*>(out16 + (size_t)pixel * ldo + c) = o;
        } else {
            for (int q = 0; q < 4 && c + q < C; ++q) out16[(size_t)pixel * ldo + c + q] = f2bf(acc[q]);
        }
    }
}

// Exact 4x upsampling (56 -> 224, the only case the heads use) to bf16 NHWC: the 4 x 4 outputs of input cell (ky, kx)
// read the same 5 x 5 input pixels (src = k + (j - 1.5) / 4: floor = k-1 for j < 2, k for j >= 2, taps floor-1 .. floor+2), so
// one thread produces the whole block for 4 channels from 25 float4 loads instead of 256 -- the generic kernel is bound by
// its 16 L1 loads per output, this one by the output stores.  Rows are combined in increasing order like torch's
// cubic_interp1d, with the same fp32 weights.
__global__ __launch_bounds__(256) void bicubic4x_kernel(const float* __restrict__ in, int B, int h, int w, int C, int ldi,
                                                        bf16_t* __restrict__ out, int ldo)
{
    const int cgroups = C / 4;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)B * h * w * cgroups) return;
    const int cg = (int)(t % cgroups);
    const long cell = t / cgroups;
    const int kx = (int)(cell % w), ky = (int)((cell / w) % h), b = (int)(cell / ((long)w * h));
    const int c = cg * 4;
    float wt[4][4];  // weights of output phase j = 0..3 (same along x and y)
    cubic_weights(0.625f, wt[0]);
    cubic_weights(0.875f, wt[1]);
    cubic_weights(0.125f, wt[2]);
    cubic_weights(0.375f, wt[3]);
    float acc[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.0f;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int yy = min(max(ky - 2 + r, 0), h - 1);
        float4 px[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            const int xx = min(max(kx - 2 + e, 0), w - 1);
            px[e] = *reinterpret_cast<const float4*>(in + (((size_t)b * h + yy) * w + xx) * ldi + c);
        }
        float hz[4][4];  // horizontal interpolants of this input row for the 4 output columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e0 = j < 2 ? 0 : 1;  // first tap column (relative to kx-2)
            hz[j][0] = px[e0].x * wt[j][0] + px[e0 + 1].x * wt[j][1] + px[e0 + 2].x * wt[j][2] + px[e0 + 3].x * wt[j][3];
            hz[j][1] = px[e0].y * wt[j][0] + px[e0 + 1].y * wt[j][1] + px[e0 + 2].y * wt[j][2] + px[e0 + 3].y * wt[j][3];
            hz[j][2] = px[e0].z * wt[j][0] + px[e0 + 1].z * wt[j][1] + px[e0 + 2].z * wt[j][2] + px[e0 + 3].z * wt[j][3];
            hz[j][3] = px[e0].w * wt[j][0] + px[e0 + 1].w * wt[j][1] + px[e0 + 2].w * wt[j][2] + px[e0 + 3].w * wt[j][3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = r - (i < 2 ? 0 : 1);  // which of output row i's four taps this input row is
            if (a < 0 || a > 3) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] += hz[j][q] * wt[i][a];
        }
    }
    const int W = 4 * w;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 o = {f2bf(acc[i][j][0]), f2bf(acc[i][j][1]), f2bf(acc[i][j][2]), f2bf(acc[i][j][3])};
            *reinterpret_cast<bf16x4*>(out + (((size_t)b * 4 * h + 4 * ky + i) * W + 4 * kx + j) * ldo + c) = o;
        }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int cmdiad_conv2d_nhwc_bf16(const cmdiad_conv_args* a, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(a && a->x && a->W, CMDIAD_ERR_ARG, "cmdiad_conv2d_nhwc_bf16: null operand");
    CMDIAD_REQUIRE(a->B > 0 && a->H > 0 && a->Wd > 0 && a->C > 0 && a->C % 64 == 0 && a->N > 0 && a->N % 4 == 0, CMDIAD_ERR_ARG,
                   "cmdiad_conv2d_nhwc_bf16: need C%%64==0 and N%%4==0 (C=%d N=%d)", a->C, a->N);
    CMDIAD_REQUIRE((a->ksize == 3 && (a->stride == 1 || a->stride == 2)) || (a->ksize == 1 && a->stride == 1), CMDIAD_ERR_ARG,
                   "cmdiad_conv2d_nhwc_bf16: 3x3 (stride 1 or 2, padding 1) or 1x1 (stride 1) only (k=%d s=%d)", a->ksize, a->stride);
    CMDIAD_REQUIRE(a->H < 32768 && a->Wd < 32768 && (long)a->B * a->H * a->Wd < (1L << 31), CMDIAD_ERR_ARG,
                   "cmdiad_conv2d_nhwc_bf16: image too large for 32-bit pixel indices");
    CMDIAD_REQUIRE(a->out_f32 || a->out_bf16, CMDIAD_ERR_ARG, "cmdiad_conv2d_nhwc_bf16: no output");
    CMDIAD_REQUIRE(aligned16(a->x) && aligned16(a->W) && (!a->bias || aligned16(a->bias)) &&
                       (!a->residual || (aligned16(a->residual) && a->ldr % 4 == 0)) &&
                       (!a->out_f32 || (aligned16(a->out_f32) && a->ldo32 % 4 == 0)) &&
                       (!a->out_bf16 || (((uintptr_t)a->out_bf16 & 7) == 0 && a->ldo16 % 4 == 0)),
                   CMDIAD_ERR_ARG, "cmdiad_conv2d_nhwc_bf16: operand alignment");
    CMDIAD_REQUIRE(a->act == CMDIAD_ACT_NONE || a->act == CMDIAD_ACT_RELU || a->act == CMDIAD_ACT_RELU_POST, CMDIAD_ERR_ARG,
                   "cmdiad_conv2d_nhwc_bf16: act must be NONE, RELU or RELU_POST");
    const int pad = a->ksize == 3 ? 1 : 0;
    const int Ho = (a->H + 2 * pad - a->ksize) / a->stride + 1, Wo = (a->Wd + 2 * pad - a->ksize) / a->stride + 1;
    const int taps = a->ksize * a->ksize, ktpt = a->C / 64, KT = taps * ktpt;
    const int magic = (65536 + ktpt - 1) / ktpt;
    for (int kt = 0; kt < KT; ++kt)
        CMDIAD_REQUIRE(((kt * magic) >> 16) == kt / ktpt, CMDIAD_ERR_ARG, "cmdiad_conv2d_nhwc_bf16: C=%d too wide for the tap decode", a->C);
    ConvParams p{(const bf16_t*)a->x, ConvGeom{a->B, a->H, a->Wd, a->C, Ho, Wo, a->ksize, a->stride, pad, ktpt, magic},
                 a->B * Ho * Wo, a->N, KT * 64, a->bias, a->residual, a->ldr, a->out_f32, a->ldo32, (bf16_t*)a->out_bf16, a->ldo16};
    GlobalTile W{(const bf16_t*)a->W, KT * 64, a->N};
    const unsigned blocks = (unsigned)(((p.M + S128::BM - 1) / S128::BM) * ((p.N + S128::BN - 1) / S128::BN));
    hipStream_t s = (hipStream_t)stream;
    const int rc = a->act == CMDIAD_ACT_RELU        ? launch_conv(conv_igemm_kernel<S128, CMDIAD_ACT_RELU>, blocks, s, W, p)
                   : a->act == CMDIAD_ACT_RELU_POST ? launch_conv(conv_igemm_kernel<S128, CMDIAD_ACT_RELU_POST>, blocks, s, W, p)
                                                    : launch_conv(conv_igemm_kernel<S128, CMDIAD_ACT_NONE>, blocks, s, W, p);
    if (rc) return rc;
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_conv_stem(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W,
                                int Cout, int stride, uint16_t* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(x && w && bias && out, CMDIAD_ERR_ARG, "cmdiad_conv_stem: null pointer");
    CMDIAD_REQUIRE(B > 0 && Cin >= 1 && Cin <= 4 && Cout > 0 && Cout % 8 == 0 && Cout <= 256 && (stride == 1 || stride == 2),
                   CMDIAD_ERR_ARG, "cmdiad_conv_stem: Cin in 1..4, Cout%%8==0 and <= 256, stride 1 or 2 (Cin=%d Cout=%d)", Cin, Cout);
    CMDIAD_REQUIRE(aligned16(out), CMDIAD_ERR_ARG, "cmdiad_conv_stem: out must be 16-byte aligned");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long threads = (long)B * Ho * Wo * (Cout / 8);
    const int lds = (Cout * Cin * 9 + Cout) * (int)sizeof(float);
    hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), lds, (hipStream_t)stream,
                       x, w, bias, B, Cin, H, W, Ho, Wo, Cout, stride, (bf16_t*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_upsample_bicubic(const float* in, int B, int h, int w, int C, int ldi, int H, int W,
                                       uint16_t* out_bf16_nhwc, int ldo, float* out_f32_nchw, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(in && ((out_bf16_nhwc != nullptr) != (out_f32_nchw != nullptr)), CMDIAD_ERR_ARG,
                   "cmdiad_upsample_bicubic: exactly one of out_bf16_nhwc / out_f32_nchw");
    CMDIAD_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && ldi >= C, CMDIAD_ERR_ARG, "cmdiad_upsample_bicubic: bad shape");
    CMDIAD_REQUIRE(aligned16(in) && ldi % 4 == 0 && (!out_bf16_nhwc || (ldo >= C && ldo % 4 == 0 && ((uintptr_t)out_bf16_nhwc & 7) == 0)),
                   CMDIAD_ERR_ARG, "cmdiad_upsample_bicubic: alignment (ldi%%4, ldo%%4)");
    const long threads = (long)B * H * W * ((C + 3) / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (out_bf16_nhwc && H == 4 * h && W == 4 * w && C % 4 == 0) {
        const long cells = (long)B * h * w * (C / 4);
        hipLaunchKernelGGL(bicubic4x_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, B, h, w, C, ldi,
                           (bf16_t*)out_bf16_nhwc, ldo);
        CMDIAD_CHECK_LAUNCH();
        return CMDIAD_OK;
    }
    if (out_f32_nchw)
        hipLaunchKernelGGL(bicubic_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, B, h, w, C, ldi, H, W, (bf16_t*)nullptr, 0, out_f32_nchw);
    else
        hipLaunchKernelGGL(bicubic_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, B, h, w, C, ldi, H, W, (bf16_t*)out_bf16_nhwc, ldo, (float*)nullptr);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
