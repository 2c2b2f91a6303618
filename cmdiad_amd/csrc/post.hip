// On-device tail of the scorer: the 8-bit Gaussian blur of the anomaly maps exactly as the
// reference's KNNGaussianBlur computes it through Pillow (utils/utils.py:71-83), and the linear one-class-SVM
// scoring of the per-pixel map pairs (features.py:352-358 fit on the host, score_samples at multiple_features.py:990-992).
//
// Blur: map / max -> *255 -> truncate to uint8 (torchvision ToPILImage on a float tensor) -> Pillow's GaussianBlur =
// three passes of an extended box filter per axis in 2^24 fixed point (restated in oracle/cmdiad_oracle.c
// orc_pil_gaussian_blur_u8, bit-exact against the installed Pillow) -> /255 -> * max.  Integer arithmetic: the
// kernel must match the oracle bit for bit; the two float steps use correctly rounded division and no contraction
// (this file is built with -ffp-contract=off).
// One 256-thread block per map; the whole 8-bit image lives in LDS twice (ping-pong, rows padded to an odd dword
// stride so that lane y reading column x of row y is conflict-free); thread y filters line y.  The third horizontal
// pass writes transposed, so the vertical passes are horizontal passes too, and the sixth pass writes the float
// result back in the original orientation.
#include <math.h>

#include "common.h"

namespace {

// one line of Pillow's ImagingLineBoxBlur8 (edgeA <= edgeB branch); in/out are strided byte accessors
template <class In, class Out>
__device__ __forceinline__ void line_box_blur8(In in, Out out, int lastx, int r, uint32_t ww, uint32_t fw)
{
    const int edgeA = r + 1, edgeB = lastx - r;  // = cols - r - 1
    uint32_t acc = in(0) * (uint32_t)(r + 1);
    for (int x = 0; x < edgeA - 1; ++x) acc += in(x);
    acc += in(lastx) * (uint32_t)(r - edgeA + 1);
    for (int x = 0; x < edgeA; ++x) {
        acc += in(x + r) - in(0);
        out(x, (uint8_t)((acc * ww + (in(0) + in(x + r + 1)) * fw + (1u << 23)) >> 24));
    }
    for (int x = edgeA; x < edgeB; ++x) {
        acc += in(x + r) - in(x - r - 1);
        out(x, (uint8_t)((acc * ww + (in(x - r - 1) + in(x + r + 1)) * fw + (1u << 23)) >> 24));
    }
    for (int x = edgeB; x <= lastx; ++x) {
        acc += in(lastx) - in(x - r - 1);
        out(x, (uint8_t)((acc * ww + (in(x - r - 1) + in(lastx)) * fw + (1u << 23)) >> 24));
    }
}

__device__ __forceinline__ int padded_stride(int n) { return ((n + 3) / 4 * 4) + 4; }  // bytes; (stride/4) is odd for n = 224

__global__ __launch_bounds__(256) void blur8_maps_kernel(const float* __restrict__ maps, int H, int W, int r, uint32_t ww,
                                                         uint32_t fw, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_u8[];
    __shared__ float s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = maps + (size_t)blockIdx.x * H * W;
    float* dst = out + (size_t)blockIdx.x * H * W;
    const int SW = padded_stride(W), SH = padded_stride(H);
    unsigned char* A = lds_u8;                      // [H][SW]  row-major image
    unsigned char* T = lds_u8 + (size_t)(H > W ? H : W) * (SW > SH ? SW : SH);  // second buffer (either orientation)

    // ---- map maximum (utils.py:81 img.max())
    float mx = -__builtin_inff();
    for (int i = tid; i < H * W; i += 256) mx = fmaxf(mx, src[i]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));

    // ---- (img / max).mul(255).byte()
    for (int i = tid; i < H * W; i += 256) {
        const float q = __fdiv_rn(src[i], mx) * 255.0f;
        A[(i / W) * SW + (i % W)] = (unsigned char)(int)q;  // .byte(): truncation; q in [0, 255]
    }
    __syncthreads();

    // ---- three passes along x (rows of A), the last one stored transposed into T as [W][SH]
    for (int y = tid; y < H; y += 256) {
        const unsigned char* a = A + y * SW;
        unsigned char* t = T + y * SW;
        line_box_blur8([&](int x) -> uint32_t { return a[x]; }, [&](int x, uint8_t v) { t[x] = v; }, W - 1, r, ww, fw);
    }
    __syncthreads();
    for (int y = tid; y < H; y += 256) {
        const unsigned char* t = T + y * SW;
        unsigned char* a = A + y * SW;
        line_box_blur8([&](int x) -> uint32_t { return t[x]; }, [&](int x, uint8_t v) { a[x] = v; }, W - 1, r, ww, fw);
    }
    __syncthreads();
    for (int y = tid; y < H; y += 256) {
        const unsigned char* a = A + y * SW;
        line_box_blur8([&](int x) -> uint32_t { return a[x]; }, [&](int x, uint8_t v) { T[x * SH + y] = v; }, W - 1, r, ww, fw);
    }
    __syncthreads();
    // ---- three passes along y (rows of the transposed image, length H)
    for (int x = tid; x < W; x += 256) {
        const unsigned char* t = T + x * SH;
        unsigned char* a = A + x * SH;
        line_box_blur8([&](int y) -> uint32_t { return t[y]; }, [&](int y, uint8_t v) { a[y] = v; }, H - 1, r, ww, fw);
    }
    __syncthreads();
    for (int x = tid; x < W; x += 256) {
        const unsigned char* a = A + x * SH;
        unsigned char* t = T + x * SH;
        line_box_blur8([&](int y) -> uint32_t { return a[y]; }, [&](int y, uint8_t v) { t[y] = v; }, H - 1, r, ww, fw);
    }
    __syncthreads();
    for (int x = tid; x < W; x += 256) {
        const unsigned char* t = T + x * SH;
        // ToTensor: byte -> float / 255, then * map_max (utils.py:82)
        line_box_blur8([&](int y) -> uint32_t { return t[y]; },
                       [&](int y, uint8_t v) { dst[(size_t)y * W + x] = __fdiv_rn((float)v, 255.0f) * mx; }, H - 1, r, ww, fw);
    }
}

struct ScoreParams {
    int K;
    float lambda[4];
    double coef[4];
    double offset;
};

// out[b, p] = ((sum_k double(float(lambda_k * map[b,k,p])) * coef_k) - offset) + offset   (score_samples = decision + offset)
__global__ __launch_bounds__(256) void ocsvm_score_maps_kernel(const float* __restrict__ maps, int HW, ScoreParams p,
                                                               double* __restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    double s = 0.0;
    for (int k = 0; k < p.K; ++k) {
        const float v = p.lambda[k] * maps[((size_t)b * p.K + k) * HW + i];
        s = s + (double)v * p.coef[k];
    }
    out[(size_t)b * HW + i] = (s - p.offset) + p.offset;
}

}  // namespace

extern "C" size_t cmdiad_blur8_lds_bytes(int H, int W)
{
    const int SW = ((W + 3) / 4 * 4) + 4, SH = ((H + 3) / 4 * 4) + 4;
    return (size_t)2 * (H > W ? H : W) * (SW > SH ? SW : SH);
}

extern "C" int cmdiad_blur8_maps(const float* maps, int n_maps, int H, int W, float radius, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(maps && out, CMDIAD_ERR_ARG, "cmdiad_blur8_maps: null pointer");
    CMDIAD_REQUIRE(n_maps >= 0 && H > 0 && W > 0 && radius > 0.0f, CMDIAD_ERR_ARG, "cmdiad_blur8_maps: bad sizes");
    if (n_maps == 0) return CMDIAD_OK;
    // Pillow BoxBlur.c _gaussian_blur_radius (float variables, double intermediates) and the 2^24 fixed-point weights of
    // ImagingHorizontalBoxBlur, evaluated on the host exactly as the C library does
    const int passes = 3;
    float sigma2, L, l, a;
    sigma2 = radius * radius / passes;
    L = sqrt(12.0 * sigma2 + 1.0);
    l = floor((L - 1.0) / 2.0);
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    const float fr = l + a;
    const int r = (int)fr;
    const uint32_t ww = (uint32_t)((uint32_t)(1 << 24) / (fr * 2 + 1));
    const uint32_t fw = ((1 << 24) - (r * 2 + 1) * ww) / 2;
    CMDIAD_REQUIRE(W >= 2 * r + 2 && H >= 2 * r + 2, CMDIAD_ERR_ARG,
                   "cmdiad_blur8_maps: a side of %dx%d is shorter than the box window (Pillow's short-line branch is not implemented)", H, W);
    const size_t lds = cmdiad_blur8_lds_bytes(H, W);
    CMDIAD_REQUIRE(lds <= 160 * 1024 - 64, CMDIAD_ERR_ARG, "cmdiad_blur8_maps: %dx%d does not fit the 160 KiB LDS", H, W);
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute((const void*)blur8_maps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            cmdiad_set_error("cmdiad_blur8_maps: hipFuncSetAttribute(%zu) failed", lds);
            return CMDIAD_ERR_LAUNCH;
        }
        attr = lds;
    }
    hipLaunchKernelGGL(blur8_maps_kernel, dim3(n_maps), dim3(256), lds, (hipStream_t)stream, maps, H, W, r, ww, fw, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_ocsvm_score_maps(const float* maps, int B, int K, int HW, const float* lambdas, const double* coef,
                                       double offset, double* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(maps && lambdas && coef && out, CMDIAD_ERR_ARG, "cmdiad_ocsvm_score_maps: null pointer");
    CMDIAD_REQUIRE(B >= 0 && K >= 1 && K <= 4 && HW > 0, CMDIAD_ERR_ARG, "cmdiad_ocsvm_score_maps: need 1 <= K <= 4 (K=%d)", K);
    if (B == 0) return CMDIAD_OK;
    ScoreParams p{};
    p.K = K;
    for (int k = 0; k < K; ++k) { p.lambda[k] = lambdas[k]; p.coef[k] = coef[k]; }
    p.offset = offset;
    hipLaunchKernelGGL(ocsvm_score_maps_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, maps, HW, p, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
