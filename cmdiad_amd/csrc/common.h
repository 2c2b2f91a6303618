// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of cmdiad_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cmdiad_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CMDIAD_WAVE 64

// Last error text, returned through cmdiad_last_error().  Never throws across the C ABI.
void cmdiad_set_error(const char* fmt, ...);

#define CMDIAD_REQUIRE(cond, code, ...)        \
    do {                                       \
        if (!(cond)) {                         \
            cmdiad_set_error(__VA_ARGS__);     \
            return (code);                     \
        }                                      \
    } while (0)

#define CMDIAD_CHECK_LAUNCH()                                                  \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            cmdiad_set_error("%s:%d launch: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return CMDIAD_ERR_LAUNCH;                                          \
        }                                                                      \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// erf-GELU (nn.GELU() default, utils/utils.py:97, models/models.py:129):  GELU(x) = x - h for x > 0, h otherwise, with
// h = 0.5 x erfc(|x| / sqrt 2) and erfc(z) = 2^(-q(z)), q a degree-6 polynomial fitted to -log2 erfc on [0, 4.3] (beyond it
// erfc < 2e-9).  |error| <= 5.3e-7 absolute on the GELU value -- a tenth of a bf16 half-ulp, the same bound as the
// Abramowitz-Stegun 7.1.26 form it replaces -- with ONE transcendental (v_exp_f32, quarter rate) instead of two (rcp + exp):
// ~16 VALU issue slots per element instead of ~22, and the Horner steps are plain FMAs that the 4-wide form below hands to
// v_pk_fma_f32 two elements at a time.  The fc1 epilogues run this on 77 M activations per ViT layer; in the two-group
// persistent GEMM both waves of a SIMD evaluate it between the same two barriers, where it is not hidden by anything.
#define CMDIAD_GELU_Q6 -2.3885208886e-04f
#define CMDIAD_GELU_Q5 4.0851729330e-03f
#define CMDIAD_GELU_Q4 -3.1410888601e-02f
#define CMDIAD_GELU_Q3 1.4969202041e-01f
#define CMDIAD_GELU_Q2 9.1851604883e-01f
#define CMDIAD_GELU_Q1 1.6277476882e+00f
#define CMDIAD_GELU_Q0 2.2232231590e-05f

__device__ __forceinline__ float gelu_erf(float x)
{
    const float z = fminf(fabsf(x) * 0.70710678118654752440f, 4.3f);
    float q = fmaf(CMDIAD_GELU_Q6, z, CMDIAD_GELU_Q5);
    q = fmaf(q, z, CMDIAD_GELU_Q4);
    q = fmaf(q, z, CMDIAD_GELU_Q3);
    q = fmaf(q, z, CMDIAD_GELU_Q2);
    q = fmaf(q, z, CMDIAD_GELU_Q1);
    q = fmaf(q, z, CMDIAD_GELU_Q0);
    const float h = (0.5f * x) * __builtin_amdgcn_exp2f(-q);
    return x > 0.0f ? x - h : h;
}

// the same on four values: identical arithmetic per element (so every kernel gives the same bits whichever form it uses)
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x)
{
    f32x4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = fminf(fabsf(x[i]) * 0.70710678118654752440f, 4.3f);
    const f32x4 c5 = {CMDIAD_GELU_Q5, CMDIAD_GELU_Q5, CMDIAD_GELU_Q5, CMDIAD_GELU_Q5};
    const f32x4 c4 = {CMDIAD_GELU_Q4, CMDIAD_GELU_Q4, CMDIAD_GELU_Q4, CMDIAD_GELU_Q4};
    const f32x4 c3 = {CMDIAD_GELU_Q3, CMDIAD_GELU_Q3, CMDIAD_GELU_Q3, CMDIAD_GELU_Q3};
    const f32x4 c2 = {CMDIAD_GELU_Q2, CMDIAD_GELU_Q2, CMDIAD_GELU_Q2, CMDIAD_GELU_Q2};
    const f32x4 c1 = {CMDIAD_GELU_Q1, CMDIAD_GELU_Q1, CMDIAD_GELU_Q1, CMDIAD_GELU_Q1};
    const f32x4 c0 = {CMDIAD_GELU_Q0, CMDIAD_GELU_Q0, CMDIAD_GELU_Q0, CMDIAD_GELU_Q0};
    f32x4 q = __builtin_elementwise_fma(f32x4{CMDIAD_GELU_Q6, CMDIAD_GELU_Q6, CMDIAD_GELU_Q6, CMDIAD_GELU_Q6}, z, c5);
    q = __builtin_elementwise_fma(q, z, c4);
    q = __builtin_elementwise_fma(q, z, c3);
    q = __builtin_elementwise_fma(q, z, c2);
    q = __builtin_elementwise_fma(q, z, c1);
    q = __builtin_elementwise_fma(q, z, c0);
    const f32x4 hx = x * 0.5f;
    f32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = hx[i] * __builtin_amdgcn_exp2f(-q[i]);
        out[i] = x[i] > 0.0f ? x[i] - h : h;
    }
    return out;
}

// d/dx of the above: Phi(x) + x phi(x), sharing the one exponential (exp(-x^2/2) = exp(-z^2))
__device__ __forceinline__ float gelu_erf_grad(float x)
{
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-z * z);
    const float erfa = 1.0f - p * t * e;
    return 0.5f * (1.0f + copysignf(erfa, x)) + x * 0.39894228040143267794f * e;
}

// 64-bit key: high word = fp32 bits of a NON-NEGATIVE value, low word = index.  Integer order of
// the key == (value, index) lexicographic order, so min over keys = smallest value, lowest index.
__device__ __forceinline__ unsigned long long pack_key(float v, unsigned idx)
{
    return ((unsigned long long)__float_as_uint(v) << 32) | idx;
}

// max over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in all 16: quad xor 1, quad xor 2, half-row mirror,
// row mirror -- VALU only (four __shfl_xor steps are four ds_bpermute round trips through the LDS pipeline).
__device__ __forceinline__ float row16_max(float v)
{
    int i = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false)));
    i = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false)));
    i = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x141, 0xF, 0xF, false)));
    i = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x140, 0xF, 0xF, false)));
    return v;
}

// The same for N >= 4 values at once, ONE instruction per value and step (v_max_f32 with the DPP modifier on its first
// source): the builtin form above costs a copy, the DPP move, a quieting v_max v, v, v and the max.  Inline asm is outside
// the compiler's hazard tracking: a DPP read needs two wait states after the VALU write of its source -- the s_nop covers
// the values' producers, and inside the batch consecutive steps of one value are N - 1 >= 3 instructions apart (the
// statements are volatile: their order is kept).  No NaN inputs.
template <int N>
__device__ __forceinline__ void row16_max_batch(float (&v)[N])
{
    static_assert(N >= 4, "spacing between dependent DPP steps");
    asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
    asm volatile("s_nop 1");   // nothing the compiler places next may read a lane-crossed result too early either
}

// Sum over the 64 lanes, result wave-uniform: four DPP steps inside the rows of 16 (as row16_max), then the four row sums are
// read as scalars and added in row order -- ~11 VALU instructions against six ds_bpermute round trips for the xor butterfly
// (whose summation order differs: results agree to fp32 rounding, not bitwise).
__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return ((r0 + r1) + r2) + r3;
}

// (a, b) = (v, v) -> v_permlane32_swap exchanges a's upper 32 lanes with b's lower 32: afterwards {a, b} = {v[lane], v[lane ^ 32]}
// in some order on every lane, so max(a, b) / a + b are the cross-half reductions -- one VALU instruction instead of the
// ds_bpermute round trip of __shfl_xor(v, 32).  (Inline asm: the builtin folds the two results of equal inputs into one;
// two wait states between the VALU write of the sources and the permlane read, outside the compiler's hazard tracking.)
__device__ __forceinline__ void half_swap(float v, float& a, float& b)
{
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_max(float v) { float a, b; half_swap(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float v) { float a, b; half_swap(v, a, b); return a + b; }

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
