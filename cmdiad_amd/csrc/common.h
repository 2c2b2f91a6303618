// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of cmdiad_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cmdiad_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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

// erf-GELU (nn.GELU() default, utils/utils.py:97, models/models.py:129).  erf by Abramowitz-Stegun 7.1.26 on |z| with the odd
// extension: |error| <= 5e-7 absolute on the GELU value, a tenth of a bf16 half-ulp -- 15 VALU operations instead of the
// ~37 of libm's erff, which made the fc1 epilogues (77 M activations per ViT layer) a third of those GEMMs' time.
__device__ __forceinline__ float gelu_erf(float x)
{
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfa = 1.0f - p * t * __expf(-z * z);  // erf(|z|)
    return 0.5f * x * (1.0f + copysignf(erfa, x));
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

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
