// One-class SVM fit on the device (reference feature_extractors/features.py:352-358: detect_fuser.fit(s_lib),
// seg_fuser.fit(s_map_lib)).  The reference calls scikit-learn's SGDOneClassSVM [external: scikit-learn 1.7,
// linear_model/_stochastic_gradient.py:_fit_one_class -> _sgd_fast._plain_sgd32, utils/_seq_dataset.ArrayDataset32,
// utils/_weight_vector.WeightVector32, utils/_random.our_rand_r]; this file restates that published algorithm for float32
// inputs so that coef_, offset_ and n_iter_ come out bit-identical (tests/test_gpu_ocsvm.py compares with scikit-learn itself):
//
//   * every epoch re-shuffles the sample order in place with a Fisher-Yates pass driven by xorshift32 from the SAME seed
//     (`dataset.shuffle(seed)` takes the seed by value), i.e. order_{e+1} = order_e o P with one fixed permutation P.
//     P is built in parallel: the generator is linear over GF(2), so thread k jumps to step 64 k with the 32 x 32 bit matrices
//     M^(2^e) and emits j_i = i + (r_i mod 2^31) mod (n - i); the swaps (i, j_i) are then resolved without executing them:
//     position i is final after step i and holds what position j_i held at that time, and a position p >= i at time i holds
//     what the LAST earlier step s < i with j_s = p moved there (the old content of position s at time s, recursively), or its
//     original content -- a chase over the sorted (j_s, s) pairs, a handful of binary searches per element.
//   * one epoch is a strictly sequential recurrence (hinge test of sample i reads the weights sample i-1 wrote): one wave
//     walks the samples (gathered into visiting order beforehand); per 64 samples the lanes compute everything that depends
//     only on the step number in parallel (learning rate eta_t = 1 / (alpha (t0 + t - 1)) in fp64, the weight-scale factor,
//     the two possible intercept updates) and the wave then runs the 64 dependent steps on uniform values.  Arithmetic types
//     as in the Cython sources: float weights, double wscale / dot product / intercept / loss, float division c / (float)wscale.
//   * stopping as in _plain_sgd: sumloss > best_loss - tol n for n_iter_no_change = 5 epochs, or max_iter.
#include <hipcub/hipcub.hpp>

#include "common.h"

#pragma clang fp contract(off)   // no fused multiply-adds: the x86-64 baseline build of scikit-learn has none

namespace {

__device__ __forceinline__ unsigned xs_step(unsigned s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

// j[i] = i + (r_i mod 2^31) mod (n - i), r_i = state after i + 1 generator steps; one thread = 64 consecutive steps
__global__ __launch_bounds__(256) void fy_targets_kernel(const unsigned* __restrict__ pow2, unsigned seed, int n, int* __restrict__ j_out)
{
    __shared__ unsigned s_pow[32 * 32];
    for (int e = threadIdx.x; e < 1024; e += 256) s_pow[e] = pow2[e];
    __syncthreads();
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i0 = k * 64;
    if (i0 >= n - 1) return;
    unsigned s = seed;
    for (int e = 0; e < 32; ++e) {   // s = M^(i0) seed
        if (!((i0 >> e) & 1)) continue;
        unsigned y = 0;
        for (int b = 0; b < 32; ++b)
            if ((s >> b) & 1) y ^= s_pow[e * 32 + b];
        s = y;
    }
    for (int d = 0; d < 64; ++d) {
        const long long i = i0 + d;
        if (i >= n - 1) break;
        s = xs_step(s);
        const unsigned r = s & 0x7FFFFFFFu;
        j_out[i] = (int)(i + (long long)(r % (unsigned)(n - i)));
    }
}

__global__ __launch_bounds__(256) void fy_keys_kernel(const int* __restrict__ j, int n, unsigned long long* __restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n - 1) keys[i] = ((unsigned long long)(unsigned)j[i] << 32) | (unsigned)i;
}

// P[i] = position (before the pass) of the element that ends at position i
__global__ __launch_bounds__(256) void fy_chase_kernel(const int* __restrict__ j, const unsigned long long* __restrict__ keys, int n,
                                                       int* __restrict__ P)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned pos = i < n - 1 ? (unsigned)j[i] : (unsigned)(n - 1), t = (unsigned)i;
    const int m = n - 1;   // number of keys
    for (;;) {
        // the largest key < (pos, t) -- if its position part is pos, step s moved the old content of position s here
        const unsigned long long q = ((unsigned long long)pos << 32) | t;
        int lo = 0, hi = m;   // first index with key >= q
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < q) lo = mid + 1;
            else hi = mid;
        }
        if (lo == 0) break;
        const unsigned long long kk = keys[lo - 1];
        if ((unsigned)(kk >> 32) != pos) break;
        pos = (unsigned)kk;   // = s
        t = pos;
    }
    P[i] = (int)pos;
}

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ a, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = i;
}

template <int F>
__global__ __launch_bounds__(256) void gather_order_kernel(const int* __restrict__ order_in, const int* __restrict__ P, const float* __restrict__ X,
                                                           int n, int* __restrict__ order_out, float* __restrict__ xp)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int src = order_in[P[i]];
    order_out[i] = src;
#pragma unroll
    for (int f = 0; f < F; ++f) xp[(size_t)i * F + f] = X[(size_t)src * F + f];
}

struct SgdState {
    double intercept, t, sumloss, wscale;   // WeightVector32.wscale is a double (utils/_weight_vector.pxd.tp), the weights are floats
    float w[4];
};

__device__ __forceinline__ float bcast(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); }
__device__ __forceinline__ double bcast(double v, int k)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), k);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// One epoch of _plain_sgd32 (hinge loss, L2 penalty, 'optimal' schedule, one_class, dense float32 data, unit sample weights).
template <int F>
__global__ __launch_bounds__(64) void ocsvm_epoch_kernel(const float* __restrict__ xp, int n, double alpha, double optimal_init, SgdState* st)
{
    const int lane = threadIdx.x;
    double intercept = st->intercept, sumloss = 0.0;
    const double t0 = st->t;
    float w[F];
#pragma unroll
    for (int f = 0; f < F; ++f) w[f] = st->w[f];
    double wscale = st->wscale;
    for (int base = 0; base < n; base += 64) {
        const int cnt = min(64, n - base);
        // per-lane: everything of sample base + lane that depends on the step number only
        float x[F];
        const int me = min(base + lane, n - 1);
#pragma unroll
        for (int f = 0; f < F; ++f) x[f] = xp[(size_t)me * F + f];
        const double t = t0 + (double)(base + lane);
        const double eta = 1.0 / (alpha * (optimal_init + t - 1.0));
        const float sc = (float)fmax(0.0, 1.0 - ((1.0 - 0.0) * eta * alpha));   // w.scale(max(0, 1 - (1 - l1_ratio) eta alpha)), l1_ratio = 0
        const float c = (float)eta;                                           // w.add(..., update): update = -eta * (-1), as c_type
        const double two_eta_alpha = 2. * eta * alpha;
        const double iu_hit = eta - two_eta_alpha, iu_miss = 0.0 - two_eta_alpha;   // intercept_update = update - 2 eta alpha
        for (int k = 0; k < cnt; ++k) {   // the dependent chain, on wave-uniform values
            double innerprod = 0.0;
            float xk[F];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                xk[f] = bcast(x[f], k);
                innerprod += (double)(w[f] * xk[f]);
            }
            innerprod *= wscale;
            const double p = (double)(float)innerprod + intercept;
            const bool hit = p <= 1.0;                       // Hinge(threshold 1): loss = 1 - p, dloss = -1
            if (hit) sumloss += 1.0 - p;
            wscale *= (double)bcast(sc, k);                  // scale(float c): self.wscale *= c
            if (wscale < 1e-6) {                             // reset_wscale below the float32 threshold: sscal with (float)wscale (uniform branch)
#pragma unroll
                for (int f = 0; f < F; ++f) w[f] *= (float)wscale;
                wscale = 1.0;
            }
            if (hit) {                                       // uniform: add(): `cdef float wscale = self.wscale`; w += x * (c / wscale)
                const float q = bcast(c, k) / (float)wscale;
#pragma unroll
                for (int f = 0; f < F; ++f) w[f] = (float)((double)w[f] + (double)xk[f] * (double)q);
            }
            intercept += hit ? bcast(iu_hit, k) : bcast(iu_miss, k);
        }
    }
    if (lane == 0) {
        st->intercept = intercept;
        st->t = t0 + (double)n;
        st->sumloss = sumloss;
#pragma unroll
        for (int f = 0; f < F; ++f) st->w[f] = w[f];
        st->wscale = wscale;
    }
}

size_t up256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" size_t cmdiad_ocsvm_fit_workspace_bytes(int n, int F)
{
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n > 1 ? n - 1 : 1);
    return up256((size_t)n * 4) * 4 + up256((size_t)n * 8) * 2 + up256((size_t)n * F * 4) + up256(sort_tmp) + up256(sizeof(SgdState)) + 4096;
}

// X [n, F] float32 on the device -> coef_out[F] (float, host), *offset_out, *n_iter_out (host).  pow2[32][32]: column b of
// M^(2^e) for the xorshift32 step M (computed by the caller: cmdiad_amd/ocsvm.py).  Synchronous: the stopping rule is evaluated
// on the host after every epoch.
extern "C" int cmdiad_ocsvm_fit(const float* X, int n, int F, double nu, int max_iter, double tol, int n_iter_no_change, uint32_t seed,
                                const uint32_t* pow2_host, float* coef_out, double* offset_out, int* n_iter_out, void* workspace,
                                size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(X && pow2_host && coef_out && offset_out && n_iter_out && workspace, CMDIAD_ERR_ARG, "cmdiad_ocsvm_fit: null pointer");
    CMDIAD_REQUIRE(n >= 2 && F >= 1 && F <= 4 && nu > 0.0 && max_iter >= 1, CMDIAD_ERR_ARG, "cmdiad_ocsvm_fit: need n >= 2, 1 <= F <= 4 (n=%d F=%d)", n, F);
    CMDIAD_REQUIRE(workspace_bytes >= cmdiad_ocsvm_fit_workspace_bytes(n, F), CMDIAD_ERR_WORKSPACE, "cmdiad_ocsvm_fit: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int* j = (int*)ws; ws += up256((size_t)n * 4);
    int* P = (int*)ws; ws += up256((size_t)n * 4);
    int* order_a = (int*)ws; ws += up256((size_t)n * 4);
    int* order_b = (int*)ws; ws += up256((size_t)n * 4);
    unsigned long long* keys = (unsigned long long*)ws; ws += up256((size_t)n * 8);
    unsigned long long* keys_sorted = (unsigned long long*)ws; ws += up256((size_t)n * 8);
    float* xp = (float*)ws; ws += up256((size_t)n * F * 4);
    SgdState* st = (SgdState*)ws; ws += up256(sizeof(SgdState));
    unsigned* pow2 = (unsigned*)ws; ws += 4096;
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, keys, keys_sorted, n - 1);
    void* sort_ws = ws;

#define OCSVM_HIP(call) do { if ((call) != hipSuccess) { cmdiad_set_error("cmdiad_ocsvm_fit: %s failed", #call); return CMDIAD_ERR_LAUNCH; } } while (0)
    OCSVM_HIP(hipMemcpyAsync(pow2, pow2_host, 4096, hipMemcpyHostToDevice, s));
    const unsigned sd = seed == 0 ? 1u : seed;   // our_rand_r: "seed shouldn't ever be 0"
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(fy_targets_kernel, dim3((unsigned)(((long long)n + 64 * 256 - 1) / (64 * 256))), dim3(256), 0, s, pow2, sd, n, j);
    hipLaunchKernelGGL(fy_keys_kernel, dim3(nb), dim3(256), 0, s, j, n, keys);
    OCSVM_HIP(hipcub::DeviceRadixSort::SortKeys(sort_ws, sort_tmp, keys, keys_sorted, n - 1, 0, 64, s));
    hipLaunchKernelGGL(fy_chase_kernel, dim3(nb), dim3(256), 0, s, j, keys_sorted, n, P);
    hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, s, order_a, n);

    // SGDOneClassSVM._fit: alpha = nu / 2, coef_ = 0, offset_ = 0 -> intercept = 1, t_ = 1; 'optimal': typw = sqrt(1 / sqrt(alpha)),
    // eta0 = typw / max(1, dloss(1, -typw)) with Hinge.dloss(y = 1, p = -typw) = -1 -> eta0 = typw, optimal_init = 1 / (eta0 alpha)
    const double alpha = nu / 2.0;
    const double typw = sqrt(1.0 / sqrt(alpha));
    const double initial_eta0 = typw / fmax(1.0, -1.0);
    const double optimal_init = 1.0 / (initial_eta0 * alpha);
    SgdState h{};
    h.intercept = 1.0;
    h.t = 1.0;
    h.wscale = 1.0;
    OCSVM_HIP(hipMemcpyAsync(st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    double best_loss = INFINITY;
    int no_improve = 0, epoch = 0;
    int* oin = order_a;
    int* oout = order_b;
    for (epoch = 0; epoch < max_iter; ++epoch) {
        switch (F) {
        case 1: hipLaunchKernelGGL(gather_order_kernel<1>, dim3(nb), dim3(256), 0, s, oin, P, X, n, oout, xp);
                hipLaunchKernelGGL(ocsvm_epoch_kernel<1>, dim3(1), dim3(64), 0, s, xp, n, alpha, optimal_init, st); break;
        case 2: hipLaunchKernelGGL(gather_order_kernel<2>, dim3(nb), dim3(256), 0, s, oin, P, X, n, oout, xp);
                hipLaunchKernelGGL(ocsvm_epoch_kernel<2>, dim3(1), dim3(64), 0, s, xp, n, alpha, optimal_init, st); break;
        case 3: hipLaunchKernelGGL(gather_order_kernel<3>, dim3(nb), dim3(256), 0, s, oin, P, X, n, oout, xp);
                hipLaunchKernelGGL(ocsvm_epoch_kernel<3>, dim3(1), dim3(64), 0, s, xp, n, alpha, optimal_init, st); break;
        default: hipLaunchKernelGGL(gather_order_kernel<4>, dim3(nb), dim3(256), 0, s, oin, P, X, n, oout, xp);
                 hipLaunchKernelGGL(ocsvm_epoch_kernel<4>, dim3(1), dim3(64), 0, s, xp, n, alpha, optimal_init, st); break;
        }
        int* tmp = oin; oin = oout; oout = tmp;
        OCSVM_HIP(hipMemcpyAsync(&h, st, sizeof(h), hipMemcpyDeviceToHost, s));
        OCSVM_HIP(hipStreamSynchronize(s));
        bool finite = std::isfinite(h.intercept);
        for (int f = 0; f < F; ++f) finite = finite && std::isfinite(h.w[f]);
        if (!finite) {
            cmdiad_set_error("cmdiad_ocsvm_fit: floating-point under-/overflow at epoch %d (scikit-learn raises ValueError here)", epoch + 1);
            return CMDIAD_ERR_ARG;
        }
        if (tol > -INFINITY && h.sumloss > best_loss - tol * (double)n) ++no_improve;
        else no_improve = 0;
        if (h.sumloss < best_loss) best_loss = h.sumloss;
        if (no_improve >= n_iter_no_change) { ++epoch; break; }
    }
#undef OCSVM_HIP
    for (int f = 0; f < F; ++f) coef_out[f] = h.w[f] * (float)h.wscale;   // w.reset_wscale(): sscal by (float)wscale
    *offset_out = 1.0 - h.intercept;
    *n_iter_out = epoch;
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
