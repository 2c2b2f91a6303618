// Farthest point sampling + centre gather for gfx950.
//
// Replaces pointnet2_ops.furthest_point_sample + gather_operation as called from the
// reference at models/models.py:76-77.  Semantics = oracle/cmdiad_oracle.c:orc_fps,
// bit-for-bit: running min initialised to 1e10, points with |p|^2 <= 1e-3 skipped, distance
// (dx*dx + dy*dy) + dz*dz with one rounding per operation (this file is compiled with
// -ffp-contract=off), argmax ties -> lowest index.
//
// One 1024-thread workgroup (16 waves) per cloud: FPS is a chain of G-1 dependent argmax
// rounds, so the design goal is the latency of ONE round.  In the fast path the cloud and
// its running-min array live entirely in registers (point k -> thread k%1024, slot k/1024,
// so the initial load is coalesced); a round is PPT x {3 sub, 3 mul, 2 add, min, cmp, 2 sel}
// per lane, a 6-step wave64 shuffle reduction of a packed (value, ~index) key, one LDS
// hand-off across the 16 waves (double-buffered, so ONE barrier per round) and a scalar
// (SGPR) fetch of the winner's coordinates.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kThreads = 1024;         // fallback kernel
constexpr int kWaves = kThreads / 64;
constexpr int kMaxRegPoints = 512 * 56;  // largest cloud the register-resident path holds

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = shfl_xor_u64(v, m);
        v = o > v ? o : v;
    }
    return v;
}

// key: (fp32 bits of running-min, 0xFFFFFFFF - index): max over keys = largest value, lowest index.
__device__ __forceinline__ unsigned long long fps_key(float v, int idx)
{
    return v < 0.0f ? 0ull : (((unsigned long long)__float_as_uint(v) << 32) | (0xFFFFFFFFu - (unsigned)idx));
}

#ifdef CMDIAD_AB_VARIANTS  // first formulation of the round: test-only build (make ab), A/B reference for fps_pk_kernel
#include "ab/fps_reg.inc"
#endif  // CMDIAD_AB_VARIANTS

// Second formulation of the same round, built for VALU throughput: one CU retires 64 lanes x 4 SIMDs / 4 cycles, and the
// round above spends 12 VALU operations per point (48 points per lane: ~4 600 cycles, most of the 2.6 us round).  Here
//   * two points share each arithmetic instruction (v_pk_add_f32 / v_pk_mul_f32 are IEEE per element: same roundings),
//   * the lane keeps only the running MAXIMUM VALUE (v_max3_f32: half an operation per point) instead of (value, slot),
//   * the wave maximum is an integer max of the float bits over DPP row permutes + 4 readlanes (values are >= +0), and
//   * the winner's index is recovered AFTER that: one v_cmp_eq per slot against the wave maximum writes a lane mask to
//     SGPRs, the scalar unit picks the lowest slot with a match and its lowest lane (= lowest point index, the tie rule).
// 6.5 operations per point instead of 12.  Same (value, ~index) keys across the waves, so results are bit-identical.

__device__ __forceinline__ unsigned dpp_max_u32(unsigned v)
{
    // xor 1, xor 2 within quads; mirror within 8, within 16: afterwards every row of 16 lanes holds its maximum
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v = o > v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v = o > v ? o : v;
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

__device__ __forceinline__ unsigned dpp_min_u32(unsigned v)
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v = o < v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v = o < v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v = o < v ? o : v;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v = o < v ? o : v;
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// The round, for the first PPT * kThreads points of a cloud in registers.  TAIL: the cloud continues beyond the register file of
// the CU -- the next `lds_cap` points (coordinates + running minimum, 16 bytes each) live in LDS, whatever is left is re-read from
// global memory every round with its running minimum in `temp` (L2-resident) -- so a cloud of up to 28 672 + 9 728 points never
// leaves the CU and the largest cloud of the 224 x 224 grid (50 176) streams a quarter of its points instead of all of them.
// A thread owns the same tail points in every round, so neither tier needs a barrier of its own.  Tail points have higher
// indices than every register point: the winner is taken from the register slots first, from the tail (lowest index among the
// lanes whose tail maximum IS the wave maximum) only when no register slot matches -- the tie rule (lowest index) holds.
template <int kThreads, int PPT, bool TAIL>
__device__ __forceinline__ void fps_pk_body(const float* __restrict__ p, const int n, const int G, int32_t* __restrict__ out,
                                            float* __restrict__ cen, float4* __restrict__ lds_pts, const int lds_cap,
                                            float* __restrict__ temp)
{
    static_assert(PPT % 2 == 0, "two points per packed operation");
    constexpr int kWaves = kThreads / 64, H = PPT / 2, R = PPT * kThreads;
    __shared__ unsigned long long s_key[2][kWaves];
    const int tid = threadIdx.x;

    f32x2 px[H], py[H], pz[H], t[H];  // slot s = 2h + e holds point s * kThreads + tid
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // unconditional loads of a clamped index: a load under a lane mask is followed by a full wait, PPT round trips one after
            // the other in front of the first round; these go out together (slots past the cloud's end are masked by t = -inf)
            const int k = (2 * h + e) * kThreads + tid;
            const int kk = max(min(k, n - 1), 0);
            const float x = p[kk * 3 + 0], y = p[kk * 3 + 1], z = p[kk * 3 + 2];
            px[h][e] = x; py[h][e] = y; pz[h][e] = z;
            const float mag = (x * x + y * y) + z * z;
            t[h][e] = (k < n && !(mag <= 1e-3f)) ? 1e10f : -__builtin_inff();
        }
    const int n_lds = TAIL ? min(max(n - R, 0), lds_cap) : 0;   // points R .. R + n_lds - 1 in LDS
    const int g0 = R + n_lds;                                    // points g0 .. n - 1 stay in global memory
    if constexpr (TAIL) {
        for (int e = tid; e < n_lds; e += kThreads) {
            const int k = R + e;
            const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
            const float mag = (x * x + y * y) + z * z;
            lds_pts[e] = float4{x, y, z, !(mag <= 1e-3f) ? 1e10f : -__builtin_inff()};
        }
        for (int k = g0 + tid; k < n; k += kThreads) {
            const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
            const float mag = (x * x + y * y) + z * z;
            temp[k] = !(mag <= 1e-3f) ? 1e10f : -__builtin_inff();
        }
    }

    int old = 0;
    if (tid == 0 && G > 0) {
        out[0] = 0;
        if (cen) { cen[0] = p[0]; cen[1] = p[1]; cen[2] = p[2]; }
    }
    for (int j = 1; j < G; ++j) {
        const int so = __builtin_amdgcn_readfirstlane(old);
        const float x1 = p[so * 3 + 0], y1 = p[so * 3 + 1], z1 = p[so * 3 + 2];
        const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
        // From here on the values are handled as their bit patterns: distances are >= +0 (sums of squares) and skipped
        // points carry -inf, so signed-integer min / max / == order them exactly like the float operations would -- without
        // the canonicalising multiply the compiler puts in front of every v_min_f32 / v_max_f32.
        int best = 0;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f32x2 dx = px[h] - X1, dy = py[h] - Y1, dz = pz[h] - Z1;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
            const int a = min((int)__float_as_uint(d[0]), (int)__float_as_uint(t[h][0]));
            const int c = min((int)__float_as_uint(d[1]), (int)__float_as_uint(t[h][1]));
            t[h][0] = __uint_as_float((unsigned)a);
            t[h][1] = __uint_as_float((unsigned)c);
            best = max(best, max(a, c));
        }
        int tb = -1, ti = 0;   // tail: this lane's largest running minimum (bits; -1 = none) and the LOWEST index that attains it
        if constexpr (TAIL) {
            for (int e = tid; e < n_lds; e += kThreads) {
                const float4 q = lds_pts[e];
                const float dx = q.x - x1, dy = q.y - y1, dz = q.z - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const int a = min((int)__float_as_uint(d), (int)__float_as_uint(q.w));
                lds_pts[e].w = __uint_as_float((unsigned)a);
                if (a > tb) { tb = a; ti = R + e; }
            }
            for (int k = g0 + tid; k < n; k += kThreads) {
                const float dx = p[k * 3] - x1, dy = p[k * 3 + 1] - y1, dz = p[k * 3 + 2] - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const int a = min((int)__float_as_uint(d), (int)__float_as_uint(temp[k]));
                temp[k] = __uint_as_float((unsigned)a);
                if (a > tb) { tb = a; ti = k; }
            }
        }
        const unsigned wbits = dpp_max_u32((unsigned)max(best, tb));
        // lowest slot, then lowest lane, whose running minimum IS the wave maximum (none if the wave holds no valid point).
        // Eight slots at a time with an early exit: the lane masks of one chunk fit the scalar registers.
        int slot = -1;
        unsigned long long lanes = 0ull;
#pragma unroll
        for (int c0 = 0; c0 < PPT; c0 += 8) {
            unsigned long long m[8];
            unsigned long long any = 0ull;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int sl = c0 + e;
                m[e] = sl < PPT ? __ballot(__float_as_uint(t[sl >> 1][sl & 1]) == wbits) : 0ull;
                any |= m[e];
            }
            if (any != 0ull) {
#pragma unroll
                for (int e = 7; e >= 0; --e)
                    if (m[e] != 0ull) { slot = c0 + e; lanes = m[e]; }
                break;
            }
        }
        unsigned long long key = 0ull;
        if (slot >= 0) {
            const int besti = slot * kThreads + (tid & ~63) + (__ffsll((long long)lanes) - 1);
            key = ((unsigned long long)wbits << 32) | (0xFFFFFFFFu - (unsigned)besti);
        } else if constexpr (TAIL) {   // wave-uniform: no register slot holds the wave maximum
            const unsigned cand = dpp_min_u32(tb == (int)wbits ? (unsigned)ti : 0xFFFFFFFFu);
            if (cand != 0xFFFFFFFFu) key = ((unsigned long long)wbits << 32) | (0xFFFFFFFFu - cand);
        }
        const int buf = j & 1;
        if ((tid & 63) == 0) s_key[buf][tid >> 6] = key;
        __syncthreads();
        unsigned long long m = s_key[buf][0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
            const unsigned long long o = s_key[buf][w];
            m = o > m ? o : m;
        }
        old = m == 0ull ? 0 : (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull));
        if (tid == 0) {
            out[j] = old;
            if (cen) { cen[j * 3 + 0] = p[old * 3 + 0]; cen[j * 3 + 1] = p[old * 3 + 1]; cen[j * 3 + 2] = p[old * 3 + 2]; }
        }
    }
}

template <int kThreads, int PPT>
__global__ __launch_bounds__(kThreads) void fps_pk_kernel(const float* __restrict__ xyz,
                                                          const int32_t* __restrict__ n_valid, int N, int G,
                                                          int32_t* __restrict__ idx_out,
                                                          float* __restrict__ center_out)
{
    const int b = blockIdx.x;
    fps_pk_body<kThreads, PPT, false>(xyz + (size_t)b * N * 3, n_valid ? n_valid[b] : N, G, idx_out + (size_t)b * G,
                                      center_out ? center_out + (size_t)b * G * 3 : nullptr, nullptr, 0, nullptr);
}

// Ragged batches and clouds beyond the register file: every cloud takes the path ITS OWN point count needs (the padded batch
// maximum used to pick one kernel for all 32 clouds -- a single cloud above 28 672 points sent the whole batch through the
// memory-resident loop, bench.py `var_n`): 40 / 48 / 56 points per lane in registers, and above 28 672 points the tiers of
// fps_pk_body.  512 threads (two waves per SIMD, 256 VGPRs) in every branch; dynamic LDS = the LDS tier (16 bytes per point).
constexpr int kRaggedThreads = 512;
constexpr int kMaxLdsPoints = 9728;   // 152 KiB of the CU's 160 KiB
__global__ __launch_bounds__(kRaggedThreads) void fps_ragged_kernel(const float* __restrict__ xyz,
                                                                    const int32_t* __restrict__ n_valid, int N, int G,
                                                                    int lds_cap, float* __restrict__ temp_ws,
                                                                    int32_t* __restrict__ idx_out, float* __restrict__ center_out)
{
    extern __shared__ __attribute__((aligned(16))) float4 fps_lds[];
    const int b = blockIdx.x;
    const int n = __builtin_amdgcn_readfirstlane(n_valid ? n_valid[b] : N);
    const float* p = xyz + (size_t)b * N * 3;
    int32_t* out = idx_out + (size_t)b * G;
    float* cen = center_out ? center_out + (size_t)b * G * 3 : nullptr;
    if (n <= kRaggedThreads * 40) fps_pk_body<kRaggedThreads, 40, false>(p, n, G, out, cen, nullptr, 0, nullptr);
    else if (n <= kRaggedThreads * 48) fps_pk_body<kRaggedThreads, 48, false>(p, n, G, out, cen, nullptr, 0, nullptr);
    else if (n <= kRaggedThreads * 56) fps_pk_body<kRaggedThreads, 56, false>(p, n, G, out, cen, nullptr, 0, nullptr);
    else fps_pk_body<kRaggedThreads, 56, true>(p, n, G, out, cen, fps_lds, lds_cap, temp_ws ? temp_ws + (size_t)b * N : nullptr);
}

#ifdef CMDIAD_AB_VARIANTS  // the first answer to clouds beyond the register file (every point re-read per round): test-only build
// Fallback for clouds that do not fit the register file of one CU (N > 24*1024): running min in
// a caller-provided global workspace (L2-resident), coordinates re-read every round.
__global__ __launch_bounds__(kThreads) void fps_mem_kernel(const float* __restrict__ xyz,
                                                           const int32_t* __restrict__ n_valid, int N, int G,
                                                           float* __restrict__ temp_ws,
                                                           int32_t* __restrict__ idx_out,
                                                           float* __restrict__ center_out)
{
    __shared__ unsigned long long s_key[2][kWaves];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = n_valid ? n_valid[b] : N;
    const float* p = xyz + (size_t)b * N * 3;
    float* temp = temp_ws + (size_t)b * N;
    int32_t* out = idx_out + (size_t)b * G;
    float* cen = center_out ? center_out + (size_t)b * G * 3 : nullptr;

    for (int k = tid; k < n; k += kThreads) {
        const float x = p[k * 3], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const float mag = (x * x + y * y) + z * z;
        temp[k] = !(mag <= 1e-3f) ? 1e10f : -__builtin_inff();
    }
    int old = 0;
    if (tid == 0 && G > 0) {
        out[0] = 0;
        if (cen) { cen[0] = p[0]; cen[1] = p[1]; cen[2] = p[2]; }
    }
    for (int j = 1; j < G; ++j) {
        const int so = __builtin_amdgcn_readfirstlane(old);
        const float x1 = p[so * 3 + 0], y1 = p[so * 3 + 1], z1 = p[so * 3 + 2];
        float best = -1.0f;
        int besti = 0;
        for (int k = tid; k < n; k += kThreads) {
            const float dx = p[k * 3] - x1, dy = p[k * 3 + 1] - y1, dz = p[k * 3 + 2] - z1;
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            const bool gt = d2 > best;
            best = gt ? d2 : best;
            besti = gt ? k : besti;
        }
        unsigned long long key = wave_max_u64(fps_key(best, besti));
        const int buf = j & 1;
        if ((tid & 63) == 0) s_key[buf][tid >> 6] = key;
        __syncthreads();
        unsigned long long m = s_key[buf][0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
            const unsigned long long o = s_key[buf][w];
            m = o > m ? o : m;
        }
        old = m == 0ull ? 0 : (int)(0xFFFFFFFFu - (unsigned)(m & 0xFFFFFFFFull));
        if (tid == 0) {
            out[j] = old;
            if (cen) { cen[j * 3 + 0] = p[old * 3 + 0]; cen[j * 3 + 1] = p[old * 3 + 1]; cen[j * 3 + 2] = p[old * 3 + 2]; }
        }
    }
}

#endif  // CMDIAD_AB_VARIANTS

}  // namespace

extern "C" size_t cmdiad_fps_workspace_bytes(int B, int N)
{
    // running minima of the points beyond the register + LDS tiers of fps_ragged_kernel (indexed by point: B * N floats)
    return N > kMaxRegPoints + kMaxLdsPoints ? (size_t)B * (size_t)N * sizeof(float) : 0;
}

extern "C" int cmdiad_fps(const float* xyz, const int32_t* n_valid, int B, int N, int G, int32_t* idx_out,
                          float* center_out, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(xyz && idx_out, CMDIAD_ERR_ARG, "cmdiad_fps: null pointer");
    CMDIAD_REQUIRE(B >= 0 && N > 0 && G >= 0, CMDIAD_ERR_ARG, "cmdiad_fps: bad sizes B=%d N=%d G=%d", B, N, G);
    if (B == 0 || G == 0) return CMDIAD_OK;
    hipStream_t s = (hipStream_t)stream;
#ifdef CMDIAD_AB_VARIANTS
    // test-only build: CMDIAD_FPS_PK=0 selects the first formulation (A/B runs and the parity tests; read per call),
    // CMDIAD_FPS_RAGGED=0 the dispatch on the padded length with the memory-resident loop above 28 672 points
    const char* e = getenv("CMDIAD_FPS_PK");
    const bool pk = !(e && e[0] == '0');
    const char* er = getenv("CMDIAD_FPS_RAGGED");
    const bool ragged_ok = !(er && er[0] == '0');
#define FPS_LAUNCH(T, P)                                                                                                      \
    do {                                                                                                                      \
        if (pk) hipLaunchKernelGGL((fps_pk_kernel<T, P>), dim3(B), dim3(T), 0, s, xyz, n_valid, N, G, idx_out, center_out);   \
        else hipLaunchKernelGGL((fps_reg_kernel<T, P>), dim3(B), dim3(T), 0, s, xyz, n_valid, N, G, idx_out, center_out);     \
    } while (0)
#else
    const bool ragged_ok = true;
#define FPS_LAUNCH(T, P) hipLaunchKernelGGL((fps_pk_kernel<T, P>), dim3(B), dim3(T), 0, s, xyz, n_valid, N, G, idx_out, center_out)
#endif
    // 1024 threads (4 waves/SIMD, 128 VGPRs) hold 16 points per lane; larger clouds use 512 threads
    // (2 waves/SIMD, 256 VGPRs): the register file of ONE CU bounds the resident cloud at ~28k points.
    // Per-cloud lengths (n_valid) above 16 384 padded points, and every cloud above 28 672: fps_ragged_kernel.
    const bool ragged = ragged_ok && N > 1024 * 16 && (n_valid != nullptr || N > kMaxRegPoints);
    if (ragged) {
        const int lds_cap = N > kMaxRegPoints ? (N - kMaxRegPoints < kMaxLdsPoints ? N - kMaxRegPoints : kMaxLdsPoints) : 0;
        const size_t need = cmdiad_fps_workspace_bytes(B, N);
        CMDIAD_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), CMDIAD_ERR_WORKSPACE,
                       "cmdiad_fps: N=%d needs %zu workspace bytes", N, need);
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)fps_ragged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kMaxLdsPoints * (int)sizeof(float4)) != hipSuccess) {
                cmdiad_set_error("cmdiad_fps: hipFuncSetAttribute failed");
                return CMDIAD_ERR_LAUNCH;
            }
            attr = true;
        }
        hipLaunchKernelGGL(fps_ragged_kernel, dim3(B), dim3(kRaggedThreads), (size_t)lds_cap * sizeof(float4), s, xyz, n_valid, N, G,
                           lds_cap, need ? (float*)workspace : nullptr, idx_out, center_out);
    }
    else if (N <= 1024 * 4) FPS_LAUNCH(1024, 4);
    else if (N <= 1024 * 8) FPS_LAUNCH(1024, 8);
    else if (N <= 1024 * 16) FPS_LAUNCH(1024, 16);
    else if (N <= 512 * 40) FPS_LAUNCH(512, 40);
    else if (N <= 512 * 48) FPS_LAUNCH(512, 48);
    else if (N <= 512 * 56) FPS_LAUNCH(512, 56);
#undef FPS_LAUNCH
    else {
#ifdef CMDIAD_AB_VARIANTS
        const size_t need = (size_t)B * (size_t)N * sizeof(float);
        CMDIAD_REQUIRE(workspace && workspace_bytes >= need, CMDIAD_ERR_WORKSPACE, "cmdiad_fps: N=%d needs %zu workspace bytes", N, need);
        hipLaunchKernelGGL(fps_mem_kernel, dim3(B), dim3(kThreads), 0, s, xyz, n_valid, N, G, (float*)workspace, idx_out, center_out);
#else
        cmdiad_set_error("cmdiad_fps: unreachable dispatch (N=%d)", N);
        return CMDIAD_ERR_ARG;
#endif
    }
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
