// Point-cloud -> patch-grid kernels for gfx950:
//   cmdiad_unorganize   reference feature_extractors/multiple_features.py:10-25
//   cmdiad_interp3nn    reference models/pointnet2_utils.py:45-75 (selection + weights)
//   cmdiad_interp_gather  ... :72 (weighted gather; only used to materialise the reference's
//                         [1,D,N] tensor for callers that insist on it)
//   cmdiad_xyz_patch_fused  reference features.py:169-184 fused with the gather above.
//
// The reference materialises interp [D,N] (154 MB), scatters it into a zero [D,224*224] map, then
// runs AvgPool2d(3,1) and AdaptiveAvgPool2d.  All three steps are linear, so the patch feature is
//   out[p][:] = sum_{pixels in p's footprint} c_y(Y) c_x(X) * sum_k w3[pt][k] * F[idx3[pt][k]][:]
// with separable coefficients.  A footprint (<= 7x7 pixels x 3 neighbours) touches only a handful
// of distinct group centres, so the kernel first folds the footprint into a short (centre, weight)
// list in LDS (deterministic order: no float atomics) and then streams those few feature rows
// from L2.  HBM traffic per image: F (3 MB) + idx3/w3 (0.6 MB) + output (9.6 MB) instead of
// ~460 MB for the unfused chain.
//
// interp3nn is compiled with -ffp-contract=off and follows oracle/cmdiad_oracle.c:orc_interp3nn
// operation by operation, so idx3 / w3 are bit-exact against the oracle.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// a1: keep pixels whose x, y, z are all non-zero, in raster order.  One 1024-thread block per image walks its 1024-pixel chunks in
// order with a running count: a chunk = coalesced loads (issued seven chunks ahead, so the walk is not a chain of load latencies),
// wave ballots + a 16-entry LDS scan, one barrier.  Earlier forms: a contiguous run of 49 pixels per thread (196-byte lane stride:
// 0.25 ms for 19 MB at the head of the step's critical path), then a block per (chunk, image) that re-counted everything in front
// of its chunk (6 272 blocks re-reading 450 MB through L2: 0.17-0.25 ms inside the pipelined step).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void unorganize_kernel(const float* __restrict__ pc, int HW, int Nmax,
                                                          float* __restrict__ xyz, int32_t* __restrict__ nz,
                                                          int32_t* __restrict__ pix2pt, int32_t* __restrict__ n_valid)
{
    constexpr int kAhead = 7;
    __shared__ int s_wave[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* px = pc + (size_t)b * 3 * HW;
    const float* py = px + HW;
    const float* pz = py + HW;
    const int nchunks = (HW + 1023) / 1024;
    float cx[kAhead], cy[kAhead], cz[kAhead], nx[kAhead], ny[kAhead], nzv[kAhead];
    auto fetch = [&](int c0, float (&X)[kAhead], float (&Y)[kAhead], float (&Z)[kAhead]) {
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const int i = (c0 + e) * 1024 + tid;
            const bool in = c0 + e < nchunks && i < HW;
            X[e] = in ? px[i] : 0.0f;
            Y[e] = in ? py[i] : 0.0f;
            Z[e] = in ? pz[i] : 0.0f;
        }
    };
    fetch(0, cx, cy, cz);
    int base = 0;
    for (int c0 = 0; c0 < nchunks; c0 += kAhead) {
        fetch(c0 + kAhead, nx, ny, nzv);
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const int c = c0 + e;
            if (c >= nchunks) break;
            const int i = c * 1024 + tid;
            const float x = cx[e], y = cy[e], z = cz[e];
            const bool keep = i < HW && x != 0.0f && y != 0.0f && z != 0.0f;
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_wave[c & 1][wave] = __popcll(m);
            __syncthreads();
            int pos = base + __popcll(m & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int n = s_wave[c & 1][w];
                pos += w < wave ? n : 0;
                total += n;
            }
            base += total;
            if (i < HW) {
                if (pix2pt) pix2pt[(size_t)b * HW + i] = (keep && pos < Nmax) ? pos : -1;
                if (keep && pos < Nmax) {
                    float* o = xyz + ((size_t)b * Nmax + pos) * 3;
                    o[0] = x; o[1] = y; o[2] = z;
                    if (nz) nz[(size_t)b * Nmax + pos] = i;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < kAhead; ++e) { cx[e] = nx[e]; cy[e] = ny[e]; cz[e] = nzv[e]; }
    }
    if (n_valid && tid == 0) n_valid[b] = min(base, Nmax);
}

// ---------------------------------------------------------------------------------------------
// a7 (selection): 3 nearest centres per point, d = -2*dot + |a|^2 + |b|^2, weights 1/(d+1e-8).
// Centres {x,y,z,|c|^2} staged in LDS; every lane scans them with broadcast ds_read_b128.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxCentres = 4096;

__global__ __launch_bounds__(256) void interp3nn_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ n_valid,
                                                        const float* __restrict__ center, int N, int S,
                                                        int32_t* __restrict__ idx3, float* __restrict__ w3)
{
    // centres in LDS as PAIRS, {x0, x1, y0, y1} {z0, z1, |c0|^2, |c1|^2}: the distance arithmetic of two centres per v_pk_*_f32
    // instruction (IEEE per element: the same single roundings in the same order as the scalar form, so d -- and with it idx3 /
    // w3 -- does not change), 8 packed operations per pair instead of 16; an odd S is padded with a centre at +inf distance.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_c = reinterpret_cast<float4*>(smem);
    const int b = blockIdx.y;
    const int n = n_valid ? n_valid[b] : N;
    const float* cb = center + (size_t)b * S * 3;
    const int SP = (S + 1) / 2;
    for (int s = threadIdx.x; s < 2 * SP; s += 256) {
        float x = 0.f, y = 0.f, z = 0.f, w = __builtin_inff();
        if (s < S) { x = cb[s * 3]; y = cb[s * 3 + 1]; z = cb[s * 3 + 2]; w = (x * x + y * y) + z * z; }
        float* q = reinterpret_cast<float*>(s_c + 2 * (s >> 1)) + (s & 1);
        q[0] = x; q[2] = y; q[4] = z; q[6] = w;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = xyz + ((size_t)b * N + i) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const float n1 = (x * x + y * y) + z * z;
    float d0 = __builtin_inff(), d1 = __builtin_inff(), d2 = __builtin_inff();
    int i0 = 0, i1 = 0, i2 = 0;
    // s increases, so strict '<' keeps the lowest index among equal distances; written as selects (the nested-if form of it,
    // inlined twice per pair, sent i0..i2 to scratch)
#define CMDIAD_TOP3_INSERT(d, s)                                               \
    {                                                                          \
        const bool l0 = (d) < d0, l1 = (d) < d1, l2 = (d) < d2;                \
        d2 = l1 ? d1 : (l2 ? (d) : d2); i2 = l1 ? i1 : (l2 ? (s) : i2);        \
        d1 = l0 ? d0 : (l1 ? (d) : d1); i1 = l0 ? i0 : (l1 ? (s) : i1);        \
        d0 = l0 ? (d) : d0;             i0 = l0 ? (s) : i0;                    \
    }
    const f32x2 X = {x, x}, Y = {y, y}, Z = {z, z}, N1 = {n1, n1}, M2 = {-2.0f, -2.0f};
    for (int sp = 0; sp < SP; ++sp) {
        const float4 a = s_c[2 * sp], c = s_c[2 * sp + 1];
        const f32x2 cx = {a.x, a.y}, cy = {a.z, a.w}, cz = {c.x, c.y}, cw = {c.z, c.w};
        const f32x2 dot = (X * cx + Y * cy) + Z * cz;
        f32x2 d = M2 * dot;
        d = d + N1;
        d = d + cw;
        if (fminf(d[0], d[1]) < d2) {   // (a padded centre is +inf: never inserted)
            CMDIAD_TOP3_INSERT(d[0], 2 * sp)
            CMDIAD_TOP3_INSERT(d[1], 2 * sp + 1)
        }
    }
#undef CMDIAD_TOP3_INSERT
    const float r0 = 1.0f / (d0 + 1e-8f);
    const float r1 = S > 1 ? 1.0f / (d1 + 1e-8f) : 0.0f;
    const float r2 = S > 2 ? 1.0f / (d2 + 1e-8f) : 0.0f;
    const float norm = (r0 + r1) + r2;
    const size_t o = ((size_t)b * N + i) * 3;
    idx3[o] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
    w3[o] = r0 / norm; w3[o + 1] = r1 / norm; w3[o + 2] = r2 / norm;
}

// ---------------------------------------------------------------------------------------------
// a7 (selection) as a NEIGHBOURHOOD search (round 6; the counterpart of knn_grid_*_kernel in knn_group.hip): the S centres of a
// cloud are binned into a 16 x 16 grid on the two axes of their largest extent (interp3nn_bin_kernel, one workgroup per cloud:
// counting sort, {x, y, z, |c|^2} + original index per centre), and a point looks at the rings of cells around it, innermost
// first, until its three best are certified -- ~40 distance evaluations per point instead of S = 1 024.
// Exactness.  The three "nearest" centres are the three smallest values of the reference's FORMULA d = -2 a.b + |a|^2 + |b|^2 in
// fp32 (pointnet2_utils.py:19-22), ties to the lowest index; the same operation sequence as interp3nn_kernel gives the same d
// for a (point, centre) pair whatever the visiting order, so the selection only has to see every centre that can matter:
//   * a centre in an unscanned cell differs from the point by more than m cells along a grid axis, so its TRUE squared distance
//     exceeds R^2 = ((m - 0.01) h)^2, and its formula value exceeds R^2 - E, E = 4e-6 (|a|^2 + max |c|^2) bounding the formula's
//     rounding (<= ~10 roundings of magnitude (|a| + |c|)^2 <= 2 (|a|^2 + |c|^2): 1.2e-6 of that sum; E is 3 x the bound);
//   * the search stops once the third best formula value is below R^2 - E (strictly); radius 16 is every centre.
// Coordinates so large that E exceeds the centre spacing (the regime where the reference's own selection is rounding noise)
// simply never certify early and scan everything: slower, still exact.
// ---------------------------------------------------------------------------------------------
constexpr int kCGrid = 16, kCCells = kCGrid * kCGrid, kCHdr = 8;

__device__ __forceinline__ int cgrid_coord(float a, float mn, float inv_h)
{
    return (int)fminf(fmaxf((a - mn) * inv_h, 0.0f), (float)(kCGrid - 1));   // (NaN -> 0; monotone in a)
}
__device__ __forceinline__ float cpick3(float x, float y, float z, int axis) { return axis == 0 ? x : (axis == 1 ? y : z); }

// workspace per cloud: float4 sorted[S] | int orig[S] | int cell_start[257] | float hdr[8]
__device__ __forceinline__ size_t cgrid_stride(int S) { return (size_t)S * 20 + (size_t)(kCCells + 1) * 4 + kCHdr * 4; }

__global__ __launch_bounds__(256) void interp3nn_bin_kernel(const float* __restrict__ center, int S, char* __restrict__ ws, size_t stride)
{
    __shared__ int s_cnt[kCCells];
    __shared__ float s_red[4][7];
    __shared__ int s_wave[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* cb = center + (size_t)b * S * 3;
    char* w = ws + (size_t)b * stride;
    float4* sorted = reinterpret_cast<float4*>(w);
    int* orig = reinterpret_cast<int*>(w + (size_t)S * 16);
    int* cs = orig + S;
    float* hd = reinterpret_cast<float*>(cs + kCCells + 1);
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float cwmax = 0.0f;
    for (int k = tid; k < S; k += 256) {
        const float x = cb[k * 3], y = cb[k * 3 + 1], z = cb[k * 3 + 2];
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        cwmax = fmaxf(cwmax, (x * x + y * y) + z * z);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], m, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], m, 64));
        }
        cwmax = fmaxf(cwmax, __shfl_xor(cwmax, m, 64));
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
        s_red[wave][6] = cwmax;
    }
    s_cnt[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = fminf(fminf(s_red[0][a], s_red[1][a]), fminf(s_red[2][a], s_red[3][a]));
        mx[a] = fmaxf(fmaxf(s_red[0][3 + a], s_red[1][3 + a]), fmaxf(s_red[2][3 + a], s_red[3][3 + a]));
    }
    cwmax = fmaxf(fmaxf(s_red[0][6], s_red[1][6]), fmaxf(s_red[2][6], s_red[3][6]));
    const float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
    int A, Bx;
    if (e0 >= e1 && e0 >= e2) { A = 0; Bx = e1 >= e2 ? 1 : 2; }
    else if (e1 >= e2) { A = 1; Bx = e0 >= e2 ? 0 : 2; }
    else { A = 2; Bx = e0 >= e1 ? 0 : 1; }
    if (A > Bx) { const int t = A; A = Bx; Bx = t; }
    const float ext = fmaxf(cpick3(e0, e1, e2, A), cpick3(e0, e1, e2, Bx));
    const float h = ext > 0.0f && ext < __builtin_inff() ? ext * (1.0f / kCGrid) : 0.0f;
    const float inv_h = h > 0.0f ? 1.0f / h : 0.0f;
    const float mnA = cpick3(mn[0], mn[1], mn[2], A), mnB = cpick3(mn[0], mn[1], mn[2], Bx);
    for (int k = tid; k < S; k += 256) {
        const float x = cb[k * 3], y = cb[k * 3 + 1], z = cb[k * 3 + 2];
        atomicAdd(&s_cnt[cgrid_coord(cpick3(x, y, z, Bx), mnB, inv_h) * kCGrid + cgrid_coord(cpick3(x, y, z, A), mnA, inv_h)], 1);
    }
    __syncthreads();
    const int mine = s_cnt[tid];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wave; ++q) base += s_wave[q];
    const int start = base + incl - mine;
    cs[tid] = start;
    if (tid == 255) cs[kCCells] = start + mine;
    __syncthreads();          // (every thread has read its count)
    s_cnt[tid] = start;       // now the cell's write cursor
    if (tid == 0) { hd[0] = mnA; hd[1] = mnB; hd[2] = inv_h; hd[3] = h; hd[4] = (float)A; hd[5] = (float)Bx; hd[6] = cwmax; hd[7] = 0.0f; }
    __syncthreads();
    for (int k = tid; k < S; k += 256) {
        const float x = cb[k * 3], y = cb[k * 3 + 1], z = cb[k * 3 + 2];
        const int cell = cgrid_coord(cpick3(x, y, z, Bx), mnB, inv_h) * kCGrid + cgrid_coord(cpick3(x, y, z, A), mnA, inv_h);
        const int pos = atomicAdd(&s_cnt[cell], 1);
        sorted[pos] = float4{x, y, z, (x * x + y * y) + z * z};
        orig[pos] = k;
    }
}

__global__ __launch_bounds__(256) void interp3nn_grid_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ n_valid,
                                                             const char* __restrict__ ws, size_t stride, int N, int S,
                                                             int32_t* __restrict__ idx3, float* __restrict__ w3)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_c = reinterpret_cast<float4*>(smem);
    int* s_i = reinterpret_cast<int*>(smem + (size_t)S * 16);
    int* s_cs = s_i + S;
    const int b = blockIdx.y;
    const int n = n_valid ? n_valid[b] : N;
    const char* w = ws + (size_t)b * stride;
    {
        const float4* g_c = reinterpret_cast<const float4*>(w);
        const int* g_i = reinterpret_cast<const int*>(w + (size_t)S * 16);
        for (int k = threadIdx.x; k < S; k += 256) { s_c[k] = g_c[k]; s_i[k] = g_i[k]; }
        for (int k = threadIdx.x; k < kCCells + 1; k += 256) s_cs[k] = g_i[S + k];
    }
    const float* hd = reinterpret_cast<const float*>(w + (size_t)S * 20 + (size_t)(kCCells + 1) * 4);
    const float mnA = hd[0], mnB = hd[1], inv_h = hd[2], h = hd[3], cwmax = hd[6];
    const int A = (int)hd[4], Bx = (int)hd[5];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = xyz + ((size_t)b * N + i) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const float n1 = (x * x + y * y) + z * z;
    const int ia = cgrid_coord(cpick3(x, y, z, A), mnA, inv_h), ib = cgrid_coord(cpick3(x, y, z, Bx), mnB, inv_h);
    float d0 = __builtin_inff(), d1 = __builtin_inff(), d2 = __builtin_inff();
    int i0 = 0, i1 = 0, i2 = 0;
    // (d, index) lexicographic order: the lowest index among equal distances, whatever the visiting order
#define CMDIAD_TOP3_LEX(d, s)                                                                        \
    {                                                                                                \
        const bool l0 = (d) < d0 || ((d) == d0 && (s) < i0), l1 = (d) < d1 || ((d) == d1 && (s) < i1),      \
                   l2 = (d) < d2 || ((d) == d2 && (s) < i2);                                          \
        d2 = l1 ? d1 : (l2 ? (d) : d2); i2 = l1 ? i1 : (l2 ? (s) : i2);                              \
        d1 = l0 ? d0 : (l1 ? (d) : d1); i1 = l0 ? i0 : (l1 ? (s) : i1);                              \
        d0 = l0 ? (d) : d0;             i0 = l0 ? (s) : i0;                                          \
    }
    auto run = [&](int s, int e) {
        for (int k = s; k < e; ++k) {
            const float4 c = s_c[k];
            const float dot = (x * c.x + y * c.y) + z * c.z;
            float d = -2.0f * dot;
            d = d + n1;
            d = d + c.w;
            if (d <= d2) {               // (NaN: never)
                const int si = s_i[k];
                CMDIAD_TOP3_LEX(d, si)
            }
        }
    };
    const float E = 4e-6f * (n1 + cwmax);
    int m_done = -1, m = h > 0.0f ? 1 : kCGrid;
    for (;;) {
        m = min(m, kCGrid);
        for (int dj = -m; dj <= m; ++dj) {
            const int j = ib + dj;
            if (j < 0 || j >= kCGrid) continue;
            const int lo = max(ia - m, 0), hi = min(ia + m, kCGrid - 1);
            if (dj < -m_done || dj > m_done || m_done < 0) {
                run(s_cs[j * kCGrid + lo], s_cs[j * kCGrid + hi + 1]);
            } else {
                if (ia - m_done - 1 >= lo) run(s_cs[j * kCGrid + lo], s_cs[j * kCGrid + ia - m_done]);
                if (ia + m_done + 1 <= hi) run(s_cs[j * kCGrid + ia + m_done + 1], s_cs[j * kCGrid + hi + 1]);
            }
        }
        m_done = m;
        if (m >= kCGrid) break;
        const float r = ((float)m - 0.01f) * h;
        if (d2 < r * r - E) break;      // three centres met (d2 finite) and nothing outside the scanned square can beat the third
        m = d2 < __builtin_inff() ? max(m + 1, (int)(sqrtf(fmaxf(d2 + E, 0.0f)) * inv_h) + 2) : m * 2;
    }
#undef CMDIAD_TOP3_LEX
    const float r0 = 1.0f / (d0 + 1e-8f);
    const float r1 = S > 1 ? 1.0f / (d1 + 1e-8f) : 0.0f;
    const float r2 = S > 2 ? 1.0f / (d2 + 1e-8f) : 0.0f;
    const float norm = (r0 + r1) + r2;
    const size_t o = ((size_t)b * N + i) * 3;
    idx3[o] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
    w3[o] = r0 / norm; w3[o + 1] = r1 / norm; w3[o + 2] = r2 / norm;
}

// out[b][n][:] = (F[i0]*w0 + F[i1]*w1) + F[i2]*w2 ; one wave per point, 16-byte accesses.
__global__ __launch_bounds__(256) void interp_gather_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx3,
                                                            const float* __restrict__ w3,
                                                            const int32_t* __restrict__ n_valid, int N, int S, int D,
                                                            float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int n = n_valid ? n_valid[b] : N;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const size_t o = ((size_t)b * N + i) * 3;
    const float* f0 = feat + ((size_t)b * S + idx3[o]) * D;
    const float* f1 = feat + ((size_t)b * S + idx3[o + 1]) * D;
    const float* f2 = feat + ((size_t)b * S + idx3[o + 2]) * D;
    const float w0 = w3[o], w1 = w3[o + 1], w2 = w3[o + 2];
    float* dst = out + ((size_t)b * N + i) * D;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(f0 + c);
        const float4 bb = *reinterpret_cast<const float4*>(f1 + c);
        const float4 cc = *reinterpret_cast<const float4*>(f2 + c);
        float4 r;
        r.x = (a.x * w0 + bb.x * w1) + cc.x * w2;
        r.y = (a.y * w0 + bb.y * w1) + cc.y * w2;
        r.z = (a.z * w0 + bb.z * w1) + cc.z * w2;
        r.w = (a.w * w0 + bb.w * w1) + cc.w * w2;
        *reinterpret_cast<float4*>(dst + c) = r;
    }
}

// ---------------------------------------------------------------------------------------------
// a7 (gather) + a9 fused: one NT-thread block per output patch.
// ---------------------------------------------------------------------------------------------
template <int NT>   // threads per patch: see cmdiad_xyz_patch_fused
__global__ __launch_bounds__(NT) void xyz_patch_fused_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx3,
                                                              const float* __restrict__ w3,
                                                              const int32_t* __restrict__ pix2pt, int B, int N, int S, int D,
                                                              int size, int P, float mean, float inv_std,
                                                              float* __restrict__ out_f32, bf16_t* __restrict__ out_bf16)
{
    constexpr int kMaxEnt = 3 * 12 * 12;
    __shared__ __attribute__((aligned(16))) int s_g[kMaxEnt];
    __shared__ __attribute__((aligned(16))) float s_w[kMaxEnt];
    __shared__ __attribute__((aligned(16))) int s_lg[kMaxEnt];
    __shared__ __attribute__((aligned(16))) float s_lw[kMaxEnt];
    __shared__ int s_cnt;

    // XCD-aware ids (workgroup L runs on XCD L % 8): every patch of image b gets the same L % 8, so that image's
    // [S, D] centre features (3 MB) are gathered through ONE L2 instead of all eight.
    const int slot = blockIdx.x >> 3, PP = P * P;
    const int b = (slot / PP) * 8 + (blockIdx.x & 7), patch = slot % PP;
    if (b >= B) return;
    const int py = patch / P, px = patch % P;
    const int L = size - 2;
    const int y0 = (py * L) / P, y1 = ((py + 1) * L + P - 1) / P;
    const int x0 = (px * L) / P, x1 = ((px + 1) * L + P - 1) / P;
    const int fh = y1 - y0 + 2, fw = x1 - x0 + 2;  // footprint in the size x size map
    const int ne = fh * fw * 3;
    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;

    // entry e = (pixel, neighbour k): centre index and combined coefficient
    for (int e = tid; e < ne; e += NT) {
        const int k = e % 3, pix = e / 3;
        const int fy = pix / fw, fx = pix % fw;
        const int Y = y0 + fy, X = x0 + fx;
        // number of 3-windows starting in [y0,y1) that cover row Y: starts y in [max(y0,Y-2), min(y1-1,Y)]
        const int cy = min(y1 - 1, Y) - max(y0, Y - 2) + 1;
        const int cx = min(x1 - 1, X) - max(x0, X - 2) + 1;
        const float coef = ((float)cy / (3.0f * (float)(y1 - y0))) * ((float)cx / (3.0f * (float)(x1 - x0)));
        const int pt = pix2pt[(size_t)b * size * size + (size_t)Y * size + X];
        int g = -1;
        float w = 0.0f;
        if (pt >= 0) {
            g = idx3[((size_t)b * N + pt) * 3 + k];
            w = coef * w3[((size_t)b * N + pt) * 3 + k];
        }
        s_g[e] = g; s_w[e] = w;
    }
    const int ne4 = (ne + 3) & ~3;
    for (int e = ne + tid; e < ne4; e += NT) { s_g[e] = -1; s_w[e] = 0.0f; }   // pad to a multiple of four
    __syncthreads();
    // deterministic fold: the first entry of every distinct centre sums all its entries in index order.  Branch-free over the
    // whole list, four entries per LDS read: the earlier form (a data-dependent `break` loop per entry, then a second loop for
    // the sums) was a chain of up to 2 x 108 dependent LDS round trips per thread -- most of the kernel's time
    for (int e = tid; e < ne; e += NT) {
        const int g = s_g[e];
        if (g < 0) continue;
        bool first = true;
        float acc = s_w[e];
        for (int j = 0; j < ne4; j += 4) {
            const int4 gj = *reinterpret_cast<const int4*>(s_g + j);
            const float4 wj = *reinterpret_cast<const float4*>(s_w + j);
            const int gs[4] = {gj.x, gj.y, gj.z, gj.w};
            const float ws[4] = {wj.x, wj.y, wj.z, wj.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool same = gs[u] == g;
                first = first && !(same && j + u < e);
                acc = (same && j + u > e) ? acc + ws[u] : acc;
            }
        }
        if (first) {
            const int slot = atomicAdd(&s_cnt, 1);
            s_lg[slot] = g; s_lw[slot] = acc;
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    // order the short list by centre index so the channel sums are run-to-run reproducible: rank sort, one thread per
    // entry (centres are distinct after the fold), into the entry arrays that are no longer needed
    const int cnt4 = (cnt + 3) & ~3;
    for (int e = cnt + tid; e < cnt4; e += NT) s_lg[e] = 0x7FFFFFFF;
    __syncthreads();
    for (int e = tid; e < cnt; e += NT) {
        const int g = s_lg[e];
        int rank = 0;
        for (int j = 0; j < cnt4; j += 4) {
            const int4 gj = *reinterpret_cast<const int4*>(s_lg + j);
            rank += (gj.x < g) + (gj.y < g) + (gj.z < g) + (gj.w < g);
        }
        s_g[rank] = g; s_w[rank] = s_lw[e];
    }
    __syncthreads();

    const size_t orow = ((size_t)b * P * P + patch) * D;
    for (int c = tid * 4; c < D; c += NT * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight centre rows in flight per thread (same summation order; sixteen measured slower: 0.65 against 0.45 ms): one row per iteration made the block a chain of
        // ~30 dependent L2 round trips -- 15 us per patch, the whole kernel (100 352 patches, eight blocks per CU) 0.71 ms
        const float* fb = feat + (size_t)b * S * D + c;
        for (int i0 = 0; i0 < cnt; i0 += 8) {
            float4 f[8];
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, cnt - 1);
                w[u] = s_w[i];
                f[u] = *reinterpret_cast<const float4*>(fb + (size_t)s_g[i] * D);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < cnt) { acc.x += w[u] * f[u].x; acc.y += w[u] * f[u].y; acc.z += w[u] * f[u].z; acc.w += w[u] * f[u].w; }
        }
        acc.x = (acc.x - mean) * inv_std; acc.y = (acc.y - mean) * inv_std;
        acc.z = (acc.z - mean) * inv_std; acc.w = (acc.w - mean) * inv_std;
        if (out_f32) *reinterpret_cast<float4*>(out_f32 + orow + c) = acc;
        if (out_bf16) {
            bf16x4 o = {f2bf(acc.x), f2bf(acc.y), f2bf(acc.z), f2bf(acc.w)};
            *reinterpret_cast<bf16x4*>(out_bf16 + orow + c) = o;
        }
    }
}

}  // namespace

extern "C" int cmdiad_unorganize(const float* organized_pc, int B, int HW, int Nmax, float* xyz, int32_t* nz,
                                 int32_t* pix2pt, int32_t* n_valid, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(organized_pc && xyz && B > 0 && HW > 0 && Nmax > 0, CMDIAD_ERR_ARG, "cmdiad_unorganize: bad args");
    hipLaunchKernelGGL(unorganize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, organized_pc, HW, Nmax, xyz,
                       nz, pix2pt, n_valid);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_interp3nn(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int S,
                                int32_t* idx3, float* w3, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(xyz && center && idx3 && w3, CMDIAD_ERR_ARG, "cmdiad_interp3nn: null pointer");
    CMDIAD_REQUIRE(B > 0 && N > 0 && S > 0 && S <= kMaxCentres, CMDIAD_ERR_ARG, "cmdiad_interp3nn: need 0<S<=%d (S=%d)",
                   kMaxCentres, S);
    dim3 grid((N + 255) / 256, B);
    hipLaunchKernelGGL(interp3nn_kernel, grid, dim3(256), (size_t)((S + 1) / 2) * 2 * sizeof(float4), (hipStream_t)stream, xyz,
                       n_valid, center, N, S, idx3, w3);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

// The same selection through the centre grid (interp3nn_bin_kernel + interp3nn_grid_kernel).  Falls back to interp3nn_kernel for
// few centres (S < 64) and with CMDIAD_INTERP_GRID=0 (A/B runs, parity tests; read per call).
extern "C" size_t cmdiad_interp3nn_workspace_bytes(int B, int S)
{
    if (B <= 0 || S <= 0) return 0;
    return (size_t)B * (((size_t)S * 20 + (size_t)(kCCells + 1) * 4 + kCHdr * 4 + 15) / 16 * 16);
}

extern "C" int cmdiad_interp3nn_ws(const float* xyz, const int32_t* n_valid, const float* center, int B, int N, int S,
                                   int32_t* idx3, float* w3, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    const char* e = getenv("CMDIAD_INTERP_GRID");
    if ((e && e[0] == '0') || S < 64 || S > kMaxCentres || B <= 0 || N <= 0)
        return cmdiad_interp3nn(xyz, n_valid, center, B, N, S, idx3, w3, stream);
    CMDIAD_REQUIRE(xyz && center && idx3 && w3, CMDIAD_ERR_ARG, "cmdiad_interp3nn_ws: null pointer");
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_interp3nn_workspace_bytes(B, S) && ((uintptr_t)workspace & 15) == 0,
                   CMDIAD_ERR_WORKSPACE, "cmdiad_interp3nn_ws: workspace too small or not 16-byte aligned");
    const size_t stride = ((size_t)S * 20 + (size_t)(kCCells + 1) * 4 + kCHdr * 4 + 15) / 16 * 16;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(interp3nn_bin_kernel, dim3(B), dim3(256), 0, s, center, S, (char*)workspace, stride);
    const size_t lds = (size_t)S * 20 + (size_t)(kCCells + 1) * 4;
    hipLaunchKernelGGL(interp3nn_grid_kernel, dim3((N + 255) / 256, B), dim3(256), lds, s, xyz, n_valid, (const char*)workspace, stride, N,
                       S, idx3, w3);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_interp_gather(const float* feat, const int32_t* idx3, const float* w3, const int32_t* n_valid,
                                    int B, int N, int S, int D, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(feat && idx3 && w3 && out, CMDIAD_ERR_ARG, "cmdiad_interp_gather: null pointer");
    CMDIAD_REQUIRE(B > 0 && N > 0 && D % 4 == 0 && (((uintptr_t)feat | (uintptr_t)out) & 15) == 0, CMDIAD_ERR_ARG,
                   "cmdiad_interp_gather: D%%4==0 and 16-byte alignment");
    dim3 grid((N + 3) / 4, B);
    hipLaunchKernelGGL(interp_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, feat, idx3, w3, n_valid, N, S, D, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_xyz_patch_fused(const float* feat, const int32_t* idx3, const float* w3, const int32_t* pix2pt,
                                      int B, int N, int S, int D, int size, int P, float mean, float inv_std,
                                      float* patch_f32, uint16_t* patch_bf16, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(feat && idx3 && w3 && pix2pt && (patch_f32 || patch_bf16), CMDIAD_ERR_ARG,
                   "cmdiad_xyz_patch_fused: null pointer");
    CMDIAD_REQUIRE(B > 0 && size > 2 && P > 0 && D % 4 == 0, CMDIAD_ERR_ARG, "cmdiad_xyz_patch_fused: bad sizes");
    const int L = size - 2;
    const int maxbin = (L + P - 1) / P + 1;  // adaptive bins are at most ceil(L/P)+1 wide
    CMDIAD_REQUIRE(maxbin + 2 <= 12, CMDIAD_ERR_ARG, "cmdiad_xyz_patch_fused: footprint %d exceeds 12 (size=%d P=%d)",
                   maxbin + 2, size, P);
    CMDIAD_REQUIRE((((uintptr_t)feat | (uintptr_t)patch_f32) & 15) == 0 && ((uintptr_t)patch_bf16 & 7) == 0, CMDIAD_ERR_ARG,
                   "cmdiad_xyz_patch_fused: alignment");
    dim3 grid((unsigned)((B + 7) / 8 * 8 * P * P));
    // One workgroup per patch is a chain of dependent phases (pixel -> point -> centres, fold, rank sort, row gather: ~7 us) that
    // only other resident workgroups hide: the fewer waves a patch takes, the more patches a CU has in flight (32 wave slots).
    // Batch 32, 224 x 224, 56 x 56 patches, D = 768 (tools/xyz_patch_time.py, identical bits): 256 threads per patch 0.386-0.403 ms,
    // 128: 0.322-0.350, 64: 0.326-0.355.
    static const int nt = getenv("CMDIAD_XYZ_PATCH_THREADS") ? atoi(getenv("CMDIAD_XYZ_PATCH_THREADS")) : 128;
    auto kern = nt == 64 ? xyz_patch_fused_kernel<64> : nt == 128 ? xyz_patch_fused_kernel<128> : xyz_patch_fused_kernel<256>;
    hipLaunchKernelGGL(kern, grid, dim3(nt == 64 ? 64 : nt == 128 ? 128 : 256), 0, (hipStream_t)stream, feat, idx3, w3, pix2pt, B, N, S, D,
                       size, P, mean, inv_std, patch_f32, (bf16_t*)patch_bf16);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
