// Re-weighting scan of compute_single_s_s_map (feature_extractors/features.py:235-254: cdist(m_star, bank) -> 3 smallest)
// for up to 32 probe rows per pass over the library, plus the two small helpers that go with it.
//
// HBM-bound by design: the fp32 library (Nb x D x 4 bytes, 235 MB for bagel-xyz) is streamed ONCE for all probes.  The
// round-1 kernel did the 32 x Nb x D products on the VALU with a 64-lane butterfly per (row, probe) and reached 0.35 TB/s;
// here the cross term runs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: 32 probes x 16 library rows per wave, 24 us of
// MFMA for the whole bagel library) so the kernel is limited by the stream.
//
//   * The library is read from a copy laid out for the MFMA B operand ("block16", cmdiad_bank_block16): groups of 16 rows,
//     [group][t = k/16][kq][row j][4 floats] with k = 16 t + 4 kq + e -- lane l = 16 kq + j of a wave reads 16 bytes at
//     offset 16 l of a contiguous 1 KiB piece, so every global load instruction is one fully coalesced 1 KiB burst and no
//     LDS staging, transposition or barrier sits on the stream.  (288 GB of HBM: a second fp32 copy of a 235 MB library is
//     free.)  The probes sit in LDS in the mirrored layout (A operand).
//   * d2 ~ |p|^2 + |b|^2 - 2 p.b in fp32 (library row norms accumulated from the streamed values, probe norms in the
//     prologue): absolute error ~1e-4 on d2 ~ 1e3.  The scan keeps the EIGHT smallest approximate keys per probe
//     ((d2 bits) << 32 | row: ties and exact duplicates resolve to the lowest row like torch.topk on the exact matrix; a lane
//     keeps four per (probe, row residue) slot -- 32 768 disjoint row classes -- while it streams), and the merge kernel
//     re-evaluates those eight EXACTLY (sum (a-b)^2 in fp32 on the row-major library) before the three smallest exact keys
//     are written: the result is what an exact scan returns unless more than FIVE other rows lie within the approximation
//     error of the third-smallest distance (libraries built without a coreset hold clusters of near-identical patches:
//     tests/test_gpu_kernels.py::test_reweight_scan_clustered_near_duplicates).
#include "common.h"

namespace {

constexpr int kProbes = 32;      // probes per pass (two 16-row MFMA A operands)
constexpr int kLane = 4;         // approximate candidates a lane keeps per (probe, row residue) slot while it streams
constexpr int kCand = 8;         // approximate candidates kept per probe from the wave merge on, all re-evaluated exactly
constexpr int kWaves = 8;        // waves per block: 2 per SIMD
constexpr int kChunk = 8;        // k-steps (of 16 floats) per software-pipeline chunk: D % 128 == 0

static inline bool aligned16h(const void* p) { return ((uintptr_t)p & 15) == 0; }

// sorted ascending list of K keys, insertion of a key known to be < t[K-1]
template <int K>
__device__ __forceinline__ void list_insert(unsigned long long (&t)[K], unsigned long long k)
{
#pragma unroll
    for (int i = K - 1; i > 0; --i) {
        const unsigned long long lo = t[i - 1];
        t[i] = k < lo ? lo : (k < t[i] ? k : t[i]);
    }
    t[0] = k < t[0] ? k : t[0];
}

__device__ __forceinline__ unsigned long long min_u64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }

// minimum over the 16 lanes of a DPP row, in all 16 (quad xor 1, quad xor 2, half-row mirror, row mirror on both halves of
// the key): the 32 dependent pop rounds of the candidate merge below were 256 ds_bpermute round trips at the end of every block
__device__ __forceinline__ unsigned long long row16_min_u64(unsigned long long v)
{
#define CMDIAD_DPP_MIN_STEP(CTRL)                                                                                          \
    {                                                                                                                       \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, false);               \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);       \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                                  \
        v = o < v ? o : v;                                                                                                  \
    }
    CMDIAD_DPP_MIN_STEP(0xB1)
    CMDIAD_DPP_MIN_STEP(0x4E)
    CMDIAD_DPP_MIN_STEP(0x141)
    CMDIAD_DPP_MIN_STEP(0x140)
#undef CMDIAD_DPP_MIN_STEP
    return v;
}

// [Nb, D] row-major -> block16 layout (rows >= Nb read as zero; they are masked by row index in the scan)
__global__ __launch_bounds__(256) void bank_block16_kernel(const float* __restrict__ bank, int Nb, int D,
                                                           float* __restrict__ out, size_t n4)
{
    const int T = D >> 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int l = (int)(i & 63);
        const size_t gt = i >> 6;
        const int t = (int)(gt % T);
        const size_t g = gt / T;
        const int j = l & 15, kq = l >> 4;
        const size_t row = g * 16 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < (size_t)Nb) v = *reinterpret_cast<const float4*>(bank + row * D + t * 16 + kq * 4);
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// One scan problem: R probes against one library.  A launch carries one or TWO of them (cmdiad_reweight_scan_pair: the two
// libraries of a scored batch -- features.py:235-254 runs once per library and sample); the grid's first `blocks` workgroups
// belong to problem 0, the rest to problem 1.
struct ScanProblem {
    const float* probes;
    const float* blocked;      // cmdiad_bank_block16 copy
    const float* bank;         // row-major fp32 library (the exact re-evaluation of the merge kernel)
    int R, Nb;
    unsigned row_offset;
    unsigned long long* partial;
    unsigned long long* top3;
    int blocks;                // workgroups of the scan kernel that stream this library
};
struct ScanPair { ScanProblem p[2]; };

// One wave = 16 library rows x 32 probes at a time.  Groups are dealt to SIMD slots first (slot = CU-major, 4 per CU) and
// alternate between the slot's two waves, so every SIMD gets the same MFMA work to within one group.
__global__ __launch_bounds__(kWaves * 64) void reweight_scan_mfma_kernel(ScanPair pair, int D)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // block-uniform: which library this workgroup streams, and its place among that library's workgroups
    const bool second = (int)blockIdx.x >= pair.p[0].blocks;
    const ScanProblem& pb = pair.p[second ? 1 : 0];
    const float* __restrict__ probes = pb.probes;
    const float* __restrict__ blocked = pb.blocked;
    unsigned long long* __restrict__ partial = pb.partial;
    const int R = pb.R, Nb = pb.Nb;
    const unsigned row_offset = pb.row_offset;
    const int bid = (int)blockIdx.x - (second ? pair.p[0].blocks : 0), nblk = pb.blocks;
    const int T = D >> 4;                                       // 16-float k-steps
    float4* s_probe = reinterpret_cast<float4*>(smem);          // [2][T][64] float4, A-operand layout
    float* s_pn = reinterpret_cast<float*>(smem + (size_t)2 * T * 64 * 16);                  // [32] probe norms
    unsigned long long* s_cand = reinterpret_cast<unsigned long long*>(s_pn + kProbes);      // [waves][32][kCand]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // ---- prologue: probes into LDS (zero rows beyond R), their norms
    for (int i = tid; i < 2 * T * 64; i += kWaves * 64) {
        const int l = i & 63, t = (i >> 6) % T, pg = (i >> 6) / T;
        const int p = pg * 16 + (l & 15), kq = l >> 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < R) v = *reinterpret_cast<const float4*>(probes + (size_t)p * D + t * 16 + kq * 4);
        s_probe[i] = v;
    }
    if (wave < 4) {
        for (int p = wave; p < kProbes; p += 4) {
            float s = 0.f;
            if (p < R)
                for (int c = lane * 4; c < D; c += 256) {
                    const float4 x = *reinterpret_cast<const float4*>(probes + (size_t)p * D + c);
                    s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
                }
            s = wave_sum(s);
            if (lane == 0) s_pn[p] = s;
        }
    }
    __syncthreads();

    // probe of accumulator (pg, v) in this lane: 16 pg + 4 (lane / 16) + v; library row of the lane: lane % 16
    float pn[2][4];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
        for (int v = 0; v < 4; ++v) pn[pg][v] = s_pn[pg * 16 + (lane >> 4) * 4 + v];
    unsigned long long top[2][4][kLane];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int k = 0; k < kLane; ++k) top[pg][v][k] = ~0ull;

    const int groups = (Nb + 15) >> 4;
    const int slots = nblk * 4;
    const int slot = bid * 4 + (wave & 3);
    // the slot's groups are slot, slot + slots, ...; its two waves take them alternately
    const int first = slot + (wave >> 2) * slots;
    const int stride = 2 * slots;
    const size_t gfloat4 = (size_t)T * 64;                       // float4 per group
    const float4* src = reinterpret_cast<const float4*>(blocked) + lane;

    // Flattened stream of this wave's (group, chunk) pieces, two register buffers used alternately: while one chunk (8 KiB)
    // feeds 64 MFMAs the next one is in flight.  Loads are unconditional (positions past the end are clamped to the last
    // piece) and the loop body is straight-line, so the compiler's counted s_waitcnt leaves the younger buffer's loads in
    // flight across the older buffer's MFMAs.
    const int chunks = T / kChunk;
    const int my_groups = first < groups ? (groups - first + stride - 1) / stride : 0;
    const int total = __builtin_amdgcn_readfirstlane(my_groups * chunks);
    const int g_first = __builtin_amdgcn_readfirstlane(first), g_stride = __builtin_amdgcn_readfirstlane(stride);
    float4 bufA[kChunk], bufB[kChunk];
    int lg = g_first, lc = 0, li = 0;   // next piece to LOAD
    auto load = [&](float4 (&buf)[kChunk]) {
        const float4* s2 = src + (size_t)lg * gfloat4 + (size_t)lc * (kChunk * 64);
#pragma unroll
        for (int u = 0; u < kChunk; ++u) buf[u] = s2[(size_t)u * 64];
        if (li + 1 < total) {  // past the end: stay on the last piece
            ++li;
            if (++lc == chunks) { lc = 0; lg += g_stride; }
        }
    };
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float bn = 0.f;
    int cg = g_first, cc = 0;           // piece being COMPUTED
    auto compute = [&](const float4 (&buf)[kChunk]) {
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
            const int t = cc * kChunk + u;
            const float4 a0 = s_probe[(size_t)t * 64 + lane];
            const float4 a1 = s_probe[(size_t)(T + t) * 64 + lane];
            const float4 b = buf[u];
            bn += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc[1], 0, 0, 0);
        }
        if (++cc == chunks) {  // the group is complete: distances of its 16 rows to the 32 probes
            // |row|^2: the four k-quarter lanes of a row (lane, lane^16, lane^32, lane^48) hold its partial sums
            bn += __shfl_xor(bn, 16, 64);
            bn += __shfl_xor(bn, 32, 64);
            const unsigned row = (unsigned)cg * 16u + (unsigned)(lane & 15);
            if (row < (unsigned)Nb) {
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float d2 = fmaxf(pn[pg][v] + bn - 2.0f * acc[pg][v], 0.0f);
                        const unsigned long long key = pack_key(d2, row_offset + row);
                        if (key < top[pg][v][kLane - 1]) list_insert<kLane>(top[pg][v], key);
                    }
            }
            acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            bn = 0.f;
            cc = 0;
            cg += g_stride;
        }
    };
    if (total > 0) {
        load(bufA);
        for (int i = 0; i < total; i += 2) {
            load(bufB);
            compute(bufA);
            load(bufA);
            if (i + 1 < total) compute(bufB);
        }
    }

    // ---- wave: the 16 lanes that share a probe set (equal lane / 16) -> the kCand smallest keys of each probe
    unsigned long long* my = s_cand + (size_t)wave * kProbes * kCand;
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int r = 0; r < kCand; ++r) {
                const unsigned long long best = row16_min_u64(top[pg][v][0]);
                if (top[pg][v][0] == best && best != ~0ull) {  // keys are unique (they carry the row): the owner pops
#pragma unroll
                    for (int k = 0; k + 1 < kLane; ++k) top[pg][v][k] = top[pg][v][k + 1];
                    top[pg][v][kLane - 1] = ~0ull;
                }
                if ((lane & 15) == 0) my[(pg * 16 + (lane >> 4) * 4 + v) * kCand + r] = best;
            }
        }
    __syncthreads();
    // ---- block: 8 waves x kCand -> kCand per probe
    for (int p = tid; p < kProbes; p += kWaves * 64) {
        unsigned long long m[kCand];
#pragma unroll
        for (int k = 0; k < kCand; ++k) m[k] = ~0ull;
        for (int w = 0; w < kWaves; ++w)
#pragma unroll
            for (int k = 0; k < kCand; ++k) {
                const unsigned long long key = s_cand[((size_t)w * kProbes + p) * kCand + k];
                if (key < m[kCand - 1]) list_insert<kCand>(m, key);
            }
        unsigned long long* o = partial + ((size_t)p * nblk + bid) * kCand;
#pragma unroll
        for (int k = 0; k < kCand; ++k) o[k] = m[k];
    }
}

// One block per probe: the kCand smallest approximate keys over all blocks, their EXACT fp32 squared distances, and the
// three smallest exact keys merged into top3 (which may already hold another shard's / an earlier call's keys).
__global__ __launch_bounds__(kCand * 64) void reweight_exact_merge_kernel(ScanPair pair, int D)
{
    __shared__ unsigned long long s_c[kCand];
    __shared__ unsigned long long s_exact[kCand];
    const bool second = (int)blockIdx.x >= pair.p[0].R;       // block-uniform: probes of problem 0 first
    const ScanProblem& pb = pair.p[second ? 1 : 0];
    const unsigned long long* __restrict__ partial = pb.partial;
    const float* __restrict__ probes = pb.probes;
    const float* __restrict__ bank = pb.bank;
    unsigned long long* __restrict__ top3 = pb.top3;
    const int nblocks = pb.blocks;
    const unsigned row_offset = pb.row_offset;
    const int pr = (int)blockIdx.x - (second ? pair.p[0].R : 0), wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        unsigned long long m[kCand];
#pragma unroll
        for (int k = 0; k < kCand; ++k) m[k] = ~0ull;
        const unsigned long long* src = partial + (size_t)pr * nblocks * kCand;
        for (int i = lane; i < nblocks * kCand; i += 64) {
            const unsigned long long key = src[i];
            if (key < m[kCand - 1]) list_insert<kCand>(m, key);
        }
#pragma unroll
        for (int r = 0; r < kCand; ++r) {
            unsigned long long best = m[0];
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) best = min_u64(best, shfl_xor_u64(best, sft));
            if (m[0] == best && best != ~0ull) {
#pragma unroll
                for (int k = 0; k + 1 < kCand; ++k) m[k] = m[k + 1];
                m[kCand - 1] = ~0ull;
            }
            if (lane == 0) s_c[r] = best;
        }
    }
    __syncthreads();
    {
        const unsigned long long key = s_c[wave];
        unsigned long long exact = ~0ull;
        if (key != ~0ull) {
            const unsigned gi = (unsigned)(key & 0xFFFFFFFFull);
            const float* a = probes + (size_t)pr * D;
            const float* b = bank + (size_t)(gi - row_offset) * D;
            float s = 0.0f;
            for (int c = lane * 4; c < D; c += 256) {
                const float4 x = *reinterpret_cast<const float4*>(a + c);
                const float4 y = *reinterpret_cast<const float4*>(b + c);
                const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
                s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
            s = wave_sum(s);
            exact = pack_key(s, gi);
        }
        if (lane == 0) s_exact[wave] = exact;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t[3] = {top3[pr * 3], top3[pr * 3 + 1], top3[pr * 3 + 2]};
        // the "nothing yet" value of the caller may be either all-ones or the signed-safe sentinel; both are larger than any key
#pragma unroll
        for (int k = 0; k < kCand; ++k)
            if (s_exact[k] < t[2]) list_insert<3>(t, s_exact[k]);
        top3[pr * 3] = t[0]; top3[pr * 3 + 1] = t[1]; top3[pr * 3 + 2] = t[2];
    }
}

// Exact fp32 distance matrix out[q][n] = || Q[q] - Bk[n] ||_2 (features.py:186-190 as a MATERIALISED matrix: API compatibility
// of calculate_dist only -- the product path never builds it).  64 x 64 tile per block, K staged through LDS in 32-float
// slices, every thread a 4 x 4 patch of sum (a - b)^2 (differences, not the norm expansion: no cancellation).
__global__ __launch_bounds__(256) void l2_dist_matrix_kernel(const float* __restrict__ q, const float* __restrict__ bank,
                                                             int Q, int Nb, int D, float* __restrict__ out)
{
    __shared__ float sa[32][65], sb[32][65];
    const int q0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < D; k0 += 32) {
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int r = i >> 5, k = i & 31;
            sa[k][r] = (q0 + r < Q && k0 + k < D) ? q[(size_t)(q0 + r) * D + k0 + k] : 0.f;
            sb[k][r] = (n0 + r < Nb && k0 + k < D) ? bank[(size_t)(n0 + r) * D + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sa[k][ty * 4 + i]; b[i] = sb[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = a[i] - b[j]; acc[i][j] += d * d; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = q0 + ty * 4 + i, nn = n0 + tx * 4 + j;
            if (qq < Q && nn < Nb) out[(size_t)qq * Nb + nn] = sqrtf(acc[i][j]);
        }
}

int scan_blocks(int Nb)
{
    const int groups = (Nb + 15) / 16;
    int blocks = (groups + 7) / 8;   // at least one group per wave
    return blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);   // one block per CU
}

}  // namespace

extern "C" size_t cmdiad_bank_block16_floats(int Nb, int D) { return (size_t)((Nb + 15) / 16) * 16 * (size_t)D; }

extern "C" int cmdiad_bank_block16(const float* bank, int Nb, int D, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(bank && out, CMDIAD_ERR_ARG, "cmdiad_bank_block16: null pointer");
    CMDIAD_REQUIRE(D % 16 == 0 && aligned16h(bank) && aligned16h(out), CMDIAD_ERR_ARG, "cmdiad_bank_block16: D%%16, alignment");
    if (Nb == 0) return CMDIAD_OK;
    const size_t n4 = cmdiad_bank_block16_floats(Nb, D) / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(bank_block16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bank, Nb, D, out, n4);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" size_t cmdiad_reweight_workspace_bytes(int R, int Nb)
{
    (void)R;
    return (size_t)kProbes * scan_blocks(Nb) * kCand * sizeof(unsigned long long);
}

static int scan_launch(ScanPair& pair, int D, hipStream_t s, const char* who)
{
    const size_t lds = (size_t)2 * (D / 16) * 64 * 16 + kProbes * 4 + (size_t)kWaves * kProbes * kCand * 8;
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute((const void*)reweight_scan_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess) {
            cmdiad_set_error("%s: %zu bytes of LDS exceed the device limit", who, lds);
            return CMDIAD_ERR_ARG;
        }
        attr = lds;
    }
    hipLaunchKernelGGL(reweight_scan_mfma_kernel, dim3(pair.p[0].blocks + pair.p[1].blocks), dim3(kWaves * 64), lds, s, pair, D);
    CMDIAD_CHECK_LAUNCH();
    hipLaunchKernelGGL(reweight_exact_merge_kernel, dim3(pair.p[0].R + pair.p[1].R), dim3(kCand * 64), 0, s, pair, D);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}

extern "C" int cmdiad_reweight_scan(const float* probes, const float* bank, const float* bank_block16, int R, int Nb, int D,
                                    uint32_t row_offset, unsigned long long* top3, void* workspace, size_t workspace_bytes,
                                    cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(probes && bank && bank_block16 && top3, CMDIAD_ERR_ARG, "cmdiad_reweight_scan: null pointer");
    CMDIAD_REQUIRE(R > 0 && R <= kProbes && D % (16 * kChunk) == 0 && D <= 1024 && aligned16h(probes) && aligned16h(bank) &&
                       aligned16h(bank_block16),
                   CMDIAD_ERR_ARG, "cmdiad_reweight_scan: 0<R<=32, D%%128==0, D<=1024, 16-byte alignment (R=%d D=%d)", R, D);
    if (Nb == 0) return CMDIAD_OK;
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_reweight_workspace_bytes(R, Nb), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_reweight_scan: workspace too small");
    ScanPair pair{};
    pair.p[0] = ScanProblem{probes, bank_block16, bank, R, Nb, row_offset, (unsigned long long*)workspace, top3, scan_blocks(Nb)};
    return scan_launch(pair, D, (hipStream_t)stream, "cmdiad_reweight_scan");
}

// The two libraries of a scored batch in ONE launch pair (features.py:235-254 runs per library and sample): the scan kernel's
// workgroups are divided between the libraries in proportion to their rows (one workgroup per CU in all), the merge kernel takes
// the probes of both.  A call of cmdiad_reweight_scan costs ~45 us of prologue (probes into LDS), candidate merges and the second
// launch on top of its stream time -- 42 us for the 59 MB rgb library of the bench (0.17 of the HBM rate) behind 68 us for the
// 235 MB xyz library; as one pair the small library's fixed cost runs beside the large library's stream.  Results are those of two
// separate calls -- every row is seen once and the eight approximate candidates per probe are re-evaluated exactly either way --
// with ONE caveat: a lane keeps kLane = 4 approximate candidates per (probe, row residue) slot, and WHICH rows share a slot depends
// on the number of workgroups a library gets (256 alone, its share of 256 in a pair).  If more than four of a probe's eight best
// approximate rows fall into one slot (clusters of near-duplicate rows at a stride of 2 x slots), the candidate lists -- and in the
// extreme the exact top-3 -- can differ between the pair and the single launch.  Both are valid answers of the same approximation;
// the parity tests (bench-sized libraries, random and duplicated rows: tests/test_gpu_kernels.py, tools/fuzz_gpu.py) have not met a case.
extern "C" size_t cmdiad_reweight_pair_workspace_bytes(int Nb0, int Nb1)
{
    return cmdiad_reweight_workspace_bytes(kProbes, Nb0) + cmdiad_reweight_workspace_bytes(kProbes, Nb1);   // (upper bound: <= 256 workgroups each)
}

extern "C" int cmdiad_reweight_scan_pair(const float* probes0, const float* bank0, const float* bank0_block16, int R0, int Nb0,
                                         uint32_t row_offset0, unsigned long long* top3_0, const float* probes1, const float* bank1,
                                         const float* bank1_block16, int R1, int Nb1, uint32_t row_offset1, unsigned long long* top3_1,
                                         int D, void* workspace, size_t workspace_bytes, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(probes0 && bank0 && bank0_block16 && top3_0 && probes1 && bank1 && bank1_block16 && top3_1, CMDIAD_ERR_ARG,
                   "cmdiad_reweight_scan_pair: null pointer");
    CMDIAD_REQUIRE(R0 > 0 && R0 <= kProbes && R1 > 0 && R1 <= kProbes && Nb0 > 0 && Nb1 > 0 && D % (16 * kChunk) == 0 && D <= 1024 &&
                       aligned16h(probes0) && aligned16h(bank0) && aligned16h(bank0_block16) && aligned16h(probes1) && aligned16h(bank1) &&
                       aligned16h(bank1_block16),
                   CMDIAD_ERR_ARG, "cmdiad_reweight_scan_pair: 0<R<=32, Nb>0, D%%128==0, D<=1024, 16-byte alignment (R=%d,%d D=%d)", R0, R1, D);
    CMDIAD_REQUIRE(workspace && workspace_bytes >= cmdiad_reweight_pair_workspace_bytes(Nb0, Nb1), CMDIAD_ERR_WORKSPACE,
                   "cmdiad_reweight_scan_pair: workspace too small");
    // one workgroup per CU in all, shared in proportion to the rows; never more than a library alone would get, never fewer than one
    const int cap0 = scan_blocks(Nb0), cap1 = scan_blocks(Nb1);
    int b0 = (int)((256.0 * Nb0) / ((double)Nb0 + Nb1) + 0.5);
    b0 = b0 < 1 ? 1 : (b0 > 255 ? 255 : b0);
    int b1 = 256 - b0;
    b0 = b0 > cap0 ? cap0 : b0;
    b1 = b1 > cap1 ? cap1 : b1;
    ScanPair pair{};
    unsigned long long* ws = (unsigned long long*)workspace;
    pair.p[0] = ScanProblem{probes0, bank0_block16, bank0, R0, Nb0, row_offset0, ws, top3_0, b0};
    pair.p[1] = ScanProblem{probes1, bank1_block16, bank1, R1, Nb1, row_offset1, ws + (size_t)kProbes * cap0 * kCand, top3_1, b1};
    return scan_launch(pair, D, (hipStream_t)stream, "cmdiad_reweight_scan_pair");
}

extern "C" int cmdiad_l2_dist_matrix(const float* q, const float* bank, int Q, int Nb, int D, float* out,
                                     cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && bank && out, CMDIAD_ERR_ARG, "cmdiad_l2_dist_matrix: null pointer");
    if (Q == 0 || Nb == 0) return CMDIAD_OK;
    hipLaunchKernelGGL(l2_dist_matrix_kernel, dim3((Nb + 63) / 64, (Q + 63) / 64), dim3(256), 0, (hipStream_t)stream, q, bank, Q,
                       Nb, D, out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
