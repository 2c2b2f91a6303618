// Fused softmax(q k^T) v for gfx950 (reference models/models.py:148-160 Attention.forward, and the
// same algebra inside timm's ViT blocks reached at models/models.py:48), head_dim = 64, bf16 MFMA with
// fp32 softmax statistics and accumulation.  The T x T score matrix never leaves the CU.
//
// Work decomposition: one block per (128-query tile, head, batch) on a 1-D XCD-aware grid; 256 threads = 4 waves; a wave owns 32 queries and
// walks the keys in tiles of 64.  Everything that belongs to ONE query lives on ONE lane:
//   S^T tile (32 keys x 32 queries) = mfma_32x32x16(A = K rows from LDS, B = Q rows in registers)
//       -> accumulator column = query (lane & 31), rows = keys spread over the 16 registers and the
//          two lane halves, so the row max / row sum are in-lane reductions + one cross-half shuffle.
//   O^T tile (32 d x 32 queries)  = mfma_32x32x16(A = V^T rows from LDS, B = P)
//       -> P is taken STRAIGHT from the S^T accumulator (guide §3 "accumulator tile as the next
//          MFMA's operand"): registers 8s..8s+7 converted to bf16 are the B fragment of k-step s, with
//          the fixed k permutation k = 16s + 8(j>>2) + 4h + (j&3) matched on the V^T side by two
//          8-byte LDS reads.  The online-softmax rescale of O^T is a per-lane scalar multiply.
// Q is pre-scaled by head_dim^-0.5 * log2(e) and V arrives transposed ([B,H,64,Tp]) from cmdiad_gemm_qkv.
// LDS tiles are padded (K rows 144 B, V^T rows 136 B) so the fragment reads are bank-conflict-free.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kKeys = 64;          // keys per tile
constexpr int kKStride = 144;      // bytes per K row in LDS  (64 bf16 + 16 pad)
constexpr int kVStride = 136;      // bytes per V^T row in LDS (64 bf16 + 8 pad)
#ifndef CMDIAD_ATT_LAZY
#define CMDIAD_ATT_LAZY 8.0f
#endif
constexpr float kLazy = CMDIAD_ATT_LAZY;   // log2 units a score may exceed the softmax reference before the accumulators are rescaled

// q is pre-multiplied by head_dim^-0.5 * log2(e) (cmdiad_gemm_qkv), so softmax is exp2 of the raw dot product.
template <int OCC>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void attention_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                             const bf16_t* __restrict__ vt, int BH, int H, int T, int Tp,
                                                             bf16_t* __restrict__ out)
{
    // two LDS stages: the next tile is written while the current one is being read -> ONE barrier per tile
    __shared__ __attribute__((aligned(16))) char s_k[2][kKeys * kKStride];
    __shared__ __attribute__((aligned(16))) char s_v[2][64 * kVStride];

    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs, so linear id L runs on XCD L % 8.
    // All query tiles of one (batch, head) are given ids with the same L % 8: its K / V^T (2 x Tp x 128 B) are then
    // pulled into ONE XCD's L2 instead of up to 8 (measured before: 5x the algorithmic fetch, L2 hit 0.60).
    const int nq = (T + 127) / 128;
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int head = (slot / nq) * 8 + xcd, qtile = slot - (slot / nq) * nq;
    if (head >= BH) return;
    const int b = head / H, h = head - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const size_t bh = (size_t)b * H + h;
    const bf16_t* qb = q + bh * Tp * 64;
    const bf16_t* kb = k + bh * Tp * 64;
    const bf16_t* vb = vt + bh * 64 * Tp;
    const int q0 = qtile * 128 + wave * 32;
    const int C = H * 64;

    // Q fragments (B operand): lane (query r, half hh) holds Q[q][16s + 8hh .. +7] for s = 0..3
    bf16x8 qf[4];
    {
        const int qi = min(q0 + r, Tp - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qb + (size_t)qi * 64 + 16 * s + 8 * hh);
    }

    f32x16 o0 = {}, o1 = {};
    float m_run = -__builtin_inff(), l_run = 0.0f;

    const int nkt = (T + kKeys - 1) / kKeys;
    // Tile loads go through buffer instructions: the (batch, head) slice is a wave-uniform resource in SGPRs, the lane's place in
    // a tile a 32-bit offset computed once, the tile index a scalar offset -- with flat pointers the loop carried four 64-bit
    // per-lane addresses (25 v_lshl_add_u64 per tile) and, at four waves per SIMD, spilled the prefetch registers right after the
    // loads were issued, i.e. waited for them (profiles/r2_notes.md).
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kb), 0, Tp * 64 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vb), 0, 64 * Tp * 2, 0x00020000);
    // thread t brings 16-byte chunk t & 7 of rows (t >> 3) and (t >> 3) + 32 of both tiles: the second row is a scalar offset
    const int koff = ((tid >> 3) * 64 + (tid & 7) * 8) * 2, voff = ((tid >> 3) * Tp + (tid & 7) * 8) * 2;
    uint4 rk[2], rv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(krs, koff, kt * (kKeys * 64 * 2) + i * (32 * 64 * 2), 0);
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, kt * (kKeys * 2) + i * (32 * Tp * 2), 0);
            rk[i] = make_uint4(a[0], a[1], a[2], a[3]);
            rv[i] = make_uint4(c[0], c[1], c[2], c[3]);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 3, ch = idx & 7;
            *reinterpret_cast<uint4*>(s_k[buf] + row * kKStride + ch * 16) = rk[i];
            uint2* dv = reinterpret_cast<uint2*>(s_v[buf] + row * kVStride + ch * 16);  // 8-byte aligned rows
            dv[0] = make_uint2(rv[i].x, rv[i].y);
            dv[1] = make_uint2(rv[i].z, rv[i].w);
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const char* sk = s_k[kt & 1];
        const char* sv = s_v[kt & 1];
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);  // global loads stay in flight under the MFMAs below

        // ---- S^T = K . Q^T for the two 32-key sub-tiles.  The accumulators start at -m_run (column = query = lane & 31: a per-lane
        //      constant), so the tile arrives as S - m_run and, while no query's maximum grows, goes into exp2 without a subtraction
        //      per element (first tile: m_run = -inf, start at 0)
        const float c0 = kt == 0 ? 0.0f : -m_run;
        f32x16 s0, s1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s0[e] = c0; s1[e] = c0; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ka = *reinterpret_cast<const bf16x8*>(sk + r * kKStride + 32 * s + 16 * hh);
            const bf16x8 kb2 = *reinterpret_cast<const bf16x8*>(sk + (32 + r) * kKStride + 32 * s + 16 * hh);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb2, qf[s], s1, 0, 0, 0);
        }
        // ---- keys >= T only exist in the last tile
        if (kt == nkt - 1 && (T & (kKeys - 1))) {
            const int kbase = kt * kKeys + 4 * hh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kbase + (e & 3) + 8 * (e >> 2);
                s0[e] = key < T ? s0[e] : -__builtin_inff();
                s1[e] = key + 32 < T ? s1[e] : -__builtin_inff();
            }
        }
        // ---- online softmax, per lane = per query.  mloc = tile maximum relative to the running one (absolute in the first
        //      tile); the accumulators are rescaled only when some query's maximum actually grew (wave-uniform branch; exact)
        float mloc;
        {
            float t[16];   // v_max3 tree: 32 values in 16 instructions (fmaxf() of MFMA results also emits a quieting v_max each)
#pragma unroll
            for (int e = 0; e < 16; ++e) t[e] = s0[e];
#pragma unroll
            for (int e = 0; e < 8; ++e) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t[e]) : "v"(t[e]), "v"(s1[2 * e]), "v"(s1[2 * e + 1]));
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t[e]) : "v"(t[e]), "v"(t[8 + 2 * e]), "v"(t[9 + 2 * e]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t[0]) : "v"(t[0]), "v"(t[4]), "v"(t[5]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t[1]) : "v"(t[1]), "v"(t[6]), "v"(t[7]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mloc) : "v"(t[0]), "v"(t[1]), "v"(t[2]));
            asm("v_max_f32 %0, %1, %2" : "=v"(mloc) : "v"(mloc), "v"(t[3]));
        }
        mloc = half_max(mloc);   // the other 32 keys of this query sit on lane ^ 32
        // m_run is a REFERENCE, not the exact running maximum: it moves only when some query of the wave found a score more than
        // kLazy above it (P <= 2^kLazy: nothing near the fp32 / bf16 range, and the normalisation O / l cancels the reference).
        // With the exact maximum the branch was taken on nearly every tile -- among a wave's 32 queries one maximum almost always
        // grows -- and its ~100 vector instructions were a third of the loop's issue slots (profiles/r5_notes.md section 10).
        const float thr = kt == 0 ? -__builtin_inff() : kLazy;   // first tile: always take the full path
        if (!__all(mloc <= thr)) {
            // delta = growth of this query's maximum (0 where it did not grow); the tile becomes S - m_new
            const float m_new = kt == 0 ? mloc : m_run + fmaxf(mloc, 0.0f);
            const float delta = kt == 0 ? mloc : fmaxf(mloc, 0.0f);
            const float alpha = kt == 0 ? 0.0f : __builtin_amdgcn_exp2f(-delta);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; s0[e] -= delta; s1[e] -= delta; }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s0[e] = __builtin_amdgcn_exp2f(s0[e]);
            s1[e] = __builtin_amdgcn_exp2f(s1[e]);
        }
        f32x2 ps = {0.0f, 0.0f};   // two partial sums: v_pk_add_f32 takes a register pair per issue slot
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            ps += f32x2{s0[e], s0[e + 1]};
            ps += f32x2{s1[e], s1[e + 1]};
        }
        l_run += ps[0] + ps[1];

        // ---- O^T += V^T . P   (k-step = 16 keys; P fragment = 8 consecutive accumulator registers)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = f2bf(sub == 0 ? s0[8 * s + j] : s1[8 * s + j]);
                const int koff = (sub * 32 + 16 * s + 4 * hh) * 2;  // bytes into the key axis
                union { uint2 u[2]; bf16x8 v; } a0, a1;
                a0.u[0] = *reinterpret_cast<const uint2*>(sv + r * kVStride + koff);
                a0.u[1] = *reinterpret_cast<const uint2*>(sv + r * kVStride + koff + 16);
                a1.u[0] = *reinterpret_cast<const uint2*>(sv + (32 + r) * kVStride + koff);
                a1.u[1] = *reinterpret_cast<const uint2*>(sv + (32 + r) * kVStride + koff + 16);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.v, pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, pf, o1, 0, 0, 0);
            }
        }
        if (more) store_tile((kt + 1) & 1);  // that stage was last read in iteration kt-1
        __syncthreads();
    }

    const float l_tot = half_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + r;
    if (qi < T) {
        bf16_t* dst = out + ((size_t)b * T + qi) * C + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 a = {f2bf(o0[4 * g] * inv), f2bf(o0[4 * g + 1] * inv), f2bf(o0[4 * g + 2] * inv), f2bf(o0[4 * g + 3] * inv)};
            bf16x4 c = {f2bf(o1[4 * g] * inv), f2bf(o1[4 * g + 1] * inv), f2bf(o1[4 * g + 2] * inv), f2bf(o1[4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(dst + 8 * g + 4 * hh) = a;
            *reinterpret_cast<bf16x4*>(dst + 32 + 8 * g + 4 * hh) = c;
        }
    }
}

}  // namespace

extern "C" int cmdiad_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, int B, int H, int T,
                                uint16_t* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(q && k && vt && out, CMDIAD_ERR_ARG, "cmdiad_attention: null pointer");
    CMDIAD_REQUIRE(B > 0 && H > 0 && T > 0, CMDIAD_ERR_ARG, "cmdiad_attention: bad sizes");
    CMDIAD_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt) & 15) == 0 && ((uintptr_t)out & 7) == 0, CMDIAD_ERR_ARG,
                   "cmdiad_attention: alignment");
    const int Tp = (T + 63) / 64 * 64;
    const int BH = B * H, nq = (T + 127) / 128;
    dim3 grid((unsigned)((BH + 7) / 8 * 8 * nq));
    static const int occ = getenv("CMDIAD_ATT_OCC") ? atoi(getenv("CMDIAD_ATT_OCC")) : 4;  // 128 VGPRs, 4 waves/SIMD: +5 % on the Point-MAE shape, neutral on ViT
    auto kern = occ == 4 ? attention_kernel<4> : occ == 3 ? attention_kernel<3> : attention_kernel<2>;
    hipLaunchKernelGGL(kern, grid, dim3(kThreads), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)vt, BH, H, T, Tp, (bf16_t*)out);
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
