/*
 * cmdiad_oracle.c -- CPU restatement of the index/selection/pooling arithmetic on
 * CMDIAD's hot path.  TEST INFRASTRUCTURE ONLY: nothing under cmdiad_amd/ may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * Compile with -O2 -ffp-contract=off (see oracle/Makefile): every fp32 expression
 * below is evaluated in exactly the written order with one rounding per operation,
 * which is what the HIP kernels in cmdiad_amd/csrc reproduce bit-for-bit for the
 * index-producing stages (FPS, kNN-group, 3-NN interpolation indices).
 *
 * Citations are file:line into the reference tree (evenrose/CMDIAD).
 *
 * Pinning status
 *   orc_fps, orc_knn_group : the arithmetic lives in third-party CUDA packages that are
 *       NOT vendored in the reference (pointnet2_ops master, KNN_CUDA 0.2; README.md:22-24).
 *       This file restates their published algorithm; "parity unpinned" for these two
 *       (no reference output can be produced in this container).  Call sites that anchor
 *       the contract: models/models.py:76-77, 86, 100-112.
 *   orc_interp3nn, orc_xyz_patch, orc_l2_min_argmin, orc_bilinear_up : pinned against
 *       outputs of the reference's own Python functions (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Farthest point sampling.  Reference call site: models/models.py:70-78 (fps) ->
 * pointnet2_ops.furthest_point_sample + gather_operation [external, unpinned master].
 * Published algorithm restated: first index 0; running min distance initialised to
 * 1e10; points with |p|^2 <= 1e-3 are skipped (never updated, never selected);
 * each round selects argmax of the running min.  Tie rule fixed here: lowest index.
 * xyz [B,N,3] f32, idx [B,G] int32, centers [B,G,3] f32 (gather fused).
 * ---------------------------------------------------------------------------------- */
void orc_fps(const float *xyz, int B, int N, int G, int32_t *idx, float *centers)
{
    float *temp = (float *)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        int32_t *out = idx + (size_t)b * G;
        for (int k = 0; k < N; ++k) temp[k] = 1e10f;
        int old = 0;
        if (G > 0) out[0] = 0;
        for (int j = 1; j < G; ++j) {
            int besti = 0;
            float best = -1.0f;
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int k = 0; k < N; ++k) {
                const float x2 = p[k * 3 + 0], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
                const float mag = (x2 * x2 + y2 * y2) + z2 * z2;
                if (mag <= 1e-3f) continue;
                const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const float d2 = d < temp[k] ? d : temp[k];
                temp[k] = d2;
                if (d2 > best) { best = d2; besti = k; }
            }
            old = besti;
            out[j] = old;
        }
        if (centers) {
            float *c = centers + (size_t)b * G * 3;
            for (int j = 0; j < G; ++j) {
                c[j * 3 + 0] = p[out[j] * 3 + 0];
                c[j * 3 + 1] = p[out[j] * 3 + 1];
                c[j * 3 + 2] = p[out[j] * 3 + 2];
            }
        }
    }
    free(temp);
}

/* ------------------------------------------------------------------------------------
 * kNN grouping.  Reference: models/models.py:88-113 (Group.forward) ->
 * knn_cuda.KNN(k, transpose_mode=True) [external, wheel 0.2]: brute-force squared L2,
 * k smallest in ascending order, int64 indices; then gather + centre subtraction
 * (models.py:105-112).  Order fixed here: ascending (d2, index).
 * xyz [B,N,3], center [B,G,3] -> idx [B,G,K] int64, neigh [B,G,K,3] f32.
 * ---------------------------------------------------------------------------------- */
typedef struct { float d; int32_t i; } orc_pair;

static int orc_pair_less(float da, int32_t ia, float db, int32_t ib)
{
    return (da < db) || (da == db && ia < ib);
}

void orc_knn_group(const float *xyz, const float *center, int B, int N, int G, int K,
                   int64_t *idx, float *neigh)
{
    orc_pair *heap = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)K);
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        for (int g = 0; g < G; ++g) {
            const float *c = center + ((size_t)b * G + g) * 3;
            int cnt = 0; /* sorted insertion list of the K best so far */
            for (int k = 0; k < N; ++k) {
                const float dx = p[k * 3 + 0] - c[0];
                const float dy = p[k * 3 + 1] - c[1];
                const float dz = p[k * 3 + 2] - c[2];
                const float d = (dx * dx + dy * dy) + dz * dz;
                if (cnt == K && !orc_pair_less(d, k, heap[K - 1].d, heap[K - 1].i)) continue;
                int pos = cnt < K ? cnt : K - 1;
                while (pos > 0 && orc_pair_less(d, k, heap[pos - 1].d, heap[pos - 1].i)) {
                    heap[pos] = heap[pos - 1];
                    --pos;
                }
                heap[pos].d = d; heap[pos].i = k;
                if (cnt < K) ++cnt;
            }
            int64_t *o = idx + ((size_t)b * G + g) * K;
            float *nb = neigh ? neigh + ((size_t)b * G + g) * K * 3 : 0;
            for (int k = 0; k < K; ++k) {
                const int32_t i = k < cnt ? heap[k].i : 0;
                o[k] = i;
                if (nb) {
                    nb[k * 3 + 0] = p[i * 3 + 0] - c[0];
                    nb[k * 3 + 1] = p[i * 3 + 1] - c[1];
                    nb[k * 3 + 2] = p[i * 3 + 2] - c[2];
                }
            }
        }
    }
    free(heap);
}

/* ------------------------------------------------------------------------------------
 * 3-NN inverse-distance feature interpolation.  Reference:
 * models/pointnet2_utils.py:45-75 (interpolating_points), square_distance :4-23
 * (dist = -2*a.b ; += |a|^2 ; += |b|^2), index_points :26-42.  The reference sorts all
 * S distances (:66) and keeps three; only the three smallest matter.
 * xyz1 [N,3] points, xyz2 [S,3] centres, feat [S,D] (points2 permuted to S-major)
 * -> out [N,D]; optional idx3 [N,3] int32 and w3 [N,3] f32.
 * ---------------------------------------------------------------------------------- */
void orc_interp3nn(const float *xyz1, const float *xyz2, const float *feat, int N, int S, int D,
                   float *out, int32_t *idx3, float *w3)
{
    float *n2 = (float *)malloc(sizeof(float) * (size_t)S);
    for (int s = 0; s < S; ++s) {
        const float x = xyz2[s * 3], y = xyz2[s * 3 + 1], z = xyz2[s * 3 + 2];
        n2[s] = (x * x + y * y) + z * z;
    }
    for (int n = 0; n < N; ++n) {
        const float x = xyz1[n * 3], y = xyz1[n * 3 + 1], z = xyz1[n * 3 + 2];
        const float n1 = (x * x + y * y) + z * z;
        float bd[3] = {INFINITY, INFINITY, INFINITY};
        int32_t bi[3] = {0, 0, 0};
        for (int s = 0; s < S; ++s) {
            const float dot = (x * xyz2[s * 3] + y * xyz2[s * 3 + 1]) + z * xyz2[s * 3 + 2];
            float d = -2.0f * dot;
            d = d + n1;
            d = d + n2[s];
            if (orc_pair_less(d, s, bd[2], bi[2])) {
                int pos = 2;
                while (pos > 0 && orc_pair_less(d, s, bd[pos - 1], bi[pos - 1])) {
                    bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos;
                }
                bd[pos] = d; bi[pos] = s;
            }
        }
        float r[3];
        const int kk = S < 3 ? S : 3;
        for (int k = 0; k < 3; ++k) r[k] = k < kk ? 1.0f / (bd[k] + 1e-8f) : 0.0f;
        const float norm = (r[0] + r[1]) + r[2];
        float w[3];
        for (int k = 0; k < 3; ++k) w[k] = r[k] / norm;
        if (idx3) { idx3[n * 3] = bi[0]; idx3[n * 3 + 1] = bi[1]; idx3[n * 3 + 2] = bi[2]; }
        if (w3) { w3[n * 3] = w[0]; w3[n * 3 + 1] = w[1]; w3[n * 3 + 2] = w[2]; }
        if (out) {
            const float *f0 = feat + (size_t)bi[0] * D, *f1 = feat + (size_t)bi[1] * D,
                        *f2 = feat + (size_t)bi[2] * D;
            float *o = out + (size_t)n * D;
            for (int c = 0; c < D; ++c) o[c] = (f0[c] * w[0] + f1[c] * w[1]) + f2[c] * w[2];
        }
    }
    free(n2);
}

/* ------------------------------------------------------------------------------------
 * xyz patch pooling.  Reference: feature_extractors/features.py:169-184
 * (get_xyz_patch): scatter interpolated features into a zero [D, S*S] map at
 * nonzero_indices (:171-173), AvgPool2d(3, stride 1) (:72,176) -> (S-2)^2,
 * AdaptiveAvgPool2d((P,P)) (:73-74) with bins [floor(i*L/P), ceil((i+1)*L/P)),
 * reshape to [P*P, D] (:177).   interp [N,D] row-major (point-major), nz [N] int64.
 * ---------------------------------------------------------------------------------- */
void orc_xyz_patch(const float *interp, const int64_t *nz, int N, int D, int S, int P, float *out)
{
    const int L = S - 2;
    float *full = (float *)calloc((size_t)S * S * D, sizeof(float));
    float *avg = (float *)malloc(sizeof(float) * (size_t)L * L * D);
    for (int n = 0; n < N; ++n)
        memcpy(full + (size_t)nz[n] * D, interp + (size_t)n * D, sizeof(float) * (size_t)D);
    for (int y = 0; y < L; ++y)
        for (int x = 0; x < L; ++x) {
            float *a = avg + ((size_t)y * L + x) * D;
            for (int c = 0; c < D; ++c) {
                float s = 0.0f;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx)
                        s += full[((size_t)(y + dy) * S + (x + dx)) * D + c];
                a[c] = s / 9.0f;
            }
        }
    for (int py = 0; py < P; ++py) {
        const int y0 = (py * L) / P, y1 = ((py + 1) * L + P - 1) / P;
        for (int px = 0; px < P; ++px) {
            const int x0 = (px * L) / P, x1 = ((px + 1) * L + P - 1) / P;
            float *o = out + ((size_t)py * P + px) * D;
            const float cnt = (float)((y1 - y0) * (x1 - x0));
            for (int c = 0; c < D; ++c) {
                float s = 0.0f;
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) s += avg[((size_t)y * L + x) * D + c];
                o[c] = s / cnt;
            }
        }
    }
    free(full);
    free(avg);
}

/* ------------------------------------------------------------------------------------
 * Patch-library nearest neighbour.  Reference: features.py:186-190 (calculate_dist,
 * torch.cdist p=2) followed by torch.min(dist, dim=1) (:227).  Restated as the exact
 * Euclidean distance accumulated in double (the reference's cdist uses a matmul
 * expansion whose own fp32 error is ~1e-3 absolute near zero; goldens are compared
 * with that tolerance).  First occurrence wins ties, as torch.min does.
 * ---------------------------------------------------------------------------------- */
void orc_l2_min_argmin(const float *q, const float *bank, int Q, int Nb, int D,
                       float *min_val, int64_t *min_idx)
{
    for (int i = 0; i < Q; ++i) {
        const float *a = q + (size_t)i * D;
        double best = INFINITY;
        int64_t bi = 0;
        for (int j = 0; j < Nb; ++j) {
            const float *b = bank + (size_t)j * D;
            double s = 0.0;
            for (int c = 0; c < D; ++c) {
                const double t = (double)a[c] - (double)b[c];
                s += t * t;
            }
            if (s < best) { best = s; bi = j; }
        }
        min_val[i] = (float)sqrt(best);
        min_idx[i] = bi;
    }
}

/* ------------------------------------------------------------------------------------
 * Bilinear up-sampling of the per-patch score map.  Reference: features.py:293-294
 * (torch.nn.functional.interpolate(mode='bilinear'), align_corners=False default).
 * in [h,h] -> out [H,H].  src = (dst+0.5)*h/H-0.5 clamped at 0 (ATen
 * area_pixel_compute_source_index).
 * ---------------------------------------------------------------------------------- */
void orc_bilinear_up(const float *in, int h, int H, float *out)
{
    const float scale = (float)h / (float)H;
    for (int oy = 0; oy < H; ++oy) {
        float sy = scale * ((float)oy + 0.5f) - 0.5f;
        if (sy < 0.0f) sy = 0.0f;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly = sy - (float)y0, hy = 1.0f - ly;
        for (int ox = 0; ox < H; ++ox) {
            float sx = scale * ((float)ox + 0.5f) - 0.5f;
            if (sx < 0.0f) sx = 0.0f;
            const int x0 = (int)sx;
            const int x1 = x0 + (x0 < h - 1 ? 1 : 0);
            const float lx = sx - (float)x0, hx = 1.0f - lx;
            out[oy * H + ox] = hy * (hx * in[y0 * h + x0] + lx * in[y0 * h + x1]) +
                               ly * (hx * in[y1 * h + x0] + lx * in[y1 * h + x1]);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * 8-bit Gaussian blur as Pillow computes it -- the arithmetic behind the reference's
 * KNNGaussianBlur (utils/utils.py:71-83: ToPILImage -> ImageFilter.GaussianBlur(radius=4)
 * -> ToTensor).  Pillow is a third-party dependency of the reference (README.md pins none;
 * this image has Pillow 12.2); its published algorithm (src/libImaging/BoxBlur.c,
 * unchanged in substance since Pillow 2.7) is restated here and PINNED by comparing with
 * the installed Pillow on random images (tests/test_oracle_golden.py::test_pil_blur_restatement).
 *
 *   box radius r = l + a from sigma^2 = radius^2 / passes (Gwosdek et al., "Theoretical
 *   foundations of Gaussian convolution by extended box filtering"), passes = 3;
 *   per line: running sum over 2*floor(r)+1 pixels with replicated edges, plus the two
 *   pixels just outside the window weighted by the fractional part; fixed point 2^24,
 *   round half up;  3 horizontal passes, transpose, 3 horizontal passes, transpose.
 * in/out [h,w] uint8 (may alias).  Returns 0, or -1 if a side is shorter than 2*floor(r)+2
 * (Pillow's short-line branch is not restated).
 * ---------------------------------------------------------------------------------- */
static float orc_gaussian_box_radius(float radius, int passes)
{
    float sigma2, L, l, a;
    sigma2 = radius * radius / passes;
    L = sqrt(12.0 * sigma2 + 1.0);
    l = floor((L - 1.0) / 2.0);
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    return l + a;
}

static void orc_line_box_blur8(uint8_t *out, const uint8_t *in, int lastx, int radius, int edgeA, int edgeB,
                               uint32_t ww, uint32_t fw)
{
    int x;
    uint32_t acc, bulk;
    acc = in[0] * (radius + 1);
    for (x = 0; x < edgeA - 1; x++) acc += in[x];
    acc += in[lastx] * (radius - edgeA + 1);
    for (x = 0; x < edgeA; x++) {
        acc += in[x + radius] - in[0];
        bulk = acc * ww + (in[0] + in[x + radius + 1]) * fw;
        out[x] = (uint8_t)((bulk + (1 << 23)) >> 24);
    }
    for (x = edgeA; x < edgeB; x++) {
        acc += in[x + radius] - in[x - radius - 1];
        bulk = acc * ww + (in[x - radius - 1] + in[x + radius + 1]) * fw;
        out[x] = (uint8_t)((bulk + (1 << 23)) >> 24);
    }
    for (x = edgeB; x <= lastx; x++) {
        acc += in[lastx] - in[x - radius - 1];
        bulk = acc * ww + (in[x - radius - 1] + in[lastx]) * fw;
        out[x] = (uint8_t)((bulk + (1 << 23)) >> 24);
    }
}

int orc_pil_gaussian_blur_u8(const uint8_t *in, int h, int w, float radius, uint8_t *out)
{
    const int passes = 3;
    const float fr = orc_gaussian_box_radius(radius, passes);
    const int r = (int)fr;
    if (w < 2 * r + 2 || h < 2 * r + 2) return -1;
    const uint32_t ww = (uint32_t)((uint32_t)(1 << 24) / (fr * 2 + 1));
    const uint32_t fw = ((1 << 24) - (r * 2 + 1) * ww) / 2;
    uint8_t *a = (uint8_t *)malloc((size_t)h * w), *t = (uint8_t *)malloc((size_t)h * w);
    uint8_t *line = (uint8_t *)malloc((size_t)(h > w ? h : w));
    memcpy(a, in, (size_t)h * w);
    for (int dim = 0; dim < 2; ++dim) {
        const int rows = dim == 0 ? h : w, cols = dim == 0 ? w : h;
        uint8_t *cur = dim == 0 ? a : t;
        if (dim == 1)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) t[(size_t)x * h + y] = a[(size_t)y * w + x];
        const int edgeA = r + 1 < cols ? r + 1 : cols, edgeB = cols - r - 1 > 0 ? cols - r - 1 : 0;
        for (int p = 0; p < passes; ++p)
            for (int y = 0; y < rows; ++y) {
                orc_line_box_blur8(line, cur + (size_t)y * cols, cols - 1, r, edgeA, edgeB, ww, fw);
                memcpy(cur + (size_t)y * cols, line, (size_t)cols);
            }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = t[(size_t)x * h + y];
    free(a); free(t); free(line);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Ball query.  pointnet2_ops.pointnet2_utils.ball_query [external, un-vendored; named by the task's
 * north_star, not called by the reference -- SURVEY F7].  Published algorithm restated: for every query point walk the
 * cloud in index order; a point is inside when dx*dx + dy*dy + dz*dz < radius^2; the first hit fills all nsample slots,
 * later hits overwrite slots 1, 2, ... until nsample points are found; no hit leaves zeros.  "Parity unpinned".
 * xyz [B,N,3], new_xyz [B,M,3] -> idx [B,M,nsample] int32.
 * ---------------------------------------------------------------------------------- */
void orc_ball_query(const float *xyz, const float *new_xyz, int B, int N, int M, float radius, int nsample, int32_t *idx)
{
    const float r2 = radius * radius;
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < M; ++j) {
            const float *q = new_xyz + ((size_t)b * M + j) * 3;
            int32_t *out = idx + ((size_t)b * M + j) * nsample;
            for (int l = 0; l < nsample; ++l) out[l] = 0;
            int cnt = 0;
            for (int k = 0; k < N && cnt < nsample; ++k) {
                const float *p = xyz + ((size_t)b * N + k) * 3;
                const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < r2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) out[l] = k;
                    out[cnt] = k;
                    ++cnt;
                }
            }
        }
}
