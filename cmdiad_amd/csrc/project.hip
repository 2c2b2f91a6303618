// Sparse random projection of the patch library in front of the greedy coreset selection (reference features.py:360-371:
// sklearn.random_projection.SparseRandomProjection(eps).fit_transform(z_lib) on the host -- 2.3 GB over PCIe and ~6 s of one core per
// bagel-sized library).  The transform is  out = X . components^T  with `components` a CSR matrix of ~d / sqrt(d) non-zeros per row;
// the library sklearn calls for it (scipy sparsetools csr_matvecs) adds, per output element, the products  data[jj] * X[i, indices[jj]]
// one by one in the order of the row's (sorted) non-zeros, each product and each sum rounded to float32.  This kernel does exactly
// that (this file is built with -ffp-contract=off; the intrinsics below pin the two roundings anyway), so the projected library --
// and with it every coreset pick -- is bit-identical to the host's (tests/test_gpu_engine.py).
#include "common.h"

namespace {

// One block: `rows` consecutive library rows staged in LDS, every thread walks the non-zeros of its output columns for all of them.
template <int ROWS>
__global__ __launch_bounds__(256) void sparse_project_kernel(const float* __restrict__ X, size_t n, int d, const int* __restrict__ indptr,
                                                             const int* __restrict__ indices, const float* __restrict__ data, int n_comp,
                                                             float* __restrict__ out)
{
    extern __shared__ float xs[];   // [ROWS][d]
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    const int live = (int)min((size_t)ROWS, n - r0);
    for (int e = threadIdx.x; e < live * d; e += 256) xs[e] = X[r0 * d + e];
    __syncthreads();
    for (int j = threadIdx.x; j < n_comp; j += 256) {
        float acc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r] = 0.0f;
        const int b = indptr[j], e = indptr[j + 1];
        for (int jj = b; jj < e; ++jj) {
            const float a = data[jj];
            const int c = indices[jj];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r] = __fadd_rn(acc[r], __fmul_rn(a, xs[r * d + c]));   // (rows >= live read stale LDS: never stored)
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            if (r < live) out[(r0 + r) * n_comp + j] = acc[r];
    }
}

}  // namespace

extern "C" int cmdiad_sparse_project_f32(const float* X, size_t n, int d, const int* indptr, const int* indices, const float* data,
                                         int n_comp, float* out, cmdiad_stream_t stream)
{
    CMDIAD_REQUIRE(X && indptr && indices && data && out && n > 0 && d > 0 && n_comp > 0, CMDIAD_ERR_ARG, "cmdiad_sparse_project_f32: bad args");
    CMDIAD_REQUIRE(d <= 4096, CMDIAD_ERR_ARG, "cmdiad_sparse_project_f32: d=%d exceeds the 4096 columns staged per row", d);
    hipStream_t s = (hipStream_t)stream;
    if (d <= 1024) {   // 8 rows x 4 KiB
        constexpr int R = 8;
        hipLaunchKernelGGL(sparse_project_kernel<R>, dim3((unsigned)((n + R - 1) / R)), dim3(256), (size_t)R * d * sizeof(float), s, X, n, d,
                           indptr, indices, data, n_comp, out);
    } else {
        constexpr int R = 2;
        hipLaunchKernelGGL(sparse_project_kernel<R>, dim3((unsigned)((n + R - 1) / R)), dim3(256), (size_t)R * d * sizeof(float), s, X, n, d,
                           indptr, indices, data, n_comp, out);
    }
    CMDIAD_CHECK_LAUNCH();
    return CMDIAD_OK;
}
