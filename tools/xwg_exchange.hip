// What one round of a cross-workgroup exchange through L2 costs on this device: NWG co-resident workgroups (one per XCD-0 CU: block
// ids 0, 8, 16, ... land on the same XCD) each publish a 64-bit key with atomicMax, bump a per-round arrival counter, and spin until
// all NWG have arrived, then read the winner -- the per-round step a farthest-point-sampling chain split over several
// workgroups would need 1 023 times per cloud.   hipcc --offload-arch=gfx950 -O3 tools/xwg_exchange.hip -o /tmp/xwg && /tmp/xwg
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void exchange(unsigned long long* keys, unsigned* arrived, int rounds, int nwg, unsigned long long* out)
{
    if (blockIdx.x % 8 != 0 || (int)(blockIdx.x / 8) >= nwg) return;   // the participants: blocks 0, 8, 16, ... (one XCD)
    const int me = blockIdx.x / 8;
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            atomicMax(keys + r, ((unsigned long long)(r * 131 + me * 7919) << 8) | (unsigned)me);
            __threadfence();
            atomicAdd(arrived + r, 1u);
            while (__hip_atomic_load(arrived + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {}
            acc += __hip_atomic_load(keys + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();   // the workgroup-wide hand-off a real round also needs
    }
    if (threadIdx.x == 0) out[me] = acc;
}
int main()
{
    const int rounds = 1023;
    unsigned long long *keys, *out; unsigned* arrived;
    hipMalloc(&keys, rounds * 8); hipMalloc(&arrived, rounds * 4); hipMalloc(&out, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nwg : {1, 2, 4}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(keys, 0, rounds * 8); hipMemset(arrived, 0, rounds * 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(exchange, dim3(8 * nwg), dim3(512), 0, 0, keys, arrived, rounds, nwg, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("xwg_exchange nwg=%d: %.3f ms per %d rounds = %.3f us per round\n", nwg, best, rounds, best * 1e3f / rounds);
    }
    return 0;
}
