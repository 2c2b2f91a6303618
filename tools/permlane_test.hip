#include <hip/hip_runtime.h>
__global__ void k(unsigned* o)
{
    unsigned a = o[threadIdx.x], b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    o[64 + threadIdx.x] = a; o[128 + threadIdx.x] = b;
    unsigned c = o[threadIdx.x], d = c;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    o[192 + threadIdx.x] = c; o[256 + threadIdx.x] = d;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096); unsigned h[320]; for (int i = 0; i < 64; ++i) h[i] = i; hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 1280, hipMemcpyDeviceToHost);
    for (int s = 1; s < 5; ++s) { printf("%d:", s); for (int i = 0; i < 64; i += 1) printf(" %u", h[64 * s + i]); printf("\n"); }
    return 0;
}
