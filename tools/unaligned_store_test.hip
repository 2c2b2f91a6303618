// Does a 16-byte global store to a 2-byte-aligned address work on this device / runtime?  (qkv V^T epilogue design question)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(unsigned short* p, int off)
{
    uint4 v = make_uint4(0x00010000u + threadIdx.x, 0x00030002u, 0x00050004u, 0x00070006u);
    *reinterpret_cast<uint4*>(p + off + threadIdx.x * 16) = v;   // byte address 2 * off + 32 * tid: 2-byte aligned when off is odd
    uint2 w = make_uint2(0xAAAA0000u + threadIdx.x, 0xBBBBCCCCu);
    *reinterpret_cast<uint2*>(p + 4096 + off + threadIdx.x * 16) = w;
}
int main()
{
    unsigned short* d; hipMalloc(&d, 65536); hipMemset(d, 0, 65536);
    for (int off : {0, 1, 3, 5}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, off);
        hipError_t e = hipDeviceSynchronize();
        unsigned short h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        bool ok = e == hipSuccess;
        for (int t = 0; t < 64 && ok; ++t) {
            const unsigned short* q = h + off + t * 16;
            ok = q[0] == t && q[1] == 1 && q[2] == 2 && q[3] == 3 && q[4] == 4 && q[5] == 5 && q[6] == 6 && q[7] == 7;
            const unsigned short* r = h + 4096 + off + t * 16;
            ok = ok && r[0] == t && r[1] == 0xAAAA && r[2] == 0xCCCC && r[3] == 0xBBBB;
        }
        printf("offset %d halfwords: %s (%s)\n", off, ok ? "correct" : "WRONG", hipGetErrorString(e));
        hipMemset(d, 0, 65536);
    }
    return 0;
}
