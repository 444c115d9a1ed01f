// Which physical CU does bit b of a hipExtStreamCreateWithCUMask mask enable?  For every bit: a stream with only that bit,
// one workgroup that reads HW_REG_XCC_ID and HW_REG_HW_ID.  Build: hipcc --offload-arch=gfx950 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[0] = xcc; out[1] = hw; }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    unsigned* d; hipMalloc(&d, 8);
    printf("ncu %d\n", ncu);
    for (int b = 0; b < ncu; ++b) {
        std::vector<uint32_t> m(words, 0u); m[b / 32] = 1u << (b % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, words, m.data()) != hipSuccess) { printf("bit %d: create failed\n", b); continue; }
        hipLaunchKernelGGL(who, dim3(1), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        // HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
        printf("bit %3d -> xcc %u se %u sh %u cu %u\n", b, h[0] & 0xf, (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 0xf);
        hipStreamDestroy(s);
    }
    return 0;
}
