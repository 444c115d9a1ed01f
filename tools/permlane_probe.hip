// Probe of the gfx950 v_permlane16_swap / v_permlane32_swap semantics used to broadcast one 16-lane row
// of a wave to all four rows without going through the LDS crossbar (ds_bpermute).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned x = threadIdx.x;
    auto ab = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    auto c = __builtin_amdgcn_permlane32_swap(ab[0], ab[0], false, false);
    auto d = __builtin_amdgcn_permlane32_swap(ab[1], ab[1], false, false);
    out[threadIdx.x] = ab[0]; out[64 + threadIdx.x] = ab[1];
    out[128 + threadIdx.x] = c[0]; out[192 + threadIdx.x] = c[1];
    out[256 + threadIdx.x] = d[0]; out[320 + threadIdx.x] = d[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 384 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"swap16[0]", "swap16[1]", "s32(a)[0]", "s32(a)[1]", "s32(b)[0]", "s32(b)[1]"};
    for (int r = 0; r < 6; ++r) {
        printf("%-10s rows:", nm[r]);
        for (int q = 0; q < 4; ++q) printf(" [%2u..%2u]", h[r * 64 + q * 16], h[r * 64 + q * 16 + 15]);
        printf("\n");
    }
    return 0;
}
