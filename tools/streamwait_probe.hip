// Probe: cost of cross-stream signalling on this device (tools only; not part of the library).
//   1. is hipStreamWaitValue64 supported;  2. latency kernel-end -> dependent kernel start through (a) an event, (b) a wait-value
//   on signal memory written by the producer kernel;  3. what a hipEventRecord / a satisfied wait-value between two kernels
//   of ONE stream adds to the gap between them.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/streamwait_probe tools/streamwait_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stamp_kernel(unsigned long long* t, int slot, int spin_us, unsigned long long* flag, unsigned long long val) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        t[2 * slot] = t0;
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(4);
        if (flag) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long* t;
    CK(hipMalloc(&t, 64 * 8));
    unsigned long long* sig = nullptr;
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(8 bytes, hipMallocSignalMemory) -> %s\n", hipGetErrorString(es));
    if (es != hipSuccess) {
        (void)hipGetLastError();
        es = hipMalloc((void**)&sig, 8);
        printf("falling back to hipMalloc memory for the wait-value word -> %s\n", hipGetErrorString(es));
    }
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::vector<unsigned long long> h(64);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const int R = 21;
    // --- same-stream gaps
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && (!can || es != hipSuccess)) continue;
        std::vector<double> g;
        for (int r = 0; r < R; ++r) {
            if (sig) { CK(hipMemset(sig, 0, 8)); }
            if (mode == 2) { unsigned long long one = 1; CK(hipMemcpy(sig, &one, 8, hipMemcpyHostToDevice)); }
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, 0, 20, (unsigned long long*)nullptr, 0ull);
            if (mode == 1) CK(hipEventRecord(ev, a));
            if (mode == 2) CK(hipStreamWaitValue64(a, sig, 1, hipStreamWaitValueGte, ~0ull));
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, 1, 1, (unsigned long long*)nullptr, 0ull);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), t, 64 * 8, hipMemcpyDeviceToHost));
            g.push_back((double)(h[2] - h[1]) / 100.0);
        }
        printf("same stream, K1 -> %s -> K2: gap %.2f us (median of %d)\n", mode == 0 ? "(nothing)" : mode == 1 ? "hipEventRecord" : "satisfied hipStreamWaitValue64", med(g), R);
    }
    // --- cross-stream latency
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1 && (!can || es != hipSuccess)) continue;
        std::vector<double> g, g2;
        for (int r = 0; r < R; ++r) {
            if (sig) CK(hipMemset(sig, 0, 8));
            CK(hipDeviceSynchronize());
            if (mode == 0) {
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, 0, 20, (unsigned long long*)nullptr, 0ull);
                CK(hipEventRecord(ev, a));
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, 2, 1, (unsigned long long*)nullptr, 0ull);   // next kernel of the recording stream
                CK(hipStreamWaitEvent(b, ev, 0));
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, b, t, 1, 1, (unsigned long long*)nullptr, 0ull);
            } else {
                CK(hipStreamWaitValue64(b, sig, 1, hipStreamWaitValueGte, ~0ull));
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, b, t, 1, 1, (unsigned long long*)nullptr, 0ull);
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, 0, 20, sig, 1ull);
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, 2, 1, (unsigned long long*)nullptr, 0ull);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), t, 64 * 8, hipMemcpyDeviceToHost));
            g.push_back(((double)h[2] - (double)h[1]) / 100.0);
            g2.push_back(((double)h[4] - (double)h[1]) / 100.0);
        }
        printf("cross stream via %s: producer end -> consumer start %.2f us; producer end -> producer's next kernel %.2f us\n",
               mode == 0 ? "event" : "wait-value on signal memory", med(g), med(g2));
    }
    return 0;
}
