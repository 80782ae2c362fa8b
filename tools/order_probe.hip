// tools/order_probe.hip - does a kernel on one stream start before kernels that were SUBMITTED EARLIER to another stream
// (a different hardware queue) and are still waiting there behind a running kernel?
//   hipcc -O3 --offload-arch=gfx950 -o build/order_probe tools/order_probe.hip && build/order_probe
// Stream a: spin 3 ms, then a second kernel; stream b (submitted after both): a time stamp. With and without an event
// dependency of b on something long finished.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
__global__ void spin(long long ticks, long long *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void mark(long long *out) {
    if (threadIdx.x == 0) *out = wall_clock64();
}
int main() {
    long long *d, h[8];
    CK(hipMalloc(&d, sizeof h));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int variant = 0; variant < 4; ++variant) {
        hipStream_t a, b;
        CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0));
        CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, (variant & 1) ? greatest : 0));
        hipEvent_t done;
        CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, b, d + 6); // something long finished, with an event behind it
        CK(hipEventRecord(done, b));
        CK(hipDeviceSynchronize());
        CK(hipMemset(d, 0, sizeof h));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 300000ll, d);     // 3 ms
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100000ll, d + 2); // 1 ms, waits behind the first
        if (variant & 2) CK(hipStreamWaitEvent(b, done, 0));
        hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, b, d + 4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        printf("stream b %s priority, %s: first kernel of a %.3f -> %.3f ms, second %.3f -> %.3f ms, b's kernel at %.3f ms\n",
               (variant & 1) ? "high" : "normal", (variant & 2) ? "behind a completed event" : "no dependency",
               0.0, (h[1] - h[0]) / 1e5, (h[2] - h[0]) / 1e5, (h[3] - h[0]) / 1e5, (h[4] - h[0]) / 1e5);
        CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b)); CK(hipEventDestroy(done));
    }
    return 0;
}
