// tools/queue_probe.hip - which HIP streams share a hardware queue on this runtime, and what sharing costs.
//   hipcc -O3 --offload-arch=gfx950 -o build/queue_probe tools/queue_probe.hip && build/queue_probe
// N streams are created (normal / high priority), optionally "touched" in a given order, then for every ordered pair
// (a, b): a one-block kernel that spins for 2 ms goes to stream a, a one-block kernel that records the time to stream b.
// b's kernel finishing before a's spin ends = the two streams run side by side; after it = they are serialised
// (same hardware queue: the packets of one queue are processed in order, and the runtime puts a barrier between packets
// of different streams). The collision matrix shows the assignment policy (by creation order? by first use? modulo 4?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void spin(long long ticks, long long *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) *out = wall_clock64();
}
__global__ void mark(long long *out) {
    if (threadIdx.x == 0) *out = wall_clock64();
}

static void matrix(const char *title, std::vector<hipStream_t> &st, long long *d_t) {
    const int n = (int)st.size();
    printf("%s\n     ", title);
    for (int b = 0; b < n; ++b) printf("%3d", b);
    printf("\n");
    for (int a = 0; a < n; ++a) {
        printf("%3d: ", a);
        for (int b = 0; b < n; ++b) {
            if (a == b) {
                printf("  .");
                continue;
            }
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[a], 200000ll /* 2 ms at 100 MHz */, d_t);
            hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, st[b], d_t + 1);
            CK(hipStreamSynchronize(st[a]));
            CK(hipStreamSynchronize(st[b]));
            long long h[2];
            CK(hipMemcpy(h, d_t, sizeof h, hipMemcpyDeviceToHost));
            printf("  %c", h[1] < h[0] ? '-' : 'X'); // X: b waited for a
        }
        printf("\n");
    }
}

int main(int argc, char **argv) {
    long long *d_t;
    CK(hipMalloc(&d_t, 16));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("priority range: least %d greatest %d\n", least, greatest);
    {
        std::vector<hipStream_t> st(10);
        for (auto &s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
        matrix("10 normal-priority streams, used in creation order (X = column stream waited for row stream)", st, d_t);
        matrix("the same streams again", st, d_t);
        for (auto &s : st) CK(hipStreamDestroy(s));
    }
    {
        std::vector<hipStream_t> st(10);
        for (auto &s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
        for (int i = 9; i >= 0; --i) { // first use in reverse order
            hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, st[i], d_t + 1);
            CK(hipStreamSynchronize(st[i]));
        }
        matrix("10 normal-priority streams, FIRST USED in reverse creation order", st, d_t);
        for (auto &s : st) CK(hipStreamDestroy(s));
    }
    {
        std::vector<hipStream_t> st(12);
        for (int i = 0; i < 6; ++i) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, 0));
        for (int i = 6; i < 12; ++i) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, greatest));
        matrix("streams 0-5 normal, 6-11 high priority", st, d_t);
        for (auto &s : st) CK(hipStreamDestroy(s));
    }
    {
        std::vector<hipStream_t> st(5);
        for (auto &s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0));
        st.push_back(nullptr); // the null stream
        matrix("5 normal streams + the null stream (index 5)", st, d_t);
    }
    return 0;
}
