// tools/fetch_probe.hip - what FETCH_SIZE / WRITE_SIZE report on gfx950 for the access shapes of this repo's kernels.
//   hipcc -O3 --offload-arch=gfx950 -o build/fetch_probe tools/fetch_probe.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- build/fetch_probe   (and again with WRITE_SIZE)
// Every kernel touches a fresh 512 MiB region exactly once, so the true HBM bytes are known:
//   rd_1B / rd_4B / rd_16B : a wave reads 64 x {1,4,16} consecutive bytes per load (lane-contiguous)
//   rd_4B_overlap          : K-B's stage 0: two 4-byte loads per lane, 4 bytes apart, lanes 4 bytes apart
//   rd_16B_lane_rows       : k_pack's input side: every lane walks its own contiguous run, 16 bytes per load
//   wr_52_of_160           : K-B's row update: 52 contiguous bytes of every 160-byte row (partial-line stores)
//   wr_16B                 : full coalesced 16 B/lane stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr size_t REGION = 512ull << 20;

__global__ void rd_1B(const uint8_t *p, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < REGION; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void rd_4B(const uint32_t *p, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < REGION / 4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void rd_16B(const uint4 *p, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < REGION / 16; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x + p[i].w;
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void rd_4B_overlap(const uint8_t *p, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 2 < REGION / 4; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t a, b;
        __builtin_memcpy(&a, p + 4 * i + 2, 4); // unaligned, like x + q0 - 2
        __builtin_memcpy(&b, p + 4 * i + 6, 4);
        acc += a ^ b;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void rd_16B_lane_rows(const uint4 *p, unsigned *out) { // lane owns a contiguous 4 KiB run
    unsigned acc = 0;
    const size_t lanes = (size_t)gridDim.x * blockDim.x, run = REGION / 16 / lanes;
    const size_t me = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t j = 0; j < run; ++j) acc += p[me * run + j].x;
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void wr_52_of_160(uint8_t *p) {
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < REGION / 160; r += (size_t)gridDim.x * blockDim.x) {
        double *d = (double *)(p + r * 160 + 64); // 5 doubles + 3 ints, as K-B's fields lie in plaac_row
        d[0] = 1.0; d[1] = 2.0; d[2] = 3.0; d[3] = 4.0; d[4] = 5.0;
        int *q = (int *)(p + r * 160 + 148);
        q[0] = 1; q[1] = 2; q[2] = 3;
    }
}
__global__ void wr_16B(uint4 *p) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < REGION / 16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}

int main() {
    uint8_t *buf;
    unsigned *out;
    const int NK = 7;
    CK(hipMalloc(&buf, REGION * NK));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, REGION * NK));
    CK(hipDeviceSynchronize());
    const dim3 g(256 * 8), b(256);
    hipLaunchKernelGGL(rd_1B, g, b, 0, 0, buf + 0 * REGION, out);
    hipLaunchKernelGGL(rd_4B, g, b, 0, 0, (const uint32_t *)(buf + 1 * REGION), out);
    hipLaunchKernelGGL(rd_16B, g, b, 0, 0, (const uint4 *)(buf + 2 * REGION), out);
    hipLaunchKernelGGL(rd_4B_overlap, g, b, 0, 0, buf + 3 * REGION, out);
    hipLaunchKernelGGL(rd_16B_lane_rows, dim3(256), dim3(64), 0, 0, (const uint4 *)(buf + 4 * REGION), out);
    hipLaunchKernelGGL(wr_52_of_160, g, b, 0, 0, buf + 5 * REGION);
    hipLaunchKernelGGL(wr_16B, g, b, 0, 0, (uint4 *)(buf + 6 * REGION));
    CK(hipDeviceSynchronize());
    printf("region %zu bytes per kernel\n", REGION);
    return 0;
}
