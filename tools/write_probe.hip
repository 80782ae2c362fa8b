// tools/write_probe.hip - what HBM takes in stores for the store shapes of track mode.
//   hipcc -O3 --offload-arch=gfx950 -o build/write_probe tools/write_probe.hip && build/write_probe
// No arithmetic: every kernel only stores, so the figures are the memory system's ceiling for the shape.
//   wave_tracks<MODE, CONSEC>: k_tracks20s' shape - a wave owns 16 "proteins" of LEN positions and walks them end to end,
//     256 positions per iteration, eight tracks of doubles.
//       MODE 0: lane l holds positions 4l..4l+3 and stores them one by one (8 bytes, lanes 32 bytes apart) - round 3
//       MODE 1: the same in two 16-byte stores                                                           - round 4
//       MODE 2: transposed, store j takes position 64j + l: 512 contiguous bytes per instruction
//       MODE 3: transposed in pairs: 1 KiB contiguous per instruction
//     CONSEC 0: protein k of wave b is item b + k * waves (the library's deal of the descending-length plan);
//     CONSEC 1: items 16b .. 16b+15, i.e. a wave's output is one contiguous region
//   lane_tracks<PITCH>: k_fwd_post's shape - a lane owns a protein and stores 16 positions (128 bytes) per iteration
//     to each of two tracks; PITCH sets the alignment of every protein's start
//   fill: grid-stride 16-byte stores (the ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int NTRACK = 8;
constexpr uint32_t LEN = 302;           // positions per protein (not a multiple of 4: starts are 8-byte aligned only)
constexpr uint32_t NPROT = 1u << 20;    // 1 Mi proteins: 317 M positions, 2.5 GB per track set of 8
constexpr size_t NPOS = (size_t)LEN * NPROT;

struct Tracks { double *t[NTRACK]; };

template <int MODE, int CONSEC>
__global__ __launch_bounds__(64) void wave_tracks(Tracks T, uint32_t waves) {
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    for (uint32_t k = 0; k < 16; ++k) {
        const uint32_t item = CONSEC ? b * 16u + k : b + k * waves;
        const size_t off = (size_t)item * LEN;
        for (uint32_t c = 0; c < LEN; c += 256u) {
#pragma unroll
            for (int tr = 0; tr < NTRACK; ++tr) {
                double *p = T.t[tr] + off + c;
                const double v = (double)(tr + c);
                if (MODE == 0) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) { const uint32_t i = 4u * lane + j; if (c + i < LEN) p[i] = v; }
                } else if (MODE == 1) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; j += 2) {
                        const uint32_t i = 4u * lane + j;
                        if (c + i + 1 < LEN) *reinterpret_cast<double2 *>(p + i) = make_double2(v, v);
                        else if (c + i < LEN) p[i] = v;
                    }
                } else if (MODE == 2) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) { const uint32_t i = 64u * j + lane; if (c + i < LEN) p[i] = v; }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 2; ++j) {
                        const uint32_t i = 128u * j + 2u * lane;
                        if (c + i + 1 < LEN) *reinterpret_cast<double2 *>(p + i) = make_double2(v, v);
                        else if (c + i < LEN) p[i] = v;
                    }
                }
            }
        }
    }
}

template <uint32_t PITCH> // doubles between the starts of consecutive proteins (<= 304): sets the alignment of the runs
__global__ __launch_bounds__(256) void lane_tracks(Tracks T, uint32_t nprot) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= nprot) return;
    double *p0 = T.t[0] + (size_t)gid * PITCH, *p1 = T.t[1] + (size_t)gid * PITCH;
    for (uint32_t t0 = 0; t0 + 16u <= 288u; t0 += 16u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            *reinterpret_cast<double2 *>(p0 + t0 + 2 * k) = make_double2(1.0, (double)t0);
            *reinterpret_cast<double2 *>(p1 + t0 + 2 * k) = make_double2(2.0, (double)t0);
        }
    }
}

// the same bytes as lane_tracks, but a store instruction covers RUN consecutive doubles of 64 / RUN proteins (what a
// transposition through LDS would buy k_fwd_post): RUN = 16: 128-byte runs, RUN = 8: 64-byte runs
template <uint32_t PITCH, uint32_t RUN>
__global__ __launch_bounds__(256) void lane_tracks_runs(Tracks T, uint32_t nprot) {
    const uint32_t lane = threadIdx.x & 63u, wave0 = (blockIdx.x * 256u + threadIdx.x) & ~63u; // first protein of the wave
    if (wave0 >= nprot) return;
    const uint32_t sub = lane % RUN, grp = lane / RUN;
    for (uint32_t t0 = 0; t0 + 16u <= 288u; t0 += RUN) {
#pragma unroll
        for (uint32_t k = 0; k < RUN; ++k) { // 64 proteins, 64 / RUN per instruction
            const uint32_t prot = wave0 + k * (64u / RUN) + grp;
            T.t[0][(size_t)prot * PITCH + t0 + sub] = 1.0;
            T.t[1][(size_t)prot * PITCH + t0 + sub] = 2.0;
        }
    }
}

__global__ __launch_bounds__(256) void fill(double2 *p, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256u) p[i] = make_double2(1.0, 2.0);
}

template <class F>
static double timed_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 3; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3.0;
}

int main() {
    Tracks T;
    const size_t per_track = ((size_t)((LEN + 15u) & ~15u) * NPROT + 1024) * sizeof(double);
    for (int k = 0; k < NTRACK; ++k) CK(hipMalloc(&T.t[k], per_track));
    const uint32_t waves = NPROT / 16u;
    const double gb8 = 8.0 * NPOS * NTRACK / 1e9, gb2 = 8.0 * 288.0 * NPROT * 2 / 1e9;
    double ms;
    ms = timed_ms([&] { hipLaunchKernelGGL(fill, dim3(256 * 16), dim3(256), 0, 0, (double2 *)T.t[0], NPOS / 2); });
    printf("fill, 16-byte stores, one array                          : %7.3f ms  %6.2f TB/s\n", ms, 8.0 * NPOS / 1e9 / ms);
#define RUN(M, C, what) \
    ms = timed_ms([&] { hipLaunchKernelGGL((wave_tracks<M, C>), dim3(waves), dim3(64), 0, 0, T, waves); }); \
    printf("%-57s: %7.3f ms  %6.2f TB/s\n", what, ms, gb8 / ms);
    RUN(0, 0, "wave, 8 tracks, 4 x 8 B per lane, dealt proteins")
    RUN(1, 0, "wave, 8 tracks, 2 x 16 B per lane, dealt proteins")
    RUN(2, 0, "wave, 8 tracks, transposed 8 B, dealt proteins")
    RUN(3, 0, "wave, 8 tracks, transposed 16 B, dealt proteins")
    RUN(0, 1, "wave, 8 tracks, 4 x 8 B per lane, consecutive proteins")
    RUN(1, 1, "wave, 8 tracks, 2 x 16 B per lane, consecutive proteins")
    RUN(2, 1, "wave, 8 tracks, transposed 8 B, consecutive proteins")
    RUN(3, 1, "wave, 8 tracks, transposed 16 B, consecutive proteins")
#define LANE(P, what) \
    ms = timed_ms([&] { hipLaunchKernelGGL(lane_tracks<P>, dim3(NPROT / 256u), dim3(256), 0, 0, T, NPROT); }); \
    printf("%-57s: %7.3f ms  %6.2f TB/s\n", what, ms, gb2 / ms);
    LANE(303, "lane per protein, 2 tracks, 128 B runs, 8 B aligned")
    LANE(302, "lane per protein, 2 tracks, 128 B runs, 16 B aligned")
    LANE(300, "lane per protein, 2 tracks, 128 B runs, 32 B aligned")
    LANE(296, "lane per protein, 2 tracks, 128 B runs, 64 B aligned")
    LANE(304, "lane per protein, 2 tracks, 128 B runs, 128 B aligned")
#define LANER(P, R, what) \
    ms = timed_ms([&] { hipLaunchKernelGGL((lane_tracks_runs<P, R>), dim3(NPROT / 256u), dim3(256), 0, 0, T, NPROT); }); \
    printf("%-57s: %7.3f ms  %6.2f TB/s\n", what, ms, gb2 / ms);
    LANER(303, 16, "the same, 16 lanes per 128 B run, 8 B aligned")
    LANER(304, 16, "the same, 16 lanes per 128 B run, 128 B aligned")
    LANER(303, 8, "the same, 8 lanes per 64 B run, 8 B aligned")
    LANER(304, 8, "the same, 8 lanes per 64 B run, 64 B aligned")
    return 0;
}
