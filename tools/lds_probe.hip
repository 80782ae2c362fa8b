// tools/lds_probe.hip - LDS instruction cost on gfx950 for the access patterns the window kernel (K-B) uses.
//   hipcc -O3 --offload-arch=gfx950 -o build/lds_probe tools/lds_probe.hip && build/lds_probe
// Blocks of ONE wave, `wpc` waves per CU (grid = CUs * wpc). Each wave issues ITER x 16 LDS instructions of one
// kind, drained by one s_waitcnt per 16, with `ADDS` independent fp64 adds per LDS instruction beside them.
// Output: cycles per LDS instruction per CU (device clock from hipDeviceProp) - compare variants, not absolutes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int ITER = 4000;
typedef double d2 __attribute__((ext_vector_type(2)));

enum Kind { NONE = 0, B64_S8, B64_S16, READ2_NEAR, READ2_FAR, B128_S8, B128_S16, U8_S1, B32_S4, B32_UNAL, B64_TABLE, B64_TABLE3, NKIND };
static const char *names[NKIND] = {"no LDS (adds only)", "ds_read_b64   lane stride 8", "ds_read_b64   lane stride 16",
                                   "ds_read2_b64  slots k,k+1, stride 8", "ds_read2_b64  slots k,k+86, stride 8",
                                   "ds_read_b128  lane stride 8", "ds_read_b128  lane stride 16", "ds_read_u8    consecutive bytes",
                                   "ds_read_b32   lane stride 4", "ds_read_b32   lane stride 1 (unaligned)",
                                   "ds_read_b64   25-row table, random row", "ds_read_b64   3 tables x random row"};

#define RD(instr, dst, off) asm volatile(instr " %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define RD2(dst, o0, o1) asm volatile("ds_read2_b64 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(dst) : "v"(addr))

template <int KIND, int ADDS>
__global__ __launch_bounds__(64) void probe(double *out, const unsigned *codes, int iters) {
    __shared__ __attribute__((aligned(16))) double lds[1536];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1536; i += 64) lds[i] = (double)i;
    __syncthreads();
    unsigned addr = 0;
    if (KIND == B64_S8 || KIND == READ2_NEAR || KIND == READ2_FAR || KIND == B128_S8) addr = lane * 8;
    if (KIND == B64_S16 || KIND == B128_S16) addr = lane * 16;
    if (KIND == U8_S1 || KIND == B32_UNAL) addr = lane;
    if (KIND == B32_S4) addr = lane * 4;
    double acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (double)(lane + i);
    unsigned c = codes[blockIdx.x * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        if (KIND == B64_TABLE || KIND == B64_TABLE3) { // a new random row per iteration, as the residue codes give
            c = c * 1664525u + 1013904223u;
            addr = ((c >> 16) % 25u) * 8u;
        }
        double v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, va, vb, vc, vd, ve, vf;
        d2 w0, w1, w2, w3, w4, w5, w6, w7;
        unsigned u0, u1, u2, u3, u4, u5, u6, u7, u8, u9, ua, ub, uc, ud, ue, uf;
        if (KIND == B64_S8 || KIND == B64_S16 || KIND == B64_TABLE) {
            RD("ds_read_b64", v0, 0); RD("ds_read_b64", v1, 8); RD("ds_read_b64", v2, 16); RD("ds_read_b64", v3, 24);
            RD("ds_read_b64", v4, 32); RD("ds_read_b64", v5, 40); RD("ds_read_b64", v6, 48); RD("ds_read_b64", v7, 56);
            RD("ds_read_b64", v8, 64); RD("ds_read_b64", v9, 72); RD("ds_read_b64", va, 80); RD("ds_read_b64", vb, 88);
            RD("ds_read_b64", vc, 96); RD("ds_read_b64", vd, 104); RD("ds_read_b64", ve, 112); RD("ds_read_b64", vf, 120);
        } else if (KIND == B64_TABLE3) {
            RD("ds_read_b64", v0, 0); RD("ds_read_b64", v1, 256); RD("ds_read_b64", v2, 512); RD("ds_read_b64", v3, 0);
            RD("ds_read_b64", v4, 256); RD("ds_read_b64", v5, 512); RD("ds_read_b64", v6, 0); RD("ds_read_b64", v7, 256);
            RD("ds_read_b64", v8, 512); RD("ds_read_b64", v9, 0); RD("ds_read_b64", va, 256); RD("ds_read_b64", vb, 512);
            RD("ds_read_b64", vc, 0); RD("ds_read_b64", vd, 256); RD("ds_read_b64", ve, 512); RD("ds_read_b64", vf, 0);
        } else if (KIND == READ2_NEAR) { // 8 instructions = 16 values
            RD2(w0, 0, 1); RD2(w1, 2, 3); RD2(w2, 4, 5); RD2(w3, 6, 7); RD2(w4, 8, 9); RD2(w5, 10, 11); RD2(w6, 12, 13); RD2(w7, 14, 15);
        } else if (KIND == READ2_FAR) {
            RD2(w0, 0, 86); RD2(w1, 1, 87); RD2(w2, 2, 88); RD2(w3, 3, 89); RD2(w4, 4, 90); RD2(w5, 5, 91); RD2(w6, 6, 92); RD2(w7, 7, 93);
        } else if (KIND == B128_S8 || KIND == B128_S16) {
            RD("ds_read_b128", w0, 0); RD("ds_read_b128", w1, 16); RD("ds_read_b128", w2, 32); RD("ds_read_b128", w3, 48);
            RD("ds_read_b128", w4, 64); RD("ds_read_b128", w5, 80); RD("ds_read_b128", w6, 96); RD("ds_read_b128", w7, 112);
        } else if (KIND == U8_S1) {
            RD("ds_read_u8", u0, 0); RD("ds_read_u8", u1, 86); RD("ds_read_u8", u2, 172); RD("ds_read_u8", u3, 258);
            RD("ds_read_u8", u4, 1); RD("ds_read_u8", u5, 87); RD("ds_read_u8", u6, 173); RD("ds_read_u8", u7, 259);
            RD("ds_read_u8", u8, 2); RD("ds_read_u8", u9, 88); RD("ds_read_u8", ua, 174); RD("ds_read_u8", ub, 260);
            RD("ds_read_u8", uc, 3); RD("ds_read_u8", ud, 89); RD("ds_read_u8", ue, 175); RD("ds_read_u8", uf, 261);
        } else if (KIND == B32_S4 || KIND == B32_UNAL) {
            RD("ds_read_b32", u0, 0); RD("ds_read_b32", u1, 4); RD("ds_read_b32", u2, 8); RD("ds_read_b32", u3, 12);
            RD("ds_read_b32", u4, 16); RD("ds_read_b32", u5, 20); RD("ds_read_b32", u6, 24); RD("ds_read_b32", u7, 28);
            RD("ds_read_b32", u8, 32); RD("ds_read_b32", u9, 36); RD("ds_read_b32", ua, 40); RD("ds_read_b32", ub, 44);
            RD("ds_read_b32", uc, 48); RD("ds_read_b32", ud, 52); RD("ds_read_b32", ue, 56); RD("ds_read_b32", uf, 60);
        }
#pragma unroll
        for (int a = 0; a < 16 * ADDS; ++a) acc[a % 12] = acc[a % 12] + 1.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = 0.0;
    for (int i = 0; i < 12; ++i) s += acc[i];
    out[blockIdx.x * 64 + lane] = s;
}

template <int KIND, int ADDS>
static void run(int cus, int wpc, double ghz, double *d_out, unsigned *d_codes) {
    const int grid = cus * wpc;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<KIND, ADDS>), dim3(grid), dim3(64), 0, 0, d_out, d_codes, 200);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((probe<KIND, ADDS>), dim3(grid), dim3(64), 0, 0, d_out, d_codes, ITER);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double cyc = ms * 1e-3 * ghz * 1e9;              // cycles of the launch
    const int per_iter = (KIND == READ2_NEAR || KIND == READ2_FAR || KIND == B128_S8 || KIND == B128_S16) ? 8 : 16;
    const double instr_cu = (double)wpc * ITER * per_iter;  // LDS instructions per CU
    printf("%-42s adds/LDS %d  waves/CU %2d  %8.3f ms  %6.2f cyc per LDS instr per CU  %6.2f cyc per 16 values per wave\n",
           names[KIND], ADDS, wpc, ms, KIND == NONE ? 0.0 : cyc / instr_cu, cyc / ITER);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, clock %.2f GHz\n", p.gcnArchName, cus, ghz);
    double *d_out;
    unsigned *d_codes;
    CK(hipMalloc(&d_out, sizeof(double) * 64 * cus * 16));
    CK(hipMalloc(&d_codes, sizeof(unsigned) * 64 * cus * 16));
    unsigned *h = (unsigned *)malloc(sizeof(unsigned) * 64 * cus * 16);
    for (int i = 0; i < 64 * cus * 16; ++i) h[i] = (unsigned)rand();
    CK(hipMemcpy(d_codes, h, sizeof(unsigned) * 64 * cus * 16, hipMemcpyHostToDevice));
#define ALL(ADDS, WPC)                                                                                         \
    run<B64_S8, ADDS>(cus, WPC, ghz, d_out, d_codes); run<B64_S16, ADDS>(cus, WPC, ghz, d_out, d_codes);      \
    run<READ2_NEAR, ADDS>(cus, WPC, ghz, d_out, d_codes); run<READ2_FAR, ADDS>(cus, WPC, ghz, d_out, d_codes); \
    run<B128_S8, ADDS>(cus, WPC, ghz, d_out, d_codes); run<B128_S16, ADDS>(cus, WPC, ghz, d_out, d_codes);     \
    run<U8_S1, ADDS>(cus, WPC, ghz, d_out, d_codes); run<B32_S4, ADDS>(cus, WPC, ghz, d_out, d_codes);         \
    run<B32_UNAL, ADDS>(cus, WPC, ghz, d_out, d_codes); run<B64_TABLE, ADDS>(cus, WPC, ghz, d_out, d_codes);   \
    run<B64_TABLE3, ADDS>(cus, WPC, ghz, d_out, d_codes);
    printf("--- LDS only, 12 waves per CU\n");
    ALL(0, 12)
    printf("--- LDS only, 4 waves per CU\n");
    ALL(0, 4)
    printf("--- with 4 fp64 adds per LDS instruction slot (K-B at 4 positions per lane), 12 waves per CU\n");
    run<NONE, 4>(cus, 12, ghz, d_out, d_codes);
    ALL(4, 12)
    printf("--- with 2 fp64 adds per LDS instruction slot, 12 waves per CU\n");
    run<NONE, 2>(cus, 12, ghz, d_out, d_codes);
    ALL(2, 12)
    return 0;
}
