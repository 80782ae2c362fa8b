// tools/issue_probe.hip - issue cost of the instruction classes this path is made of, on gfx950 (VERDICT r02 weak #6:
// "4 cycles for every VALU and LDS wave-instruction" was an assumption; the guide's table says 2 for 32-bit VALU with
// more than one wave per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o build/issue_probe tools/issue_probe.hip && build/issue_probe
// One-wave blocks, W waves per SIMD (grid = CUs * 4 * W; a CU places consecutive one-wave blocks round-robin on its four
// SIMDs). Every wave issues ITER x 64 instructions of one class from `CH` independent dependency chains. Output: SIMD
// cycles per wave-instruction = launch cycles * (SIMDs) / (wave-instructions), at the clock hipDeviceProp reports - the
// same accounting bench.py's `roofline.issue` uses (wave-instructions x cost / 1024 SIMDs / clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

enum Kind { ADD_U32 = 0, MOV_B32, ADD_F64, MUL_F64, FMA_F64, MOV_DPP, MAD_I24, CNDMASK, BFE_U32, ADD_F32, LSHL_ADD, CMP_F64, ADD_U32_F64_MIX,
            DS_READ_B64, SALU_ADD, NKIND };
static const char *names[NKIND] = {"v_add_u32", "v_mov_b32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_mov_b32 dpp row_shr:1",
                                   "v_mad_i32_i24", "v_cndmask_b32", "v_bfe_u32", "v_add_f32", "v_lshl_add_u32", "v_cmp_gt_f64 (vcc)",
                                   "v_add_f64 + 3 v_add_u32 (per 4 instr)", "ds_read_b64 (conflict-free) + waitcnt per 16", "s_add_u32"};

template <int KIND, int CH>
__global__ __launch_bounds__(64) void probe(unsigned *out, int iters) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (double)i;
    __syncthreads();
    unsigned u[16];
    double d[16];
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = lane * 7 + i, d[i] = (double)(lane + i) * 1e-3, f[i] = (float)(lane + i);
    unsigned s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
    const unsigned addr = lane * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            const int c = k % CH;
            if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));
            if (KIND == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(u[(c + 1) % CH]));
            if (KIND == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));
            if (KIND == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));
            if (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[c]) : "v"(d[15]));
            if (KIND == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]) : "v"(u[(c + 1) % CH]));
            if (KIND == MAD_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));
            if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15]));
            if (KIND == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[c]));
            if (KIND == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[15]));
            if (KIND == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[c]) : "v"(u[15]));
            if (KIND == CMP_F64) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[c]), "v"(d[15]) : "vcc");
            if (KIND == ADD_U32_F64_MIX) {
                if (k % 4 == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));
            }
            if (KIND == DS_READ_B64) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(d[c]) : "v"(addr));
                if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (KIND == SALU_ADD) {
                if (c % 4 == 0) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s3) : "scc");
                if (c % 4 == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s3) : "scc");
                if (c % 4 == 2) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s3) : "scc");
                if (c % 4 == 3) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
            }
        }
    }
    unsigned acc = s0 + s1 + s2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += u[i] + (unsigned)d[i] + (unsigned)f[i];
    out[blockIdx.x * 64 + lane] = acc;
}

static double g_ghz;
static int g_cus;

template <int KIND, int CH>
static void run(int wps, unsigned *d_out) {
    const int ITER = 2000;
    const int grid = g_cus * 4 * wps;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<KIND, CH>), dim3(grid), dim3(64), 0, 0, d_out, 100);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((probe<KIND, CH>), dim3(grid), dim3(64), 0, 0, d_out, ITER);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double cyc = best * 1e-3 * g_ghz * 1e9;
    const double per_simd = (double)wps * ITER * 64.0; // wave-instructions per SIMD
    printf("%-46s chains %2d  waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", names[KIND], CH, wps,
           best, cyc / per_simd);
}

template <int KIND>
static void sweep(unsigned *d_out) {
    for (int wps = 1; wps <= 4; ++wps) run<KIND, 8>(wps, d_out);
    run<KIND, 1>(1, d_out); // one dependent chain, one wave: the latency of the class
    run<KIND, 1>(4, d_out);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    g_ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, clock %.3f GHz (cycles below are at this clock)\n", p.gcnArchName, g_cus, g_ghz);
    unsigned *d_out;
    CK(hipMalloc(&d_out, sizeof(unsigned) * 64 * g_cus * 16));
    sweep<ADD_U32>(d_out);
    sweep<MOV_B32>(d_out);
    sweep<LSHL_ADD>(d_out);
    sweep<BFE_U32>(d_out);
    sweep<MAD_I24>(d_out);
    sweep<CNDMASK>(d_out);
    sweep<MOV_DPP>(d_out);
    sweep<ADD_F32>(d_out);
    sweep<ADD_F64>(d_out);
    sweep<MUL_F64>(d_out);
    sweep<FMA_F64>(d_out);
    sweep<CMP_F64>(d_out);
    sweep<ADD_U32_F64_MIX>(d_out);
    sweep<DS_READ_B64>(d_out);
    sweep<SALU_ADD>(d_out);
    return 0;
}
