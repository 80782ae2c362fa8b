#!/usr/bin/env python3
"""tools/make_issue_probe.py - writes tools/issue_probe.hip: one probe kernel per row of the table below
(name, body of one unrolled slot; `c` = chain index, u/d/f = register arrays, u[15]/d[15]/f[15] = a second operand)."""
import os

ROWS = [
    ('v_add_u32 (VOP2)',
     'asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_mov_b32 (VOP1)',
     'asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(u[(c + 1) % CH]));'),
    ('v_and_b32 (VOP2)',
     'asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_lshlrev_b32 (VOP2)',
     'asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[c]));'),
    ('v_mul_u32_u24 (VOP2)',
     'asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_max_u32 (VOP2)',
     'asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_add_u32 literal (VOP2 + 32-bit literal)',
     'asm volatile("v_add_u32 %0, 0x6b6b6b6b, %0" : "+v"(u[c]));'),
    ('v_add_u32_e64 (VOP3 encoding of a VOP2 op)',
     'asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_add3_u32 (VOP3)',
     'asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_lshl_add_u32 (VOP3)',
     'asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_bfe_u32 (VOP3)',
     'asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(u[c]));'),
    ('v_mad_i32_i24 (VOP3)',
     'asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_mad_u32_u24 (VOP3)',
     'asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_alignbit_b32 (VOP3)',
     'asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_or3_b32 (VOP3)',
     'asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_bitop3_b32 (VOP3)',
     'asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0xc8" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_mul_lo_u32 (VOP3)',
     'asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_sad_u32 (VOP3)',
     'asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_perm_b32 (VOP3)',
     'asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_lshlrev_b64 (VOP3)',
     'asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[c]));'),
    ('v_lshl_add_u64 (VOP3)',
     'asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_cndmask_b32 vcc (VOP2)',
     'asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_cndmask_b32_e64 s[mask] (VOP3)',
     'asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[15]), "s"(smask));'),
    ('v_cndmask vcc : v_add_u32 = 1 : 1',
     'if (k & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_cndmask vcc : v_add_u32 = 1 : 3',
     'if ((k & 3) == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_cndmask vcc : v_add_f64 = 1 : 3',
     'if ((k & 3) == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_cmp_gt_u32 vcc + v_cndmask vcc (pair = 2 instr)',
     'if (k & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[c]), "v"(u[15]) : "vcc");'),
    ('v_cmp_gt_u32_e64 s[] + v_cndmask_e64 s[] (pair)',
     'if (k & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[15]), "s"(smask)); else asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(smask) : "v"(u[c]), "v"(u[15]));'),
    ('v_cmp_gt_u32 vcc (VOPC)',
     'asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[c]), "v"(u[15]) : "vcc");'),
    ('v_cmp_gt_u32_e64 s[] (VOP3)',
     'asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(smask) : "v"(u[c]), "v"(u[15]));'),
    ('v_cmp_gt_f64 vcc (VOPC)',
     'asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[c]), "v"(d[15]) : "vcc");'),
    ('v_mov_b32 dpp row_shr:1',
     'asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]) : "v"(u[(c + 1) % CH]));'),
    ('v_add_u32 dpp row_shr:1',
     'asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[c]) : "v"(u[(c + 1) % CH]));'),
    ('v_add_u32 sdwa byte select',
     'asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_readlane_b32 (to SGPR)',
     'asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(u[c]));'),
    ('v_readfirstlane_b32',
     'asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(u[c]));'),
    ('v_add_f32 (VOP2)',
     'asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[15]));'),
    ('v_fma_f32 (VOP3)',
     'asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(f[15]));'),
    ('v_add_f64',
     'asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_mul_f64',
     'asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_fma_f64',
     'asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_max_f64',
     'asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_add_f64 |abs| modifier',
     'asm volatile("v_add_f64 %0, |%0|, -%1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_cvt_f64_u32',
     'asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[c]) : "v"(u[c]));'),
    ('v_rcp_f64 (transcendental)',
     'asm volatile("v_rcp_f64 %0, %0" : "+v"(d[c]));'),
    ('v_add_f64 : v_add_u32 = 1 : 3',
     'if ((k & 3) == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_add_f64 : s_add_u32 = 1 : 1',
     'if (k & 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15])); else asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s3) : "scc");'),
    ('v_add_u32 : s_add_u32 = 1 : 1',
     'if (k & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15])); else asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s3) : "scc");'),
    ('s_add_u32',
     'asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s3) : "scc");'),
    ('s_nop 0',
     'asm volatile("s_nop 0");'),
    ('ds_read_b64 conflict-free, waitcnt per 16',
     'asm volatile("ds_read_b64 %0, %1" : "=v"(d[c]) : "v"(addr)); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('ds_read_b32 conflict-free, waitcnt per 16',
     'asm volatile("ds_read_b32 %0, %1" : "=v"(u[c]) : "v"(addr)); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('ds_read_b128 conflict-free, waitcnt per 16',
     'asm volatile("ds_read_b128 %0, %1" : "=v"(q4) : "v"(addr16)); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('ds_add_u32 lane-private, waitcnt per 16',
     'asm volatile("ds_add_u32 %0, %1" : : "v"(addr4), "v"(u[15]) : "memory"); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('ds_write_b64 conflict-free, waitcnt per 16',
     'asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(d[15]) : "memory"); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('ds_bpermute_b32, waitcnt per 16',
     'asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[c]) : "v"(addr4), "v"(u[15])); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('ds_read_b64 : v_add_f64 = 1 : 3, waitcnt per 16',
     'if ((k & 3) == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(d[8 + (c & 3)]) : "v"(addr)); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c & 7]) : "v"(d[15])); if (k % 16 == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");'),
    ('2 x v_cndmask vcc + 2 x v_add_u32 (per 4)',
     'if ((k & 3) < 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('2 x v_cndmask vcc + 6 x v_add_u32 (per 8)',
     'if ((k & 7) < 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('2 x v_cndmask vcc + 6 x v_add_f64 (per 8)',
     'if ((k & 7) < 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('4 x v_cndmask vcc + 4 x v_add_u32 (per 8)',
     'if ((k & 7) < 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('2 x v_cndmask_e64 s[] + 2 x v_add_u32 (per 4)',
     'if ((k & 3) < 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[c]) : "v"(u[15]), "s"(smask)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[15]));'),
    ('v_cmp_gt_f64 vcc + 2 x v_cndmask vcc + v_add_f64 (per 4)',
     'if ((k & 3) == 0) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[c]), "v"(d[15]) : "vcc"); else if ((k & 3) < 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_max_f64 + v_add_f64 (per 2)',
     'if (k & 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15])); else asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[15]));'),
    ('v_cndmask vcc, different vcc writer every 8 (s_mov vcc)',
     'if ((k & 7) == 0) asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc"); else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[15]));'),
]

HEAD = '// tools/issue_probe.hip - issue cost of the instruction classes this path is made of, on gfx950 (VERDICT r02 weak #6:\n// "4 cycles for every VALU and LDS wave-instruction" was an assumption; the guide\'s table says 2 for 32-bit VALU with\n// more than one wave per SIMD). GENERATED by tools/make_issue_probe.py from its table of instruction templates.\n//   hipcc -O3 --offload-arch=gfx950 -o build/issue_probe tools/issue_probe.hip && build/issue_probe\n// Four-wave blocks (one wave per SIMD of a CU, by construction), W blocks per CU = W waves per SIMD (round 3 used one-wave\n// blocks and trusted their placement: a SIMD that got one wave more than its share set the launch time). Every wave issues ITER x 64 instructions of one class from CH independent dependency chains. Output: SIMD\n// cycles per wave-instruction = launch cycles / (wave-instructions per SIMD), at the clock hipDeviceProp reports - the\n// same accounting bench.py\'s `roofline.issue` uses (wave-instructions x cost / 1024 SIMDs / clock).\n#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n\n#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)\ntypedef unsigned u4v __attribute__((ext_vector_type(4)));\n\n#define PROBE_KERNEL(ID, BODY)                                                                            \\\n    template <int CH>                                                                                     \\\n    __global__ __launch_bounds__(256) void probe_##ID(unsigned *out, int iters) {                          \\\n        __shared__ __attribute__((aligned(16))) double lds[2048];                                         \\\n        const int lane = threadIdx.x;                                                                     \\\n        for (int i = lane; i < 2048; i += 256) lds[i] = (double)i;                                         \\\n        __syncthreads();                                                                                  \\\n        unsigned u[16];                                                                                   \\\n        double d[16];                                                                                     \\\n        float f[16];                                                                                      \\\n        _Pragma("unroll") for (int i = 0; i < 16; ++i) u[i] = lane * 7 + i, d[i] = (double)(lane + i) * 1e-3, f[i] = (float)(lane + i); \\\n        unsigned s0 = blockIdx.x, s3 = 7;                                                                 \\\n        unsigned long long smask = 0x5555555555555555ull + blockIdx.x;                                    \\\n        const unsigned addr = lane * 8, addr4 = lane * 4, addr16 = lane * 16;                             \\\n        u4v q4 = {0, 0, 0, 0};                                                                            \\\n        asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc");                                         \\\n        for (int it = 0; it < iters; ++it) {                                                              \\\n            _Pragma("unroll") for (int k = 0; k < 64; ++k) {                                              \\\n                const int c = k % CH;                                                                     \\\n                (void)c;                                                                                  \\\n                BODY                                                                                      \\\n            }                                                                                             \\\n        }                                                                                                 \\\n        unsigned acc = s0 + s3 + (unsigned)smask + q4.x + q4.y + q4.z + q4.w;                             \\\n        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc += u[i] + (unsigned)d[i] + (unsigned)f[i];     \\\n        out[blockIdx.x * 256 + lane] = acc + (unsigned)lds[lane];                                          \\\n    }\n'

MID = r"""
static double g_ghz;
static int g_cus;
static hipStream_t g_s2;
static unsigned long long *g_clk;

// The shader clock DURING a probe (VERDICT r03 #4: the probe's "cycles" were wall time x the nominal 2.4 GHz, and they moved
// with the waves per SIMD because the clock does): a lone wave on a second stream sleeps in s_sleep 127 steps (64 x 127
// shader cycles each, 8,128 .. 8,192: priced at 8,160) for the duration of the timed launches and counts its steps against
// the constant 100 MHz counter - as plaac_clock_probe does beside the scoring kernels.
__global__ void clock_wave(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long steps = 0, now = t0;
    do {
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(127);
        steps += 8;
        now = wall_clock64();
    } while (now - t0 < ticks);
    if (threadIdx.x == 0) {
        out[0] = now - t0;
        out[1] = steps;
    }
}

template <class K>
static void run(const char *name, K kern, int ch, int wps, unsigned *d_out) {
    const int ITER = 2000, REPS = 3;
    const int grid = g_cus * wps; // 256-thread blocks: one wave on each SIMD of a CU
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, 100);
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float warm = 0;
    CK(hipEventElapsedTime(&warm, a, b));
    // the clock wave covers the timed launches (estimated from the warm-up launch of 100 iterations)
    const double est_us = (double)warm * 1e3 * (ITER / 100.0) * REPS * 1.05 + 50.0;
    hipLaunchKernelGGL(clock_wave, dim3(1), dim3(64), 0, g_s2, (unsigned long long)(est_us * 100.0), g_clk);
    float best = 1e30f;
    for (int rep = 0; rep < REPS; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_out, ITER);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    CK(hipStreamSynchronize(g_s2));
    unsigned long long h[2] = {1, 0};
    CK(hipMemcpy(h, g_clk, sizeof h, hipMemcpyDeviceToHost));
    const double mhz = (double)h[1] * 8160.0 / ((double)h[0] / 100.0);
    const double per_simd = (double)wps * ITER * 64.0; // wave-instructions per SIMD
    printf("%-52s chains %2d  waves/SIMD %d  %8.3f ms  clock %6.0f MHz  %6.2f cycles (%5.2f at the nominal clock) per wave-instruction per SIMD\n",
           name, ch, wps, best, mhz, best * 1e-3 * mhz * 1e6 / per_simd, best * 1e-3 * g_ghz * 1e9 / per_simd);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

#define SWEEP(ID, NAME)                                                  \
    for (int wps = 1; wps <= 4; ++wps) run(NAME, probe_##ID<8>, 8, wps, d_out); \
    run(NAME, probe_##ID<1>, 1, 1, d_out);                               \
    run(NAME, probe_##ID<1>, 1, 4, d_out);

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    g_ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, nominal clock %.3f GHz; cycles = launch time x the shader clock MEASURED during the launch / wave-instructions per SIMD\n", p.gcnArchName, g_cus, g_ghz);
    unsigned *d_out;
    CK(hipMalloc(&d_out, sizeof(unsigned) * 64 * g_cus * 16));
    CK(hipStreamCreateWithFlags(&g_s2, hipStreamNonBlocking));
    CK(hipMalloc(&g_clk, 2 * sizeof(unsigned long long)));
"""



def main():
    out = [HEAD]
    for i, (name, body) in enumerate(ROWS):
        out.append("PROBE_KERNEL(%d, %s)\n" % (i, body))
    out.append(MID)
    for i, (name, body) in enumerate(ROWS):
        out.append('    if (argc < 2 || strstr("%s", argv[1])) { SWEEP(%d, "%s") }\n' % (name, i, name))
    out.append("    return 0;\n}\n")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "issue_probe.hip"), "w") as fh:
        fh.write("".join(out))


if __name__ == "__main__":
    main()
