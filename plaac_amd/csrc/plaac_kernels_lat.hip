// plaac_kernels_lat.hip — the LATENCY-FORM chain kernels once more, as a translation unit of their own that the Makefile
// builds with the compiler's "max-ilp" instruction scheduling (-mllvm -amdgpu-sched-strategy=max-ilp). Round 5, late
// (profiles/r05_ab_sched_max_ilp.txt): the whole library under that strategy runs the throughput-bound headline 3.7 % SLOWER
// and the chain-bound configuration 3 (one wave's dependent chain) 6 % FASTER - so the strategy goes to the kernels whose
// time IS one wave's chain, and nowhere else. Same sources (kernels_*.hip.inc, included here a second time in this unit's own
// anonymous namespace), same arithmetic, same results: the parity tests run through these copies by default
// (PLAAC_LAT_UNIT=0: the copies of the main unit). Summary mode only: the track-mode chains (k_bwd_pair, the window halves,
// k_fwd_pair<TRACKS>) and the sweeps' k_vit<NC, LAT> measured the same or slower from this unit and stay in the main one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "plaac_native.h"

namespace {
#include "kernels_tables_plan.hip.inc"
#include "kernels_chains.hip.inc"
} // namespace

namespace plaac_lat {

// (DevTables / SweepTargets are this unit's own types of the same layout: the main unit passes pointers / bytes)
void launch_long(dim3 grid, hipStream_t s, const uint64_t *offsets, const uint32_t *neff, const void *order, uint32_t nprot,
                 const void *tab, const void *packed, const uint32_t *grow, const void *tg_bytes, double *lmarg, double *h0) {
    SweepTargets tg;
    std::memcpy(&tg, tg_bytes, sizeof tg);
    hipLaunchKernelGGL(k_long, grid, dim3(KA_THREADS), 0, s, offsets, neff, (const uint4 *)order, nprot, (const DevTables *)tab,
                       (const uint4 *)packed, grow, tg, lmarg, h0);
}

void launch_fwd_pair(dim3 grid, hipStream_t s, const void *order, uint32_t nprot, const void *tab, const void *packed,
                     const uint32_t *grow, double *lmarg) {
    hipLaunchKernelGGL(k_fwd_pair<false>, grid, dim3(KA_THREADS), 0, s, (const uint4 *)order, nprot, (const DevTables *)tab,
                       (const uint4 *)packed, grow, lmarg, (double2 *)nullptr);
}

// k_vit<1, LAT = true, EXT = true, LIST = false>: the Viterbi pass of a chain-bound single-point call
void launch_vit_lat_ext(dim3 grid, hipStream_t s, const uint8_t *codes, const uint64_t *offsets, const uint32_t *neff, const void *order,
                        uint32_t nprot, const void *tab, const void *packed, const uint32_t *grow, uint32_t *bits, const void *tg_bytes,
                        double *vend) {
    SweepTargets tg;
    std::memcpy(&tg, tg_bytes, sizeof tg);
    hipLaunchKernelGGL((k_vit<1, true, true, false>), grid, dim3(KA_THREADS), 0, s, codes, offsets, neff, (const uint4 *)order, nprot,
                       (const DevTables *)tab, (const uint4 *)packed, grow, bits, tg, (uint32_t *)nullptr, (uint32_t *)nullptr, vend);
}

size_t sizeof_sweep_targets() { return sizeof(SweepTargets); }
size_t sizeof_dev_tables() { return sizeof(DevTables); }

} // namespace plaac_lat
