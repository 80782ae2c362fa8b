// plaac_kernels.hip — gfx950 (CDNA4) kernels + the device half of the C ABI (include/plaac_native.h).
//
// MUST be compiled with -ffp-contract=off: the reference (Java) never fuses a*b+c and the
// tie-breaks of the window searches depend on the exact rounding of every add.
//
// Kernel map (reference = cli/src/plaac.java):
//   k_plan_*      length histogram / scan / scatter: descending-length order, so that the 64 lanes
//                 of a wave run recurrences of similar length and the longest chains start first.
//   k_vit/k_fwd/k_win  "K-A", ONE LANE PER PROTEIN. All order-sensitive serial fp64 chains:
//                   Viterbi + traceback            hmm.viterbidecodel   :3077-3121
//                   forward with the LUT log-sum-exp  hmm.posteriorl    :3354-3375, logeapeb :1024-1047
//                   hmm0 (degenerates to a running sum)                  :795, SURVEY H4
//                   MW and LLR fixed-width windows over prefix sums  hss2 :1206-1257 via :767-783
//                   masked core window, PRD expansion, PRD score          :816-880
//                   mean hydropathy / charge / FoldIndex                  :4877-4885
//                 A serial chain cannot be re-associated (bit-exactness), so the parallelism is
//                 ACROSS proteins: 64 independent chains per wave, tables + 32 KB loglut in LDS.
//   k_group_rows / k_scan_u32 / k_pack  group-interleaved copy of the residues for the K-A kernels (coalesced rows).
//   k_bwd / k_post  (track mode) backward recurrence as its own chain; posteriors, MAP and path bytes per packed row.
//   k_tracks*     "K-B", position-parallel. The window tracks of disorderreport (:4866-5068): hydro/charge/
//                 FoldIndex, PLAAC-LLR and PAPA sliding means (fixed-order 41-term sums per position), their
//                 weighted second smoothing, the PAPA arg-max and the FoldIndex run statistics, as three
//                 pipelined stages over LDS rings with wave scans / shuffles for the reductions.
//                   k_tracks20s   ww = 41/40 (default): ONE WAVE PER 32 PROTEINS laid end to end on one position axis
//                   k_tracks20    same windows, one wave per protein (PLAAC_KB_PER_PROTEIN=1; very large batches)
//                   k_tracks<R>   any other window size, one wave per protein
//   k_llr_at_centre  sweeps: the two llr-dependent K-B outputs at the known PAPA centre, for the 2nd, 3rd ... alpha.
//   k_hist / k_validate  22-bin histogram over valid records (countaas/isvalidprotein :1698-1739); code range check.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "plaac_native.h"

// ------------------------------------------------------------------------------------------------
// device tables
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int NAA = PLAAC_NAA;
constexpr int LUTLEN = PLAAC_LUTLEN;
constexpr int LEN_BINS = 1 << 16; // lengths >= LEN_BINS-1 share the last bin

// per-code row used by the K-A kernels: one LDS row per residue code
enum { R_LE0 = 0, R_LE1 = 1, R_LLR = 2, R_HYD = 3, R_LE0H = 4, R_PAD = 5, R_W = 6 };

struct DevTables {
    double row[NAA][R_W]; // {hmm1.le[0], hmm1.le[1], llr, hydro2, hmm0.le[0], -}
    double lod[NAA];      // papa log-odds
    double hyd[NAA];      // hydro2
    double llr[NAA];
    int32_t chg[NAA];     // aacharge as integers (-1, 0, 1)
    int32_t pad0_[2];
    double lt[2][2], li[2], lf[2]; // hmm1
    double h0_lt00, h0_li0, h0_lf0; // hmm0 (identity transitions: only state 0 is reachable)
    double cc[3];
    double big_neg;
    int32_t corelength, ww1, ww2, ww3, adjustprolines, pad1_;
    double loglut[LUTLEN + 1]; // [LUTLEN] = 0.0: lse_lut clamps out-of-range differences onto (LUTLEN-1, LUTLEN)
};

struct TrackPtrs {
    uint8_t *vit, *map;
    double *charge, *hydro, *fi, *plaacllr, *papa, *fix2, *plaacllrx2, *papax2, *post0, *post1;
};

// ------------------------------------------------------------------------------------------------
// planning kernels: effective lengths, descending-length counting sort
// ------------------------------------------------------------------------------------------------
// Both counting-sort passes privatise the popular bins (lengths < PLAN_LDS_BINS) in LDS: one global atomic
// per touched bin per block instead of one per protein (the length histogram is very peaked).
constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_ITEMS = 8;          // proteins per thread
constexpr int PLAN_LDS_BINS = 8192;

__device__ __forceinline__ uint32_t plan_bin(uint32_t len) {
    return len < (uint32_t)(LEN_BINS - 1) ? len : (uint32_t)(LEN_BINS - 1);
}

__global__ __launch_bounds__(PLAN_THREADS) void k_plan_lengths(const uint8_t *__restrict__ codes,
                                                               const uint64_t *__restrict__ offsets, uint32_t nprot,
                                                               uint32_t *__restrict__ neff,
                                                               uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[PLAN_LDS_BINS];
    for (int i = threadIdx.x; i < PLAN_LDS_BINS; i += PLAN_THREADS) h[i] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (PLAN_THREADS * PLAN_ITEMS);
    for (int k = 0; k < PLAN_ITEMS; ++k) {
        const uint32_t p = base + (uint32_t)k * PLAN_THREADS + threadIdx.x;
        if (p >= nprot) break;
        const uint64_t b = offsets[p], e = offsets[p + 1];
        uint64_t len = e > b ? e - b : 0;
        if (len > 0 && codes[e - 1] == 21) --len; // one trailing stop is dropped before scoring (:758)
        if (len > 0x7fffffffu) len = 0x7fffffffu;
        neff[p] = (uint32_t)len;
        const uint32_t bin = plan_bin((uint32_t)len);
        if (bin < (uint32_t)PLAN_LDS_BINS) atomicAdd(&h[bin], 1u);
        else atomicAdd(&hist[bin], 1u);
        // hist[LEN_BINS] = "the batch holds a protein of >= LEN_BINS-1 residues": the stream form of the window kernel
        // (32 proteins on one int32 position axis) then leaves the batch to the one-protein-at-a-time form
        if (bin == (uint32_t)(LEN_BINS - 1)) hist[LEN_BINS] = 1u;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PLAN_LDS_BINS; i += PLAN_THREADS)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// single block: cursor[b] = number of proteins in bins > b (descending order start positions)
__global__ __launch_bounds__(1024) void k_plan_scan(uint32_t *__restrict__ hist) {
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    constexpr int PER = LEN_BINS / 1024;
    // thread tid owns bins [hi - PER + 1, hi] with hi = LEN_BINS-1 - tid*PER (descending)
    const int hi = LEN_BINS - 1 - tid * PER;
    uint32_t s = 0;
    for (int j = 0; j < PER; ++j) s += hist[hi - j];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
        uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s; // exclusive prefix for this thread's first (largest) bin
    for (int j = 0; j < PER; ++j) {
        uint32_t c = hist[hi - j];
        hist[hi - j] = run;
        run += c;
    }
}

__global__ __launch_bounds__(PLAN_THREADS) void k_plan_scatter(const uint32_t *__restrict__ neff, uint32_t nprot,
                                                               uint32_t *__restrict__ cursor,
                                                               const uint64_t *__restrict__ offsets,
                                                               uint4 *__restrict__ order) {
    __shared__ uint32_t cnt[PLAN_LDS_BINS]; // per-bin count of this block, then running rank
    __shared__ uint32_t start[PLAN_LDS_BINS];
    for (int i = threadIdx.x; i < PLAN_LDS_BINS; i += PLAN_THREADS) cnt[i] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (PLAN_THREADS * PLAN_ITEMS);
    uint32_t bins[PLAN_ITEMS], lens[PLAN_ITEMS];
    for (int k = 0; k < PLAN_ITEMS; ++k) {
        const uint32_t p = base + (uint32_t)k * PLAN_THREADS + threadIdx.x;
        lens[k] = p < nprot ? neff[p] : 0u;
        bins[k] = p < nprot ? plan_bin(lens[k]) : 0xffffffffu;
        if (bins[k] < (uint32_t)PLAN_LDS_BINS) atomicAdd(&cnt[bins[k]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PLAN_LDS_BINS; i += PLAN_THREADS) { // reserve this block's slots of every bin
        const uint32_t c = cnt[i];
        start[i] = c ? atomicAdd(&cursor[i], c) : 0u;
        cnt[i] = 0u;
    }
    __syncthreads();
    for (int k = 0; k < PLAN_ITEMS; ++k) {
        const uint32_t p = base + (uint32_t)k * PLAN_THREADS + threadIdx.x;
        if (p >= nprot) break;
        const uint32_t bin = bins[k];
        const uint32_t pos = bin < (uint32_t)PLAN_LDS_BINS ? start[bin] + atomicAdd(&cnt[bin], 1u)
                                                           : atomicAdd(&cursor[bin], 1u);
        // the sorted plan carries everything the scoring kernels need about a protein, so that they read it
        // coalesced / sequentially instead of gathering neff[p] and offsets[p] (a 128-byte line per 4-8 bytes)
        const uint64_t off = offsets[p];
        order[pos] = make_uint4((uint32_t)off, (uint32_t)(off >> 32), lens[k], p);
    }
}

// ------------------------------------------------------------------------------------------------
// K-A: one lane per protein — the serial recurrences, split into three independent roles so that
// the chains of one (long) protein advance concurrently in different waves:
//   k_vit   Viterbi + traceback + masked core / PRD                    (3 dependent sweeps)
//   k_fwd   forward recurrence with the LUT log-sum-exp (+ backward / posteriors in track mode)
//   k_win   MW and LLR prefix-sum windows, mean hydropathy / charge
// The kernel time is set by the longest protein's chain (n x per-step issue time of ONE wave), so the
// per-step instruction count of each role and the wave's issue priority matter, not the lane count.
//
// Residues are consumed in blocks of 16 steps, phase-locked across the 64 lanes (block boundaries are
// relative to each protein's start, hence the unaligned 16-byte loads): every lane reloads in the same
// iteration, the next block is already in flight while the current one is consumed (memory latency stays
// off the serial chain), and inside the fully unrolled block every byte / bit extraction has an
// immediate shift. (A per-lane aligned stream would reload in a different iteration per lane, and the
// wave-wide vmcnt wait would then expose the full memory latency on every step.)
// ------------------------------------------------------------------------------------------------
// logeapeb (:1024-1047), branch-free. For the values that occur (no NaN) max/min select exactly the
// operands the reference's if/else picks. a==b (incl. -inf,-inf) falls out of the same formula:
// dex=0 gives a + (0*lut[1] + 1*lut[0]) = a + ln 2, and -inf - -inf = NaN fails (c<40) -> hi.
__device__ __forceinline__ double lse_lut(const double *__restrict__ lut, double a, double b) {
    const double hi = __builtin_fmax(a, b);
    const double lo = __builtin_fmin(a, b);
    const double c = hi - lo;
    const bool inrange = c < 40.0;
    // The dependent chain of this function bounds the forward kernel for a long protein, so it is kept short: out-of-range
    // differences (c >= 40, +inf, NaN) are clamped onto the padded last table entry by one min (their result is
    // discarded below) instead of a select in front of the multiply; the integer part comes from a truncating
    // conversion of x itself (x >= 0: trunc == floor) and the fraction from v_fract, side by side, instead of
    // floor -> convert and floor -> subtract. x - floor(x) is exact, so fract(x) has the same bits.
    const double x = __builtin_fmin(100.0 * c, (double)(LUTLEN - 1));
    const int dex = (int)x;
    // (dex + 1 - 100c) of the reference: 1 - (x - floor(x)) is the same real number rounded once
    const double fr = __builtin_amdgcn_fract(x);
    const double r = hi + (fr * lut[dex + 1] + (1.0 - fr) * lut[dex]);
    return inrange ? r : hi;
}

__device__ __forceinline__ uint32_t ld_code(const uint8_t *__restrict__ x, uint32_t t) {
    uint32_t c = x[t];
    return c > 21u ? 0u : c;
}

constexpr int KA_THREADS = 256;
constexpr int ROWS = NAA + 1; // LDS row 22 duplicates row 0: any byte > 21 is scored as X with one v_min

// 16 residues starting at p (any alignment). [lo, end) is the codes buffer; a block that is not wholly
// inside it (only the first / last few residues of the whole batch) is assembled byte by byte.
__device__ __noinline__ uint4 load16_edge(const uint8_t *p, const uint8_t *lo, const uint8_t *end) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int i = 0; i < 16; ++i) {
        const uint8_t *q = p + i;
        const uint32_t b = (q >= lo && q < end) ? (uint32_t)*q : 0u;
        w[i >> 2] |= b << (8 * (i & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 load16(const uint8_t *p, const uint8_t *lo, const uint8_t *end) {
    if (p >= lo && p + 16 <= end) {
        uint4 v;
        __builtin_memcpy(&v, p, 16); // one (possibly unaligned) global_load_dwordx4
        return v;
    }
    return load16_edge(p, lo, end);
}
// ---- group-interleaved copy of the residues for the lane-per-protein kernels ------------------------------
// Wave-group g = the 64 proteins order[64g .. 64g+63] (similar lengths after the sort). Its residues are
// re-packed as rows of 64 x 16 bytes: row j holds residues 16j..16j+15 of every lane, lane-major, so a wave's
// block load is ONE fully coalesced 1 KiB read (instead of 64 scattered 16-byte reads that each drag a
// whole sector through L2, measured 8-20x over-fetch), and every residue crosses HBM once per sweep.
// The traceback / Viterbi-path bit words use the same row numbering (one 32-bit word per lane per row).
// rows[g] = ceil(longest length in group / 16); grow[] = exclusive prefix sum (grow[ngroups] = total rows).
__global__ void k_group_rows(const uint32_t *__restrict__ neff, const uint4 *__restrict__ order, uint32_t nprot,
                             uint32_t ngroups, uint32_t *__restrict__ grow) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    // Lane 0 is the longest of its group, EXCEPT inside the last length bin (lengths >= LEN_BINS-1 are not
    // ordered among themselves): take the maximum of the group when its first member is that long.
    uint32_t n0 = order[64u * g].z;
    if (n0 >= (uint32_t)(LEN_BINS - 1)) {
        const uint32_t e = 64u * g + 64u < nprot ? 64u * g + 64u : nprot;
        for (uint32_t i = 64u * g + 1u; i < e; ++i) {
            const uint32_t v = order[i].z;
            n0 = v > n0 ? v : n0;
        }
    }
    grow[g] = (n0 + 15u) >> 4;
}

// single block: in-place exclusive scan of a[0..n), total written to a[n]
// (256 threads: a 1024-thread block needs 4 waves per SIMD at once and can wait milliseconds for room beside the window
// kernel)
__global__ __launch_bounds__(256) void k_scan_u32(uint32_t *__restrict__ a, uint32_t n) {
    __shared__ uint32_t part[256];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 255u) / 256u;
    const uint32_t b = tid * per, e = b + per < n ? b + per : n;
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += a[i];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t c = a[i];
        a[i] = run;
        run += c;
    }
    if (tid == 255u) a[n] = part[255];
}

// 16 lanes per protein: a protein's residues are read as contiguous 256-byte pieces (coalesced; one lane per
// protein reading 16 bytes at a time dragged every 128-byte line through L2 2.5 times), the 16-byte writes of the 64
// proteins of a group land in the same 1 KiB rows at about the same time and merge in L2.
__global__ __launch_bounds__(256) void k_pack(const uint8_t *__restrict__ codes, const uint64_t *__restrict__ offsets,
                                              const uint32_t *__restrict__ neff, const uint4 *__restrict__ order,
                                              uint32_t nprot, uint64_t total, const uint32_t *__restrict__ grow,
                                              uint4 *__restrict__ packed) {
    __builtin_amdgcn_s_setprio(3); // on the critical path of the serial chains, beside the window kernel
    const uint32_t gid = blockIdx.x * 16u + (threadIdx.x >> 4); // 16 proteins per block
    const uint32_t sub = threadIdx.x & 15u;
    if (gid >= nprot) return;
    const uint4 it = order[gid];
    const uint32_t n = it.z;
    const uint8_t *x = codes + (((uint64_t)it.y << 32) | it.x);
    const uint8_t *cend = codes + total;
    uint4 *__restrict__ col = packed + (size_t)grow[gid >> 6] * 64u + (gid & 63u);
    const uint32_t nj = (n + 15u) >> 4;
    uint32_t j = sub;
    for (; j + 16u < nj; j += 32u) { // two independent loads in flight per lane
        const uint4 v0 = load16(x + 16u * j, codes, cend), v1 = load16(x + 16u * j + 256u, codes, cend);
        col[(size_t)j * 64u] = v0;
        col[(size_t)(j + 16u) * 64u] = v1;
    }
    if (j < nj) col[(size_t)j * 64u] = load16(x + 16u * j, codes, cend);
}

// one lane's view of its group's packed rows
struct PackedLane {
    const uint4 *__restrict__ pk; // residues: row j at pk[64*j]
    uint32_t *__restrict__ wb;    // bit words: row j at wb[64*j]
    __device__ __forceinline__ uint4 chunk(uint32_t j) const { return pk[(size_t)j * 64u]; }
    __device__ __forceinline__ uint32_t word(uint32_t j) const { return wb[(size_t)j * 64u]; }
    __device__ __forceinline__ void set_word(uint32_t j, uint32_t v) const { wb[(size_t)j * 64u] = v; }
    // the 16 residues at positions s .. s+15 (any alignment, s >= -15; positions < 0 read as 0). The byte
    // shift s & 15 is wave-uniform and loop-invariant for the trailing streams (s = t0 - c).
    __device__ __forceinline__ uint4 window(int s) const {
        const int js = s >> 4;
        const uint32_t sh = (uint32_t)s & 15u;
        const uint4 lo = js >= 0 ? pk[(size_t)js * 64u] : make_uint4(0u, 0u, 0u, 0u);
        if (sh == 0u) return lo;
        const uint4 hi = pk[(size_t)(js + 1) * 64u];
        const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const uint32_t bs = sh & 3u;
        uint32_t o[4];
        switch (sh >> 2) { // wave-uniform
        case 0:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], bs);
            break;
        case 1:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbyte(d[k + 2], d[k + 1], bs);
            break;
        case 2:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbyte(d[k + 3], d[k + 2], bs);
            break;
        default:
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_alignbyte(d[k + 4], d[k + 3], bs);
            break;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};

__device__ __forceinline__ PackedLane packed_lane(const uint4 *packed, uint32_t *bits, const uint32_t *grow) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const size_t base = (size_t)grow[gid >> 6] * 64u + (gid & 63u);
    return PackedLane{packed + base, bits + base};
}

// residue j (compile-time after unrolling) of a 16-residue block
__device__ __forceinline__ uint32_t block_code(const uint4 &c, int j) {
    const uint32_t w = j < 4 ? c.x : j < 8 ? c.y : j < 12 ? c.z : c.w;
    const uint32_t b = (w >> (8 * (j & 3))) & 0xffu;
    return b < 22u ? b : 22u;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
        v = o > v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// issue priority by chain length: the longest chains define the kernel's duration
__device__ __forceinline__ void set_wave_priority(uint32_t n) {
    const uint32_t n0 = __builtin_amdgcn_readfirstlane(n); // lane 0 holds the wave's longest protein
    if (n0 >= 4096u) __builtin_amdgcn_s_setprio(3);
    else if (n0 >= 1024u) __builtin_amdgcn_s_setprio(2);
    else if (n0 >= 256u) __builtin_amdgcn_s_setprio(1);
}

struct LaneJob {
    uint32_t p, n;
    uint64_t off;
};

__device__ __forceinline__ LaneJob lane_job(const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ neff,
                                            const uint4 *__restrict__ order, uint32_t nprot) {
    LaneJob j{0u, 0u, 0ull};
    const uint32_t gid = blockIdx.x * KA_THREADS + threadIdx.x;
    if (gid < nprot) {
        const uint4 it = order[gid];
        j.p = it.w;
        j.n = it.z;
        j.off = ((uint64_t)it.y << 32) | it.x;
    }
    return j;
}

__device__ __forceinline__ void load_rows(double *s_row, const DevTables *__restrict__ T) {
    for (int i = threadIdx.x; i < ROWS * R_W; i += KA_THREADS) {
        const int r = i / R_W, k = i - r * R_W;
        s_row[i] = T->row[r == NAA ? 0 : r][k];
    }
}

// ---- role V: Viterbi (:3077-3121), traceback, longest run (:1787-1804), masked core + PRD (:816-880) ----
// Every sweep has a guard-free straight-line body for blocks that lie wholly inside the lane's protein
// (all but the last block of each lane) so that the LDS lookups of later steps can be hoisted above the
// serial chain, and a guarded body for the last, partial block.
struct VitState {
    double s0, s1, h0;
};

template <bool GUARD, bool PIN = true>
__device__ __forceinline__ uint32_t vit_block(VitState &S, const double *__restrict__ s_row, const uint4 cur,
                                              uint32_t t0, uint32_t n, double lt00, double lt01, double lt10,
                                              double lt11, double h0lt, int jfirst) {
    uint32_t tbw = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < jfirst) continue; // the very first residue (t = 0) is the initialisation, not a step
        if (!GUARD || t0 + (uint32_t)j < n) {
            const double *__restrict__ r = s_row + block_code(cur, j) * R_W;
            const double e0 = r[R_LE0], e1 = r[R_LE1];
            // (:3087-3100): state 0 stays the arg-max on ties (strict >)
            const double v00 = lt00 + S.s0, v10 = lt10 + S.s1, v01 = lt01 + S.s0, v11 = lt11 + S.s1;
            const bool g0 = v10 > v00, g1 = v11 > v01;
            // the larger candidate as a VALUE is the same whichever way a tie is broken: one max instead of a
            // two-register select; the tie rule lives in the traceback bit only
            S.s0 = __builtin_fmax(v00, v10) + e0;
            S.s1 = __builtin_fmax(v01, v11) + e1;
            if (PIN) S.h0 = (h0lt + S.h0) + r[R_LE0H]; // hmm0: Viterbi == forward == running sum; the
                                                      // latency form (PIN = false) leaves it to k_win / k_finish
            tbw |= ((uint32_t)g0 | ((uint32_t)g1 << 1)) << (2 * j);
        }
        // Left alone, the scheduler defers all 32 compares (they are off the critical chain) to the end of the block and
        // keeps their 64 operands alive: 180 VGPRs, so that a wave of this kernel never fits on a SIMD beside three
        // waves of the window kernel. Pinning the traceback word every four steps brings the kernel to 62 VGPRs.
        // (PIN = false, latency mode: the wave with the longest protein has its SIMD to itself, registers are free, and
        // without the barriers the LDS lookups of all 16 steps are issued ahead of the chain)
        if (PIN && (j & 3) == 3) {
            asm volatile("" : "+v"(tbw));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return tbw;
}

template <bool GUARD>
__device__ __forceinline__ uint32_t traceback_block(uint32_t &state, int &cur, int &maxrun, uint32_t word,
                                                    uint32_t t0, uint32_t n) {
    uint32_t vw = 0;
#pragma unroll
    for (int j = 15; j >= 0; --j) {
        if (!GUARD || t0 + (uint32_t)j < n) {
            // here `state` = vit[t]
            vw |= state << j;
            cur = (int)((uint32_t)cur * state + state); // state ? cur + 1 : 0
            maxrun = cur > maxrun ? cur : maxrun;
            const uint32_t wj = word >> (2 * j);         // off the serial chain
            state = (wj >> state) & 1u;                  // tb[vit[t]][t] = vit[t-1]
        }
    }
    return vw;
}

// Parameter sweeps (BASELINE config 5): the Viterbi parse depends on the background mix alpha but not on the core
// length, so ONE pass of sweeps 1-2 serves up to MAXC core lengths; sweep 3 then carries one trailing chain
// and one best-window record per core length. NC = 1 is the ordinary single-parameter call.
constexpr uint32_t CORE_LONG_ROWS = 128; // latency form: wave-groups with >= 2048-residue proteins (k_core_*)
constexpr unsigned CORE_MAX_GROUPS = 2048;
constexpr int MAXC = 4;
struct SweepTargets {
    uint32_t c[MAXC];     // core lengths
    plaac_row *rows[MAXC]; // one row array per core length
    uint32_t stop_after;   // DIAGNOSTIC (PLAAC_VIT_STOP): k_vit returns after this sweep (0: runs all three)
};

template <int NC>
struct CoreState {
    double mL;
    double mT[NC], best[NC];
    int bstart[NC];
};

// STEADY: the whole block has t >= c for every core length, i.e. all chains run and every step closes a window
template <bool GUARD, bool STEADY, int NC>
__device__ __forceinline__ void core_block(CoreState<NC> &S, const double *__restrict__ s_mask, const uint4 cur,
                                           const uint4 (&tcur)[NC], uint32_t wl, const uint32_t (&tv)[NC], uint32_t t0,
                                           uint32_t n, const uint32_t (&c)[NC]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t t = t0 + (uint32_t)j;
        if (!GUARD || t < n) {
            // maa3[i] (:818-823) = llr if the path bit is set, else big_neg: one lookup in a two-plane table
            S.mL = S.mL + s_mask[(((wl >> j) & 1u) << 5) | block_code(cur, j)]; // psum[i+1] = psum[i] + maa3[i]
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (STEADY || t >= c[k]) { // same chain, c steps later
                    S.mT[k] = S.mT[k] + s_mask[(((tv[k] >> j) & 1u) << 5) | block_code(tcur[k], j)];
                }
                if (STEADY || t + 1 >= c[k]) {
                    const bool first = !STEADY && t + 1 == c[k];
                    const double d = first ? S.mL : S.mL - S.mT[k];
                    const bool upd = first || d > S.best[k]; // strict >: the first window wins ties
                    S.best[k] = upd ? d : S.best[k];
                    S.bstart[k] = upd ? (int)(t + 1 - c[k]) : S.bstart[k];
                }
            }
        }
    }
}

template <int NC, bool LAT = false>
__global__ __launch_bounds__(KA_THREADS) void k_vit(const uint8_t *__restrict__ codes,
                                                    const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ neff,
                                                    const uint4 *__restrict__ order, uint32_t nprot,
                                                    const DevTables *__restrict__ T,
                                                    const uint4 *__restrict__ packed,
                                                    const uint32_t *__restrict__ grow, uint32_t *__restrict__ bits,
                                                    SweepTargets tg) {
    __shared__ double s_row[ROWS * R_W];
    __shared__ double s_mask[64]; // [0..31]: big_neg (path bit 0), [32..63]: llr by code (path bit 1)
    load_rows(s_row, T);
    if (threadIdx.x < 64) {
        const int k = threadIdx.x & 31;
        s_mask[threadIdx.x] = threadIdx.x < 32 ? T->big_neg : (k < NAA ? T->llr[k] : T->llr[0]);
    }
    __syncthreads();
    const LaneJob J = lane_job(offsets, neff, order, nprot);
    const uint32_t n = J.n;
    set_wave_priority(n);
    if (blockIdx.x * KA_THREADS + threadIdx.x >= nprot) return;
    if (n == 0) { // skipped record (:762): zero the fields this kernel owns
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            plaac_row *row = tg.rows[k] + J.p;
            row->core_score = row->prd_score = row->hmm_vit = 0.0;
            row->vit_maxrun = row->core_start = row->core_end = row->prd_start = row->prd_end = 0;
        }
        return;
    }
    const uint8_t *__restrict__ x = codes + J.off;
    // packed residues of this lane + its column of bit words (2 bits/residue of traceback, overwritten in
    // place by 1 bit/residue of Viterbi path)
    const PackedLane PL = packed_lane(packed, bits, grow);
    const uint32_t nw = (n + 15u) >> 4;

    const double lt00 = T->lt[0][0], lt01 = T->lt[0][1], lt10 = T->lt[1][0], lt11 = T->lt[1][1];
    const double lf0 = T->lf[0], lf1 = T->lf[1];
    const double h0lt = T->h0_lt00;
    uint32_t c[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) c[k] = tg.c[k];

    // ---------------- sweep 1: t = 0 .. n-1 ----------------
    VitState V;
    {
        uint4 nxt = PL.chunk(0);
        {
            const double *__restrict__ r = s_row + block_code(nxt, 0) * R_W;
            V.s0 = T->li[0] + r[R_LE0];
            V.s1 = T->li[1] + r[R_LE1];
            V.h0 = T->h0_li0 + r[R_LE0H];
        }
        for (uint32_t t0 = 0; t0 < n; t0 += 16u) {
            const uint4 cur = nxt;
            if (t0 + 16u < n) nxt = PL.chunk((t0 >> 4) + 1u);
            uint32_t tbw;
            if (t0 == 0u) tbw = vit_block<true, !LAT>(V, s_row, cur, t0, n, lt00, lt01, lt10, lt11, h0lt, 1);
            else if (t0 + 16u <= n) tbw = vit_block<false, !LAT>(V, s_row, cur, t0, n, lt00, lt01, lt10, lt11, h0lt, 0);
            else tbw = vit_block<true, !LAT>(V, s_row, cur, t0, n, lt00, lt01, lt10, lt11, h0lt, 0);
            PL.set_word(t0 >> 4, tbw);
        }
    }
    // end of Viterbi (:3102-3109)
    const double vend0 = V.s0 + lf0, vend1 = V.s1 + lf1;
    uint32_t state = vend1 > vend0 ? 1u : 0u;
    {
        const double hv = LAT ? (state ? vend1 : vend0) : (state ? vend1 : vend0) - (V.h0 + T->h0_lf0);
#pragma unroll
        for (int k = 0; k < NC; ++k) tg.rows[k][J.p].hmm_vit = hv; // LAT: k_finish subtracts hmm0's total
    }

    if (tg.stop_after == 1u) return;
    // ---------------- sweep 2: traceback t = n-1 .. 0 (:3111-3113), longest run ----------------
    {
        int cur = 0, maxrun = 0;
        // Phase-locked like the forward sweeps: all lanes walk the SAME block index down from the group's last block
        // (a lane joins when the index reaches its own last block), so a word load is one coalesced 256-byte row
        // instead of 64 lines; words are fetched two blocks ahead (a traceback block is only ~150 instructions).
        const uint32_t nwmax = wave_max_u32(nw);
        auto word_at = [&](uint32_t wi) { return wi < nw ? PL.word(wi) : 0u; };
        uint32_t q0 = word_at(nwmax - 1u), q1 = nwmax > 1u ? word_at(nwmax - 2u) : 0u;
        for (uint32_t wi = nwmax; wi-- > 0u;) {
            const uint32_t word = q0;
            q0 = q1;
            if (wi >= 2u) q1 = word_at(wi - 2u);
            if (wi < nw) {
                const uint32_t t0 = wi << 4;
                const uint32_t vw = (t0 + 16u <= n) ? traceback_block<false>(state, cur, maxrun, word, t0, n)
                                                    : traceback_block<true>(state, cur, maxrun, word, t0, n);
                PL.set_word(wi, vw); // this word now holds vit[16*wi .. 16*wi+15]
            }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) tg.rows[k][J.p].vit_maxrun = maxrun;
    }

    if (tg.stop_after == 2u) return;
    // latency form: the core window of the wave-groups with very long proteins is found by k_core_chain / _eval /
    // _reduce (wave-uniform: nw of lane 0 ... the group's row count is the maximum over its lanes)
    if (LAT && wave_max_u32(nw) >= CORE_LONG_ROWS && ((blockIdx.x * KA_THREADS + threadIdx.x) >> 6) < CORE_MAX_GROUPS)
        return;
    // ---------------- sweep 3: masked core window(s) (:818-833) ----------------
    CoreState<NC> C;
    C.mL = 0.0;
    uint32_t cmax = 0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        C.mT[k] = 0.0;
        C.best[k] = -INFINITY;
        C.bstart[k] = -1;
        cmax = c[k] > cmax ? c[k] : cmax;
    }
    const double big_neg = T->big_neg;
    {
        // lead stream: residues + path bits of block t0; trailing stream k: the same, c[k] steps later, i.e. the
        // 16 positions s .. s+15 with s = t0 - c[k]. All are fetched one block ahead.
        uint4 nxt = PL.chunk(0), tnxt[NC];
        uint32_t wlnext = PL.word(0), plo[NC], phi[NC];
        auto prefetch_trail = [&](int k, int s) { // block starting at position s (>= -15) of this protein
            tnxt[k] = PL.window(s);
            const uint32_t w0 = s >= 0 ? (uint32_t)s >> 4 : 0u;
            plo[k] = PL.word(w0);
            phi[k] = (s >= 0 && w0 + 1u < nw) ? PL.word(w0 + 1u) : 0u;
        };
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            tnxt[k] = make_uint4(0u, 0u, 0u, 0u);
            plo[k] = phi[k] = 0u;
            if (15 >= (int)c[k]) prefetch_trail(k, -(int)c[k]);
        }
        for (uint32_t t0 = 0; t0 < n; t0 += 16u) {
            const uint4 cur = nxt;
            uint4 tcur[NC];
            uint32_t tv[NC];
            const uint32_t wl = wlnext;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                tcur[k] = tnxt[k];
                const int s = (int)t0 - (int)c[k];
                tv[k] = s >= 0 ? ((plo[k] | (phi[k] << 16)) >> (s & 15)) : (s > -16 ? (plo[k] << (-s)) : 0u);
            }
            if (t0 + 16u < n) {
                nxt = PL.chunk((t0 >> 4) + 1u);
                wlnext = PL.word((t0 >> 4) + 1u);
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int s = (int)t0 - (int)c[k];
                    if (s + 31 >= 0) prefetch_trail(k, s + 16);
                }
            }
            const bool full = t0 + 16u <= n;
            if (t0 >= cmax) { // wave-uniform
                if (full) core_block<false, true, NC>(C, s_mask, cur, tcur, wl, tv, t0, n, c);
                else core_block<true, true, NC>(C, s_mask, cur, tcur, wl, tv, t0, n, c);
            } else {
                core_block<true, false, NC>(C, s_mask, cur, tcur, wl, tv, t0, n, c);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        plaac_row *row = tg.rows[k] + J.p;
        if (C.best[k] > big_neg / 2) { // :861 — a core exists; expand it to the whole Viterbi run (:863-866)
            auto bit = [&](int q) { return (PL.word((uint32_t)q >> 4) >> (q & 15)) & 1u; };
            int a = C.bstart[k], z = C.bstart[k] + (int)c[k] - 1;
            while (a > 0 && bit(a - 1)) --a;
            while (z + 1 < (int)n && bit(z + 1)) ++z;
            double prd = 0.0; // PRDscore: left-to-right sum over the run (:870-872)
            for (int q = a; q <= z; ++q) prd = prd + s_row[ld_code(x, (uint32_t)q) * R_W + R_LLR];
            row->core_score = C.best[k];
            row->core_start = C.bstart[k];
            row->core_end = C.bstart[k] + (int)c[k] - 1;
            row->prd_score = prd;
            row->prd_start = a;
            row->prd_end = z;
        } else { // :873-880
            row->core_score = __builtin_nan("");
            row->core_start = -1;
            row->core_end = -2;
            row->prd_score = 0.0;
            row->prd_start = -1;
            row->prd_end = -2;
        }
    }
}

// ---- role F: forward (:3354-3375) with logeapeb (:1024-1047); track mode adds backward, posteriors
//      and the MAP path (:3377-3405, :4032-4045) ----
struct FwdState {
    double a0, a1, h0;
};

template <bool GUARD, bool TRACKS>
__device__ __forceinline__ void fwd_block(FwdState &S, const double *__restrict__ s_row,
                                          const double *__restrict__ s_lut, const uint4 cur, uint32_t t0, uint32_t n,
                                          double lt00, double lt01, double lt10, double lt11, double h0lt, int jfirst,
                                          double2 *__restrict__ fw) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < jfirst) continue;
        if (!GUARD || t0 + (uint32_t)j < n) {
            const double *__restrict__ r = s_row + block_code(cur, j) * R_W;
            const double e0 = r[R_LE0], e1 = r[R_LE1], eh = r[R_LE0H];
            // (:3360-3368): LSE(-inf, u) returns u unchanged, so one LUT-LSE per state
            const double f0 = lse_lut(s_lut, lt00 + S.a0, lt10 + S.a1);
            const double f1 = lse_lut(s_lut, lt01 + S.a0, lt11 + S.a1);
            S.a0 = f0 + e0;
            S.a1 = f1 + e1;
            S.h0 = (h0lt + S.h0) + eh;
            if (TRACKS) fw[(size_t)(t0 + (uint32_t)j) * 64u] = make_double2(S.a0, S.a1);
        }
    }
}

// backward recurrence (:3377-3391) over one block, t = t0+15 .. t0: stores b[.][t] next to the forward pair
// (the posterior needs a[.][t] + b[.][t], :3403) and steps b to t-1 with the emission of residue t
template <bool GUARD>
__device__ __forceinline__ void bwd_block(double &b0, double &b1, const double *__restrict__ s_row,
                                          const double *__restrict__ s_lut, const uint4 cur, uint32_t t0, uint32_t n,
                                          double lt00, double lt01, double lt10, double lt11,
                                          double2 *__restrict__ bw) {
#pragma unroll
    for (int j = 15; j >= 0; --j) {
        const uint32_t t = t0 + (uint32_t)j;
        if (!GUARD || t < n) {
            bw[(size_t)t * 64u] = make_double2(b0, b1);
            const double *__restrict__ r = s_row + block_code(cur, j) * R_W;
            const double u0 = (lt00 + b0) + r[R_LE0], u1 = (lt01 + b1) + r[R_LE1];
            const double w0 = (lt10 + b0) + r[R_LE0], w1 = (lt11 + b1) + r[R_LE1];
            b0 = lse_lut(s_lut, u0, u1); // (for t = 0 this steps to "t = -1"; that value is never used)
            b1 = lse_lut(s_lut, w0, w1);
        }
    }
}

template <bool TRACKS>
__global__ __launch_bounds__(KA_THREADS) void k_fwd(const uint8_t *__restrict__ codes,
                                                    const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ neff,
                                                    const uint4 *__restrict__ order, uint32_t nprot,
                                                    const DevTables *__restrict__ T,
                                                    const uint4 *__restrict__ packed,
                                                    const uint32_t *__restrict__ grow, plaac_row *__restrict__ rows,
                                                    double2 *__restrict__ fwd) {
    __shared__ double s_lut[LUTLEN + 1];
    __shared__ double s_row[ROWS * R_W];
    for (int i = threadIdx.x; i < LUTLEN; i += KA_THREADS) s_lut[i] = T->loglut[i];
    if (threadIdx.x == 0) s_lut[LUTLEN] = 0.0;
    load_rows(s_row, T);
    __syncthreads();
    const LaneJob J = lane_job(offsets, neff, order, nprot);
    const uint32_t n = J.n;
    set_wave_priority(n);
    const uint32_t gid = blockIdx.x * KA_THREADS + threadIdx.x;
    if (gid >= nprot) return;
    plaac_row *row = rows + J.p;
    if (n == 0) {
        row->hmm_all = 0.0;
        return;
    }
    (void)codes;
    const PackedLane PL = packed_lane(packed, nullptr, grow);
    const double lt00 = T->lt[0][0], lt01 = T->lt[0][1], lt10 = T->lt[1][0], lt11 = T->lt[1][1];
    const double lf0 = T->lf[0], lf1 = T->lf[1];
    const double h0lt = T->h0_lt00;
    // track mode: forward pairs in the same group-interleaved row numbering as the residues (16 steps per row)
    double2 *fw = TRACKS ? fwd + ((size_t)grow[gid >> 6] * 16u) * 64u + (gid & 63u) : nullptr;

    FwdState F;
    {
        uint4 nxt = PL.chunk(0);
        {
            const double *__restrict__ r = s_row + block_code(nxt, 0) * R_W;
            F.a0 = T->li[0] + r[R_LE0];
            F.a1 = T->li[1] + r[R_LE1];
            F.h0 = T->h0_li0 + r[R_LE0H];
            if (TRACKS) fw[0] = make_double2(F.a0, F.a1);
        }
        for (uint32_t t0 = 0; t0 < n; t0 += 16u) {
            const uint4 cur = nxt;
            if (t0 + 16u < n) nxt = PL.chunk((t0 >> 4) + 1u);
            if (t0 == 0u) fwd_block<true, TRACKS>(F, s_row, s_lut, cur, t0, n, lt00, lt01, lt10, lt11, h0lt, 1, fw);
            else if (t0 + 16u <= n)
                fwd_block<false, TRACKS>(F, s_row, s_lut, cur, t0, n, lt00, lt01, lt10, lt11, h0lt, 0, fw);
            else fwd_block<true, TRACKS>(F, s_row, s_lut, cur, t0, n, lt00, lt01, lt10, lt11, h0lt, 0, fw);
        }
    }
    const double lmarg1 = lse_lut(s_lut, F.a0 + lf0, F.a1 + lf1); // (:3369-3375)
    row->hmm_all = lmarg1 - (F.h0 + T->h0_lf0);

}

// ---- masked core window, latency form (wave-groups whose longest protein has >= CORE_LONG_ROWS rows) ----
// In k_vit the third sweep walks every protein once more with ~22 instructions per step; for the wave that holds
// the longest protein that is another n x 160 cycles on one SIMD while the chip idles. Only ONE thing in it is serial:
// the masked prefix sum psum[i+1] = psum[i] + maa3[i] (:818-823, its rounding is part of the answer). So:
//   k_core_chain   lane per protein: the prefix chain alone (lookup + add + store), P[t] = psum[t+1]
//   k_core_eval    16 waves per group, a wave per packed row: every window d = P[t] - P[t-c] (the trailing chain of the
//                  in-kernel form IS the leading chain c steps earlier, bit for bit), first maximum per row
//   k_core_reduce  lane per protein: first maximum over the rows (strict >, increasing row), then PRD as in k_vit
struct CorePart {
    double d;
    int start, pad;
};

__global__ __launch_bounds__(64) void k_core_chain(const uint4 *__restrict__ order, uint32_t nprot,
                                                   const DevTables *__restrict__ T, const uint4 *__restrict__ packed,
                                                   const uint32_t *__restrict__ grow, const uint32_t *__restrict__ bits,
                                                   double *__restrict__ pfx) {
    __shared__ double s_mask[64];
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t r0 = grow[g];
    if (grow[g + 1] - r0 < CORE_LONG_ROWS) return;
    {
        const int k = lane & 31;
        s_mask[lane] = lane < 32 ? T->big_neg : (k < NAA ? T->llr[k] : T->llr[0]);
    }
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const uint32_t gid = g * 64u + lane;
    const uint32_t n = gid < nprot ? order[gid].z : 0u;
    if (n == 0u) return;
    const uint4 *__restrict__ pk = packed + (size_t)r0 * 64u + lane;
    const uint32_t *__restrict__ wb = bits + (size_t)r0 * 64u + lane;
    double *__restrict__ pf = pfx + (size_t)r0 * 1024u + lane;
    double mL = 0.0;
    uint4 nxt = pk[0];
    uint32_t wnext = wb[0];
    for (uint32_t t0 = 0; t0 < n; t0 += 16u) {
        const uint4 cur = nxt;
        const uint32_t wl = wnext;
        if (t0 + 16u < n) {
            nxt = pk[(size_t)((t0 >> 4) + 1u) * 64u];
            wnext = wb[(size_t)((t0 >> 4) + 1u) * 64u];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { // positions past the end of a shorter protein extend the chain harmlessly
            mL = mL + s_mask[(((wl >> j) & 1u) << 5) | block_code(cur, j)]; // psum[i+1] = psum[i] + maa3[i]
            pf[(size_t)(t0 + (uint32_t)j) * 64u] = mL;
        }
    }
}

__global__ __launch_bounds__(64) void k_core_eval(const uint4 *__restrict__ order, uint32_t nprot, uint32_t ngroups,
                                                  const uint32_t *__restrict__ grow, const double *__restrict__ pfx,
                                                  CorePart *__restrict__ part, uint32_t c) {
    const uint32_t r = blockIdx.x, lane = threadIdx.x; // one wave per packed row of the long wave-groups
    const uint32_t glong = ngroups < CORE_MAX_GROUPS ? ngroups : CORE_MAX_GROUPS;
    if (r >= grow[glong]) return; // the long groups are the first ones of the descending-length plan
    // group of this row: the largest g with grow[g] <= r (64-ary search, wave-uniform; as k_post)
    uint32_t lo = 0, hi = glong;
    while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + (lane + 1u) * step;
        const bool le = idx < hi && grow[idx] <= r;
        const uint32_t k = (uint32_t)__popcll(__ballot(le));
        lo += k * step;
        hi = lo + step < hi ? lo + step : hi;
    }
    const uint32_t g = lo, r0 = grow[g];
    if (grow[g + 1] - r0 < CORE_LONG_ROWS) return;
    const uint32_t gid = g * 64u + lane;
    const uint32_t n = gid < nprot ? order[gid].z : 0u;
    const double *__restrict__ pf = pfx + (size_t)r0 * 1024u + lane;
    double best = -INFINITY;
    int bstart = -1;
    const uint32_t t0 = (r - r0) << 4;
    if (t0 < n) {
        double lead[16], trail[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { // all 32 loads in flight together
            const uint32_t t = t0 + (uint32_t)j;
            const bool in = t < n && t + 1u >= c;
            lead[j] = in ? pf[(size_t)t * 64u] : 0.0;
            trail[j] = (in && t >= c) ? pf[(size_t)(t - c) * 64u] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t t = t0 + (uint32_t)j;
            if (t < n && t + 1u >= c) { // window [t+1-c, t]
                const double d = t + 1u == c ? lead[j] : lead[j] - trail[j]; // window 0: psum[c] itself (:1226)
                const bool upd = d > best; // strict >: the first window wins ties
                best = upd ? d : best;
                bstart = upd ? (int)(t + 1u - c) : bstart;
            }
        }
    }
    part[(size_t)r * 64u + lane] = CorePart{best, bstart, 0};
}

__global__ __launch_bounds__(64) void k_core_reduce(const uint8_t *__restrict__ codes, const uint4 *__restrict__ order,
                                                    uint32_t nprot, const DevTables *__restrict__ T,
                                                    const uint32_t *__restrict__ grow, const uint32_t *__restrict__ bits,
                                                    const CorePart *__restrict__ part, plaac_row *__restrict__ rows,
                                                    uint32_t c) {
    __shared__ double s_llr[ROWS];
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t r0 = grow[g];
    if (grow[g + 1] - r0 < CORE_LONG_ROWS) return;
    if (lane < (uint32_t)ROWS) s_llr[lane] = T->llr[lane < (uint32_t)NAA ? lane : 0];
    __syncthreads();
    const uint32_t gid = g * 64u + lane;
    if (gid >= nprot) return;
    const uint4 it = order[gid];
    const uint32_t n = it.z;
    if (n == 0u) return; // k_vit has zeroed the fields of a skipped record
    plaac_row *row = rows + it.w;
    const uint8_t *__restrict__ x = codes + (((uint64_t)it.y << 32) | it.x);
    const CorePart *__restrict__ pt = part + (size_t)r0 * 64u + lane;
    const uint32_t *__restrict__ wb = bits + (size_t)r0 * 64u + lane;
    const uint32_t nw = (n + 15u) >> 4;
    double best = -INFINITY;
    int bstart = -1;
    for (uint32_t r = 0; r < nw; r += 8u) { // eight independent loads in flight, then the ordered comparison
        CorePart q[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) q[k] = r + k < nw ? pt[(size_t)(r + k) * 64u] : CorePart{-INFINITY, -1, 0};
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const bool upd = q[k].d > best;
            best = upd ? q[k].d : best;
            bstart = upd ? q[k].start : bstart;
        }
    }
    if (best > T->big_neg / 2) { // :861 - a core exists; expand it to the whole Viterbi run (:863-866)
        auto bit = [&](int q) { return (wb[(size_t)((uint32_t)q >> 4) * 64u] >> (q & 15)) & 1u; };
        int a = bstart, z = bstart + (int)c - 1;
        while (a > 0 && bit(a - 1)) --a;
        while (z + 1 < (int)n && bit(z + 1)) ++z;
        double prd = 0.0; // PRDscore: left-to-right sum over the run (:870-872)
        for (int q = a; q <= z; ++q) {
            const uint32_t cq = x[q];
            prd = prd + s_llr[cq < 22u ? cq : 22u];
        }
        row->core_score = best;
        row->core_start = bstart;
        row->core_end = bstart + (int)c - 1;
        row->prd_score = prd;
        row->prd_start = a;
        row->prd_end = z;
    } else { // :873-880
        row->core_score = __builtin_nan("");
        row->core_start = -1;
        row->core_end = -2;
        row->prd_score = 0.0;
        row->prd_start = -1;
        row->prd_end = -2;
    }
}

// ---- role F, latency form: TWO LANES PER PROTEIN (lane 2q: background state, lane 2q+1: PrD state). One wave
//      instruction advances both states of 32 proteins, so the wave that holds the longest protein issues half the fp64
//      operations per step (a single wave issues one fp64 operation per ~8 cycles: the chain of the longest protein,
//      not throughput, bounds small batches). The partner's value crosses with one quad-permute DPP move per dword.
//      hmm0's running sum is left to k_win (role 1); k_finish forms HMMall / HMMvit once both are known.
__device__ __forceinline__ double dpp_pair_swap_f64(double v) { // lane 2q <-> lane 2q+1
    const long long u = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u & 0xffffffffll), 0xB1, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)u >> 32), 0xB1, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <bool GUARD>
__device__ __forceinline__ void fwd_pair_block(double &a, const double *__restrict__ s_row,
                                               const double *__restrict__ s_lut, const uint4 cur, uint32_t t0, uint32_t n,
                                               double ltself, double ltcross, int ecol, int jfirst) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < jfirst) continue;
        if (!GUARD || t0 + (uint32_t)j < n) {
            const double e = s_row[block_code(cur, j) * R_W + ecol];
            // (:3360-3368) a[i][t] = LSE(lt[0][i] + a[0][t-1], lt[1][i] + a[1][t-1]) + le[i][x_t]; the LUT-LSE is symmetric
            const double f = lse_lut(s_lut, ltself + a, ltcross + dpp_pair_swap_f64(a));
            a = f + e;
        }
    }
}

__global__ __launch_bounds__(KA_THREADS) void k_fwd_pair(const uint4 *__restrict__ order, uint32_t nprot,
                                                         const DevTables *__restrict__ T,
                                                         const uint4 *__restrict__ packed,
                                                         const uint32_t *__restrict__ grow, double *__restrict__ lmarg) {
    __shared__ double s_lut[LUTLEN + 1];
    __shared__ double s_row[ROWS * R_W];
    for (int i = threadIdx.x; i < LUTLEN; i += KA_THREADS) s_lut[i] = T->loglut[i];
    if (threadIdx.x == 0) s_lut[LUTLEN] = 0.0;
    load_rows(s_row, T);
    __syncthreads();
    const uint32_t st = threadIdx.x & 1u;                                  // this lane's state
    const uint32_t gid = blockIdx.x * (KA_THREADS / 2) + (threadIdx.x >> 1); // plan index of the lane pair's protein
    uint32_t n = 0, p = 0;
    if (gid < nprot) {
        const uint4 it = order[gid];
        n = it.z;
        p = it.w;
    }
    set_wave_priority(n);
    if (gid >= nprot) return;
    if (n == 0) {
        if (st == 0u) lmarg[p] = 0.0;
        return;
    }
    const uint4 *__restrict__ pk = packed + (size_t)grow[gid >> 6] * 64u + (gid & 63u);
    const double ltself = T->lt[st][st], ltcross = T->lt[1u - st][st], lf = T->lf[st];
    const int ecol = st ? R_LE1 : R_LE0;
    uint4 nxt = pk[0];
    double a = T->li[st] + s_row[block_code(nxt, 0) * R_W + ecol];
    for (uint32_t t0 = 0; t0 < n; t0 += 16u) {
        const uint4 cur = nxt;
        if (t0 + 16u < n) nxt = pk[(size_t)((t0 >> 4) + 1u) * 64u];
        if (t0 == 0u) fwd_pair_block<true>(a, s_row, s_lut, cur, t0, n, ltself, ltcross, ecol, 1);
        else if (t0 + 16u <= n) fwd_pair_block<false>(a, s_row, s_lut, cur, t0, n, ltself, ltcross, ecol, 0);
        else fwd_pair_block<true>(a, s_row, s_lut, cur, t0, n, ltself, ltcross, ecol, 0);
    }
    const double mine = a + lf; // (:3369-3375)
    const double lm = lse_lut(s_lut, mine, dpp_pair_swap_f64(mine));
    if (st == 0u) lmarg[p] = lm;
}

// latency forms: HMMall = lmarginalprob(hmm1) - total(hmm0), HMMvit = lviterbiprob(hmm1) - total(hmm0) (:797-798) once the
// three kernels that produced the terms have finished (k_fwd_pair -> lmarg, k_vit<.,true> -> row.hmm_vit, k_win role 1 -> h0)
__global__ void k_finish(plaac_row *__restrict__ rows, const double *__restrict__ lmarg, const double *__restrict__ h0,
                         uint32_t nprot) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprot) return;
    const double h = h0[p];
    rows[p].hmm_all = lmarg[p] - h;
    rows[p].hmm_vit = rows[p].hmm_vit - h;
}

// ---- role B (track mode): backward recurrence (:3377-3391). It does not depend on the forward values, so it is
//      its own chain on its own stream, concurrent with k_fwd; k_post combines the two. ----
__global__ __launch_bounds__(KA_THREADS) void k_bwd(const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ neff,
                                                    const uint4 *__restrict__ order, uint32_t nprot,
                                                    const DevTables *__restrict__ T,
                                                    const uint4 *__restrict__ packed,
                                                    const uint32_t *__restrict__ grow, double2 *__restrict__ bwd) {
    __shared__ double s_lut[LUTLEN + 1];
    __shared__ double s_row[ROWS * R_W];
    for (int i = threadIdx.x; i < LUTLEN; i += KA_THREADS) s_lut[i] = T->loglut[i];
    if (threadIdx.x == 0) s_lut[LUTLEN] = 0.0;
    load_rows(s_row, T);
    __syncthreads();
    const LaneJob J = lane_job(offsets, neff, order, nprot);
    const uint32_t n = J.n;
    set_wave_priority(n);
    const uint32_t gid = blockIdx.x * KA_THREADS + threadIdx.x;
    if (gid >= nprot || n == 0) return;
    const PackedLane PL = packed_lane(packed, nullptr, grow);
    const double lt00 = T->lt[0][0], lt01 = T->lt[0][1], lt10 = T->lt[1][0], lt11 = T->lt[1][1];
    double2 *bw = bwd + ((size_t)grow[gid >> 6] * 16u) * 64u + (gid & 63u);
    double b0 = T->lf[0], b1 = T->lf[1]; // b[.][n-1] = lfprob (:3378)
    const uint32_t nw = (n + 15u) >> 4;
    // phase-locked downwards from the group's last block (see k_vit's traceback): coalesced row loads
    const uint32_t nwmax = wave_max_u32(nw);
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    uint4 nxt = nwmax - 1u < nw ? PL.chunk(nwmax - 1u) : zero4;
    for (uint32_t wi = nwmax; wi-- > 0u;) {
        const uint4 cur = nxt;
        if (wi > 0u) nxt = wi - 1u < nw ? PL.chunk(wi - 1u) : zero4;
        if (wi < nw) {
            const uint32_t t0 = wi << 4;
            if (t0 + 16u <= n) bwd_block<false>(b0, b1, s_row, s_lut, cur, t0, n, lt00, lt01, lt10, lt11, bw);
            else bwd_block<true>(b0, b1, s_row, s_lut, cur, t0, n, lt00, lt01, lt10, lt11, bw);
        }
    }
}

// ---- posteriors / MAP / Viterbi bytes of track mode (:3393-3405, :4037-4041): position-parallel ----
// One wave per packed row (= 16 steps of the 64 proteins of one wave-group): finds its group in the row prefix
// sums, reads the interleaved forward and backward pairs coalesced (1 KiB per step and array), takes the exp,
// transposes through LDS and writes each protein's 16 consecutive values as one contiguous 128-byte run.
constexpr int PT = 16; // steps per tile = one packed row
__global__ __launch_bounds__(64) void k_post(const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ neff,
                                              const uint4 *__restrict__ order, uint32_t nprot, uint32_t ngroups,
                                              const uint32_t *__restrict__ grow, const DevTables *__restrict__ T,
                                              const double2 *__restrict__ fwd, const double2 *__restrict__ bwd,
                                              const uint32_t *__restrict__ bits, TrackPtrs tr) {
    __shared__ double s0[PT][65], s1[PT][65];
    __shared__ uint8_t sb[PT][64]; // bit 0 = Viterbi state, bit 1 = MAP state
    __shared__ uint32_t sh_n[64];
    __shared__ uint64_t sh_off[64];
    const uint32_t r = blockIdx.x; // packed row
    const int lane = threadIdx.x;
    // group of this row: the largest g with grow[g] <= r (64-ary search, wave-uniform; empty groups repeat a value)
    uint32_t lo = 0, hi = ngroups;
    while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + ((uint32_t)lane + 1u) * step;
        const bool le = idx < hi && grow[idx] <= r;
        const uint32_t k = (uint32_t)__popcll(__ballot(le));
        lo += k * step;
        hi = lo + step < hi ? lo + step : hi;
    }
    const uint32_t g = lo;
    const size_t rbase = (size_t)grow[g];
    const uint32_t t0 = (r - (uint32_t)rbase) * 16u;
    const uint32_t gid = g * 64u + (uint32_t)lane;
    uint32_t n = 0;
    uint64_t off = 0;
    double lp = 0.0;
    if (gid < nprot) {
        const uint4 it = order[gid];
        n = it.z;
        off = ((uint64_t)it.y << 32) | it.x;
        if (n) { // lpseq (:3393-3396) from a[.][0] + b[.][0]
            const double2 a = fwd[rbase * 16u * 64u + (size_t)lane], b = bwd[rbase * 16u * 64u + (size_t)lane];
            lp = lse_lut(T->loglut, a.x + b.x, a.y + b.y);
        }
    }
    sh_n[lane] = n;
    sh_off[lane] = off;
    const double2 *__restrict__ fw = fwd + (size_t)r * 16u * 64u + (size_t)lane;
    const double2 *__restrict__ bw = bwd + (size_t)r * 16u * 64u + (size_t)lane;
    const uint32_t w = t0 < n ? bits[(size_t)r * 64u + (size_t)lane] : 0u;
#pragma unroll 4
    for (int tt = 0; tt < PT; ++tt) {
        if (t0 + (uint32_t)tt < n) {
            const double2 a = fw[(size_t)tt * 64u], b = bw[(size_t)tt * 64u];
            const double pp0 = exp((a.x + b.x) - lp), pp1 = exp((a.y + b.y) - lp); // (:3403)
            s0[tt][lane] = pp0;
            s1[tt][lane] = pp1;
            sb[tt][lane] = (uint8_t)(((w >> tt) & 1u) | (pp1 > pp0 ? 2u : 0u)); // MAP ties -> 0 (:4039)
        }
    }
    __syncthreads();
    const int sub = lane >> 4, tt2 = lane & 15;
    const uint32_t t = t0 + (uint32_t)tt2;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int L = 4 * it + sub;
        if (t < sh_n[L]) {
            const uint64_t o = sh_off[L] + t;
            tr.post0[o] = s0[tt2][L];
            tr.post1[o] = s1[tt2][L];
            const uint8_t bb = sb[tt2][L];
            tr.vit[o] = bb & 1u;
            tr.map[o] = bb >> 1;
        }
    }
}

// ---- role W: MW (:767-771) and LLR (:782-783) windows over prefix sums (hss2 :1206-1257 with
//      min == max), mean hydropathy / charge / FoldIndex (:4877-4885) ----
template <int NC>
struct WinState {
    double hydsum, psL, h0;
    double psT[NC], llrbest[NC];
    int chg, cntL, cntT, mwbest, mwstart;
    int llrstart[NC];
};

__device__ __forceinline__ int is_nq(uint32_t c) { return (c == 12u || c == 14u) ? 1 : 0; }

// STEADY: the whole block has t >= max(c, 80): all trailing streams run and every step closes every window
// ROLE 0: everything; 1: MW window + means; 2: LLR window(s); 3: role 1 + hmm0's running sum (for k_finish); 4: MW window;
// 5: means + hmm0's running sum. When the step is bound by the serial chain of the
// longest protein (small batches, very long proteins) the two halves run as two kernels side by side: each wave then
// issues about half the instructions per step.
template <bool GUARD, bool STEADY, int NC, int ROLE>
__device__ __forceinline__ void win_block(WinState<NC> &S, const double *__restrict__ s_row, const uint4 cur,
                                          const uint4 (&ccur)[NC], const uint4 mcur, uint32_t t0, uint32_t n,
                                          const uint32_t (&c)[NC], uint32_t mw, double h0li, double h0lt) {
    constexpr bool DO_MW = ROLE == 0 || ROLE == 1 || ROLE == 3 || ROLE == 4, DO_MEAN = ROLE == 0 || ROLE == 1 || ROLE == 3 || ROLE == 5,
                   DO_LLR = ROLE == 0 || ROLE == 2, DO_H0 = ROLE == 3 || ROLE == 5;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t t = t0 + (uint32_t)j;
        if (!GUARD || t < n) {
            const uint32_t xc = block_code(cur, j);
            const double *__restrict__ r = s_row + xc * R_W;
            if (DO_H0) { // hmm0: S = li0 + le0[x0]; S = (lt00 + S) + le0[xt]
                const double eh = r[R_LE0H];
                S.h0 = (!STEADY && t == 0u) ? h0li + eh : (h0lt + S.h0) + eh;
            }
            if (DO_MEAN) {
                S.hydsum = S.hydsum + r[R_HYD]; // mean (:1584-1588)
                S.chg += (xc == 3u || xc == 4u) ? 1 : ((xc == 9u || xc == 15u) ? -1 : 0);
            }
            if (DO_MW) {
                S.cntL += is_nq(xc);
                if (STEADY || t >= 80u) S.cntT += is_nq(block_code(mcur, j)); // only reached when mw == 80
                if (STEADY) {
                    const int d = S.cntL - S.cntT;
                    const bool upd = d > S.mwbest;
                    S.mwbest = upd ? d : S.mwbest;
                    S.mwstart = upd ? (int)(t + 1u - 80u) : S.mwstart;
                } else {
                    const int d = S.cntL - S.cntT;
                    const bool upd = (t + 1 >= mw) && (t + 1 == mw || d > S.mwbest);
                    S.mwbest = upd ? d : S.mwbest;
                    S.mwstart = upd ? (int)(t + 1 - mw) : S.mwstart;
                }
            }
            // psum[i+1] = psum[i] + llr[x_i]; each trailing prefix sum is the same chain c steps later
            if (DO_LLR) S.psL = S.psL + r[R_LLR];
#pragma unroll
            for (int k = 0; k < (DO_LLR ? NC : 0); ++k) {
                if (STEADY || t >= c[k]) S.psT[k] = S.psT[k] + s_row[block_code(ccur[k], j) * R_W + R_LLR];
                if (STEADY || t + 1 >= c[k]) {
                    const bool first = !STEADY && t + 1 == c[k];
                    const double d = first ? S.psL : S.psL - S.psT[k];
                    const bool upd = first || d > S.llrbest[k];
                    S.llrbest[k] = upd ? d : S.llrbest[k];
                    S.llrstart[k] = upd ? (int)(t + 1 - c[k]) : S.llrstart[k];
                }
            }
        }
        if (ROLE == 0 && (j & 3) == 3) { // keeps the integer side work of four steps from piling up at the block's end (80 VGPRs
                            // instead of 86: the wave then fits beside three window-kernel waves of 144)
            asm volatile("" : "+v"(S.mwbest), "+v"(S.chg));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NC, int ROLE>
__global__ __launch_bounds__(KA_THREADS) void k_win(const uint8_t *__restrict__ codes,
                                                    const uint64_t *__restrict__ offsets,
                                                    const uint32_t *__restrict__ neff,
                                                    const uint4 *__restrict__ order, uint32_t nprot,
                                                    const DevTables *__restrict__ T,
                                                    const uint4 *__restrict__ packed,
                                                    const uint32_t *__restrict__ grow, SweepTargets tg,
                                                    double *__restrict__ h0out) {
    constexpr bool DO_MW = ROLE == 0 || ROLE == 1 || ROLE == 3 || ROLE == 4, DO_MEAN = ROLE == 0 || ROLE == 1 || ROLE == 3 || ROLE == 5,
                   DO_LLR = ROLE == 0 || ROLE == 2, DO_H0 = ROLE == 3 || ROLE == 5;
    __shared__ double s_row[ROWS * R_W];
    load_rows(s_row, T);
    __syncthreads();
    const LaneJob J = lane_job(offsets, neff, order, nprot);
    const uint32_t n = J.n;
    set_wave_priority(n);
    if (blockIdx.x * KA_THREADS + threadIdx.x >= nprot) return;
    if (n == 0) {
        if (DO_H0) h0out[J.p] = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            plaac_row *row = tg.rows[k] + J.p;
            if (DO_MEAN) row->fi_meanhydro = row->fi_meancharge = row->fi_meancombo = 0.0;
            if (DO_MW) {
                row->mw_score = row->mw_start = row->mw_end = 0;
                row->prot_len = 0;
            }
            if (DO_LLR) {
                row->llr_score = 0.0;
                row->llr_start = row->llr_end = 0;
            }
        }
        return;
    }
    const PackedLane PL = packed_lane(packed, nullptr, grow);
    uint32_t c[NC];
    uint32_t steady_from = 80u;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        c[k] = tg.c[k];
        steady_from = c[k] > steady_from ? c[k] : steady_from;
    }
    const uint32_t mw = n < 80u ? n : 80u; // :769-770 (a protein shorter than 80 has a single window)

    WinState<NC> W;
    W.hydsum = W.psL = W.h0 = 0.0;
    const double h0li = T->h0_li0, h0lt = T->h0_lt00;
    W.chg = W.cntL = W.cntT = W.mwbest = W.mwstart = 0;
    // phase-locked streams: residues at t, at t - c[k] (LLR windows) and at t - 80 (MW window)
    uint4 nxt = PL.chunk(0), cnxt[NC], mnxt = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        W.psT[k] = 0.0;
        W.llrbest[k] = -INFINITY;
        W.llrstart[k] = -1;
        cnxt[k] = make_uint4(0u, 0u, 0u, 0u);
        if (15 >= (int)c[k]) cnxt[k] = PL.window(-(int)c[k]);
    }
    for (uint32_t t0 = 0; t0 < n; t0 += 16u) {
        const uint4 cur = nxt, mcur = mnxt;
        uint4 ccur[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) ccur[k] = cnxt[k];
        if (t0 + 16u < n) {
            nxt = PL.chunk((t0 >> 4) + 1u);
            const int sm = (int)t0 + 16 - 80;
            if (DO_MW && sm + 15 >= 0) mnxt = PL.window(sm);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int sc = (int)t0 + 16 - (int)c[k];
                if (DO_LLR && sc + 15 >= 0) cnxt[k] = PL.window(sc);
            }
        }
        const bool full = t0 + 16u <= n;
        if (t0 >= steady_from) { // wave-uniform
            if (full) win_block<false, true, NC, ROLE>(W, s_row, cur, ccur, mcur, t0, n, c, mw, h0li, h0lt);
            else win_block<true, true, NC, ROLE>(W, s_row, cur, ccur, mcur, t0, n, c, mw, h0li, h0lt);
        } else {
            win_block<true, false, NC, ROLE>(W, s_row, cur, ccur, mcur, t0, n, c, mw, h0li, h0lt);
        }
    }
    if (DO_H0) h0out[J.p] = W.h0 + T->h0_lf0; // hmm0.lviterbiprob == hmm0.lmarginalprob
    const double meanhydro = (1.0 * W.hydsum) / (double)(int)n;
    const double meancharge = (1.0 * (double)W.chg) / (double)(int)n;
    const double meanfi = (T->cc[2] + T->cc[1] * fabs(meancharge)) + T->cc[0] * meanhydro; // :4885
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        plaac_row *row = tg.rows[k] + J.p;
        if (DO_MW) {
            row->prot_len = (int32_t)n;
            row->mw_score = W.mwbest;
            row->mw_start = W.mwstart;
            row->mw_end = W.mwstart + (int)mw - 1;
        }
        if (DO_MEAN) {
            row->fi_meanhydro = meanhydro;
            row->fi_meancharge = meancharge;
            row->fi_meancombo = meanfi;
        }
        if (DO_LLR) {
            row->llr_score = W.llrbest[k];
            row->llr_start = W.llrstart[k];
            row->llr_end = W.llrstart[k] < 0 ? -2 : W.llrstart[k] + (int)c[k] - 1;
        }
    }
}

// copies the fields that do not depend on the core length (k_fwd's and K-B's) from the first row array of a
// sweep group to the other core lengths' row arrays
__global__ void k_replicate(const plaac_row *__restrict__ src, SweepTargets tg, int ndst, uint32_t nprot) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprot) return;
    const plaac_row r = src[p];
    for (int k = 1; k <= ndst; ++k) {
        plaac_row *d = tg.rows[k] + p;
        d->hmm_all = r.hmm_all;
        d->papa_combo = r.papa_combo;
        d->papa_prop = r.papa_prop;
        d->papa_fi = r.papa_fi;
        d->papa_llr = r.papa_llr;
        d->papa_llr2 = r.papa_llr2;
        d->fi_numaa = r.fi_numaa;
        d->fi_maxrun = r.fi_maxrun;
        d->papa_cen = r.papa_cen;
    }
}

// ------------------------------------------------------------------------------------------------
// K-B: one wave per protein — window tracks (disorderreport :4866-5068)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// sum over p in [i-w, i+w] of min(p, w) for i >= w  (weights of the second smoothing, exact integers)
__device__ __forceinline__ int sum_min_left(int i, int w) {
    // p < w contributes p, otherwise w
    if (i - w >= w) return (2 * w + 1) * w;
    const int lo = i - w;            // first p
    const int cnt = w - lo;          // p = lo .. w-1
    const int s = (lo + (w - 1)) * cnt / 2;
    return s + (2 * w + 1 - cnt) * w;
}

template <int RING, bool TRACKS>
__global__ __launch_bounds__(64) void k_tracks(const uint8_t *__restrict__ codes, const uint64_t *__restrict__ offsets,
                                               const uint32_t *__restrict__ neff,
                                               const uint4 *__restrict__ order, uint32_t nprot,
                                               const DevTables *__restrict__ T, plaac_row *__restrict__ rows,
                                               TrackPtrs tr) {
    constexpr int M = RING - 1;
    __shared__ double t_hyd[NAA], t_llr[NAA], t_lod[NAA];
    __shared__ int t_chg[NAA];
    __shared__ double in_h[RING], in_l[RING], in_p[RING];   // mapped inputs, 0.0 outside [0,n)
    __shared__ int in_c[RING];                               // charge as integer
    __shared__ double w_fi[RING], w_ll[RING], w_pa[RING];    // weight * first-level track
    __shared__ double r_ll[RING];                            // first-level PLAAC-LLR (for PAPAllr)

    const int lane = threadIdx.x;
    const uint4 it = order[blockIdx.x];
    const uint32_t p = it.w;
    const int n = (int)it.z;
    plaac_row *row = rows + p;
    if (n == 0) {
        if (lane == 0) {
            row->papa_combo = row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = 0.0;
            row->fi_numaa = row->fi_maxrun = row->papa_cen = 0;
        }
        return;
    }
    if (lane < NAA) {
        t_hyd[lane] = T->hyd[lane];
        t_llr[lane] = T->llr[lane];
        t_lod[lane] = T->lod[lane];
        t_chg[lane] = T->chg[lane];
    }
    for (int i = lane; i < RING; i += 64) {
        in_h[i] = 0.0;
        in_l[i] = 0.0;
        in_p[i] = 0.0;
        in_c[i] = 0;
        w_fi[i] = 0.0;
        w_ll[i] = 0.0;
        w_pa[i] = 0.0;
        r_ll[i] = 0.0;
    }
    const uint64_t off = ((uint64_t)it.y << 32) | it.x;
    const uint8_t *__restrict__ x = codes + off;
    const int ww1 = T->ww1, ww2 = T->ww2, ww3 = T->ww3;
    // w = ww/2 clamped to n-1 (:2588-2589)
    const int w1 = imin(ww1 / 2, n - 1), w2 = imin(ww2 / 2, n - 1), w3 = imin(ww3 / 2, n - 1);
    const int wm = imax(w1, imax(w2, w3));
    const bool adjust = T->adjustprolines != 0;
    const double cc0 = T->cc[0], cc1 = T->cc[1], cc2 = T->cc[2];
    // FoldIndex run scan domain (:5010-5013) and PAPA centre range (:4942)
    int halfw = (ww1 - 1) / 2;
    if (halfw > n / 2) halfw = n / 2;
    const int dlo = halfw, dhi = n - halfw - 1;
    const int plo = (ww2 - 1) / 2, phi = n - (ww2 - 1) / 2; // k in [plo, phi)

    // per-lane PAPA arg-max state (positions visited in increasing order -> strict > keeps the first max)
    double pbest = -INFINITY, pfi = 0.0, pll = 0.0, pll2 = 0.0;
    int pcen = -1;
    // wave-uniform FoldIndex run state
    bool run_open = false;
    int run_start = 0, numaa = 0, maxlen = 0;
    auto close_run = [&](int s, int e) {
        if (s == dlo) s = 0;
        if (e == dhi) e = n - 1;
        const int len = e - s + 1;
        if (len >= 5) {
            numaa += len;
            maxlen = len > maxlen ? len : maxlen;
        }
    };
    __syncthreads();

    const int nchunks = (n + 2 * wm + 63) / 64;
    for (int k = 0; k < nchunks; ++k) {
        // ---- stage 0: residues -> mapped inputs at q = 64k + lane
        {
            const int q = 64 * k + lane;
            double vh = 0.0, vl = 0.0, vp = 0.0;
            int vc = 0;
            if (q < n) {
                const uint32_t cq = ld_code(x, (uint32_t)q);
                vh = t_hyd[cq];
                vl = t_llr[cq];
                vc = t_chg[cq];
                bool skip = false;
                if (adjust && cq == 13u) { // only the first P of PP / PxP scores (:2653-2654)
                    skip = (q >= 1 && ld_code(x, (uint32_t)(q - 1)) == 13u) ||
                           (q >= 2 && ld_code(x, (uint32_t)(q - 2)) == 13u);
                }
                vp = skip ? 0.0 : t_lod[cq];
            }
            in_h[q & M] = vh;
            in_l[q & M] = vl;
            in_p[q & M] = vp;
            in_c[q & M] = vc;
        }
        __syncthreads();
        // ---- stage 1: first-level tracks at i = 64k + lane - wm
        {
            const int i = 64 * k + lane - wm;
            const bool live = i >= 0 && i < n;
            bool neg = false;
            if (live) {
                double sh = 0.0, sl = 0.0, sp = 0.0;
                int sc = 0;
                for (int j = -w1; j <= w1; ++j) {
                    sh = sh + in_h[(i + j) & M];
                    sc += in_c[(i + j) & M];
                }
                for (int j = -w3; j <= w3; ++j) sl = sl + in_l[(i + j) & M];
                for (int j = -w2; j <= w2; ++j) sp = sp + in_p[(i + j) & M];
                const double d1 = (double)(imin(i + w1, n - 1) - imax(i - w1, 0) + 1);
                const double d2 = (double)(imin(i + w2, n - 1) - imax(i - w2, 0) + 1);
                const double d3 = (double)(imin(i + w3, n - 1) - imax(i - w3, 0) + 1);
                const double hydro = sh / d1;
                const double charge = (double)sc / d1;
                const double fi = (cc0 * hydro + cc1 * fabs(charge)) + cc2; // axpbypc (:2050)
                const double llr1 = sl / d3;
                const double papa = sp / d2;
                const double wt1 = (double)(1 + imin(i, w1) + imin(n - i - 1, w1));
                const double wt2 = (double)(1 + imin(i, w2) + imin(n - i - 1, w2));
                const double wt3 = (double)(1 + imin(i, w3) + imin(n - i - 1, w3));
                w_fi[i & M] = wt1 * fi;
                w_pa[i & M] = wt2 * papa;
                w_ll[i & M] = wt3 * llr1;
                r_ll[i & M] = llr1;
                neg = (fi < 0.0) && i >= dlo && i <= dhi;
                if (TRACKS) {
                    tr.charge[off + i] = charge;
                    tr.hydro[off + i] = hydro;
                    tr.fi[off + i] = fi;
                    tr.plaacllr[off + i] = llr1;
                    tr.papa[off + i] = papa;
                }
            }
            // FoldIndex<0 runs (:5020-5058): ballot -> wave-uniform run-length machine
            unsigned long long m = __ballot(neg);
            const int base = 64 * k - wm;
            if (run_open) {
                const int t1 = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
                if (t1 < 64) {
                    close_run(run_start, base + t1 - 1);
                    run_open = false;
                    m &= ~((1ull << t1) - 1ull);
                } else {
                    m = 0ull;
                }
            }
            while (m) {
                const int s = __builtin_ctzll(m);
                const unsigned long long rest = ~(m >> s);
                const int len = rest == 0ull ? 64 - s : __builtin_ctzll(rest);
                if (s + len >= 64) {
                    run_open = true;
                    run_start = base + s;
                    break;
                }
                close_run(base + s, base + s + len - 1);
                m &= ~(((1ull << len) - 1ull) << s);
            }
        }
        __syncthreads();
        // ---- stage 2: weighted second smoothing at i = 64k + lane - 2*wm, PAPA arg-max
        {
            const int i = 64 * k + lane - 2 * wm;
            if (i >= 0 && i < n) {
                double fix2 = __builtin_nan(""), llx2 = __builtin_nan(""), pax2 = __builtin_nan("");
                if (i >= w1 && i <= n - w1 - 1) {
                    double s = 0.0;
                    for (int j = -w1; j <= w1; ++j) s = s + w_fi[(i + j) & M];
                    const int den = (2 * w1 + 1) + sum_min_left(i, w1) + sum_min_left(n - 1 - i, w1);
                    fix2 = s / (double)den;
                }
                if (i >= w3 && i <= n - w3 - 1) {
                    double s = 0.0;
                    for (int j = -w3; j <= w3; ++j) s = s + w_ll[(i + j) & M];
                    const int den = (2 * w3 + 1) + sum_min_left(i, w3) + sum_min_left(n - 1 - i, w3);
                    llx2 = s / (double)den;
                }
                if (i >= w2 && i <= n - w2 - 1) {
                    double s = 0.0;
                    for (int j = -w2; j <= w2; ++j) s = s + w_pa[(i + j) & M];
                    const int den = (2 * w2 + 1) + sum_min_left(i, w2) + sum_min_left(n - 1 - i, w2);
                    pax2 = s / (double)den;
                }
                if (TRACKS) {
                    tr.fix2[off + i] = fix2;
                    tr.plaacllrx2[off + i] = llx2;
                    tr.papax2[off + i] = pax2;
                }
                if (i >= plo && i < phi && (pax2 > pbest) && (fix2 < 0.0)) { // papamode 1 (:4942-4948)
                    pbest = pax2;
                    pcen = i;
                    pfi = fix2;
                    pll2 = llx2;
                    pll = r_ll[i & M];
                }
            }
        }
        // the next iteration's stage 0 only overwrites ring slots that no later read needs
    }
    if (run_open) close_run(run_start, dhi);

    // wave arg-max: largest papax2, smallest centre among equals (first max of the serial loop)
    for (int d = 32; d >= 1; d >>= 1) {
        const double ob = __shfl_xor(pbest, d);
        const int oc = __shfl_xor(pcen, d);
        const double ofi = __shfl_xor(pfi, d), oll = __shfl_xor(pll, d), oll2 = __shfl_xor(pll2, d);
        const bool take = (oc >= 0) && (pcen < 0 || ob > pbest || (ob == pbest && oc < pcen));
        if (take) {
            pbest = ob;
            pcen = oc;
            pfi = ofi;
            pll = oll;
            pll2 = oll2;
        }
    }
    if (lane == 0) {
        row->fi_numaa = numaa;
        row->fi_maxrun = maxlen;
        row->papa_cen = pcen;
        if (pcen >= 0) {
            row->papa_combo = pbest;
            row->papa_prop = pbest;
            row->papa_fi = pfi;
            row->papa_llr = pll;
            row->papa_llr2 = pll2;
        } else {
            row->papa_combo = -INFINITY;
            row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = __builtin_nan("");
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K-B fast path: all three half-windows equal 20 (the default ww = 41, also ww = 40).
// B consecutive positions per lane (64*B positions per iteration), B chosen PER PROTEIN from {2, 3, 4} so
// that the last, partly empty iteration wastes as few position slots as possible (UniRef-shaped lengths:
// 65 % -> 80 % of the slots carry a residue). The 40+B values under the union of a lane's B windows are
// read once and feed all B fixed-order 41-term sums (B = 4: 0.27 LDS values per add instead of 1), which
// makes the loop fp64-VALU-bound instead of LDS-bound. Values live in LDS split by position mod B so that
// lane l reads [base + l + g] with an immediate offset g (conflict-free, no address arithmetic); the head
// of every sub-ring is mirrored behind it so base + g never wraps.
// Charge sums are exact integers -> prefix counts (wave scan) instead of 41 adds. The weights of the
// second smoothing depend only on the position, so weight*value is formed once per position. The four
// (three) quotients of a position share their denominator: one reciprocal refinement, then a 3-instruction
// correctly rounded quotient each (the same FMA sequence the compiler emits for a double division, minus the
// range scaling that small-integer denominators never need).
// Each block scores 16 proteins (descending-length order, stride gridDim.x, so every block gets the same mix of
// long and short ones) and prefetches the next protein's metadata, so the dependent
// order->length->offset->residues load chain is off the path.
// ------------------------------------------------------------------------------------------------
constexpr int TW = 20;        // half window
constexpr int RING_DOUBLES = 348; // doubles per ring (all sub-rings of one track), enough for every B below
enum { RG_WF = 0, RG_WL = 1, RG_WP = 2 };

// geometry of the B-positions-per-lane variant
template <int B>
struct KbGeom {
    static constexpr int C = 64 * B;                       // positions per iteration
    static constexpr int LAG1 = ((TW + B - 1) / B) * B;    // first-level lag: multiple of B, >= 20
    static constexpr int LAG2 = ((LAG1 + TW + B - 1) / B) * B; // second-level lag: multiple of B, >= LAG1 + 20
    static constexpr int C0 = ((-TW) % B + B) % B;          // class of the first window position (i0 - 20)
    static constexpr int GMAX = (2 * TW + B - 1 + C0) / B;  // largest slot offset a window read uses
    static constexpr int LIVE_LAG = (C + LAG2 - LAG1 + TW > C + LAG1 + TW ? C + LAG2 - LAG1 + TW : C + LAG1 + TW);
    static constexpr int LIVE_NOLAG = C + 2 * TW + C0 + B;  // single-iteration variant: positions -20-C0 .. C+20+B-1
    static constexpr int LIVE = LIVE_LAG > LIVE_NOLAG ? LIVE_LAG : LIVE_NOLAG;
    static constexpr int RB = (LIVE + B - 1) / B + 1;       // sub-ring entries per class
    static constexpr int MIR = GMAX + 1;                    // mirrored head
    static constexpr int SUB = RB + MIR;
    static_assert(B * SUB <= RING_DOUBLES, "ring too small");
};

__device__ __forceinline__ uint32_t load4(const uint8_t *p, const uint8_t *lo, const uint8_t *end) {
    if (p >= lo && p + 4 <= end) {
        uint32_t v;
        __builtin_memcpy(&v, p, 4);
        return v;
    }
    uint32_t v = 0u;
    for (int i = 0; i < 4; ++i) {
        const uint8_t *q = p + i;
        v |= ((q >= lo && q < end) ? (uint32_t)*q : 0u) << (8 * i);
    }
    return v;
}

// sum over p in [i-w, i+w] of min(p, w), for i >= w: every term is w except the first m = max(0, 2w-i),
// which fall short of w by m, m-1, ..., 1
__device__ __forceinline__ int window_weight_side(int i, int w) {
    const int m = imax(0, 2 * w - i);
    return (2 * w + 1) * w - ((m * (m + 1)) >> 1);
}

// lane broadcast through SGPRs (v_readlane), no LDS round trip; `lane` must be a compile-time constant
__device__ __forceinline__ int bcast_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double bcast_lane(double v, int lane) {
    const long long u = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffll), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double bcast_lane_dyn(double v, int lane) { // `lane` wave-uniform, not constant
    return bcast_lane(v, __builtin_amdgcn_readfirstlane(lane));
}

constexpr int TW_CONST = 20; // = TW (declared below): half window of the fast path

// Correctly rounded a/d for several numerators sharing one small positive integer-valued denominator.
struct SharedDiv {
    double d, y;
    __device__ __forceinline__ explicit SharedDiv(double den) : d(den) {
        const double y0 = __builtin_amdgcn_rcp(den);
        const double e0 = __builtin_fma(-den, y0, 1.0);
        const double y1 = __builtin_fma(y0, e0, y0);
        const double e1 = __builtin_fma(-den, y1, 1.0);
        y = __builtin_fma(y1, e1, y1);
    }
    __device__ __forceinline__ double operator()(double a) const {
        const double q0 = a * y;
        const double r = __builtin_fma(-d, q0, a);
        return __builtin_fma(r, y, q0);
    }
    __device__ __forceinline__ SharedDiv(double den, double recip) : d(den), y(recip) {} // from a KbDivTab entry
    // every later quotient becomes NaN unless ok (one select instead of one per quotient)
    __device__ __forceinline__ void poison_unless(bool ok) { y = ok ? y : __builtin_nan(""); }
};

// The denominators of the two window levels at half width 20 take few values: the refined reciprocals that
// SharedDiv's constructor computes (1 transcendental + 4 fma, after up to 13 integer operations for the
// second-level denominator) are tabulated once per context by the same instructions and fetched with one load.
//   first[c]          c = 1 .. 41 residues under the window (0: unused)
//   second[ml*41+mr]  ml = max(0, 40 - i), mr = max(0, 40 - (n-1-i)): den = 41 + (820 - ml(ml+1)/2) + (820 - mr(mr+1)/2)
//   second[41*41]     = (1, NaN): positions where the second smoothing is undefined
//   second[41*41+1]   = (1, 1): the single position of a one-residue protein (w = 0)
struct KbDivTab {
    double2 first[2 * TW_CONST + 2];
    double2 second[(2 * TW_CONST + 1) * (2 * TW_CONST + 1) + 2];
};
__global__ void k_build_divtab(KbDivTab *t) {
    constexpr int W = TW_CONST, M = 2 * W + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M + 1) {
        const SharedDiv dv((double)(i > 0 ? i : 1));
        t->first[i] = make_double2(dv.d, dv.y);
    }
    if (i < M * M) {
        const int ml = i / M, mr = i % M;
        const int den = M + (M * W - ((ml * (ml + 1)) >> 1)) + (M * W - ((mr * (mr + 1)) >> 1));
        const SharedDiv dv((double)den);
        t->second[i] = make_double2(dv.d, dv.y);
    }
    if (i == M * M) t->second[i] = make_double2(1.0, __builtin_nan(""));
    if (i == M * M + 1) t->second[i] = make_double2(1.0, SharedDiv(1.0).y); // n = 1: the value itself
}

// The B 41-term sums of positions I .. I+B-1 (I = 0 mod B) for THREE tracks at once, each in increasing position
// order: s[track][b] over I+b-20 .. I+b+20. Value e of the union (e = 0 .. 40+B-1, position I-20+e) is class
// (e + C0) % B, slot base + (e + C0) / B, where base = slot of position I-20-C0. `src.get(v, cls, slot)` delivers
// the three tracks' values of one position (from the weighted rings, or from the residue-code ring + tables).
template <int B, class Src>
__device__ __forceinline__ void window_sums3(const Src &src, int base, double (&s)[3][B]) {
    using G = KbGeom<B>;
    constexpr int NE = 2 * TW + B; // values in the union
    // value index e lies in window b  <=>  b <= e <= b + 40. `e` is always a compile-time constant after
    // unrolling (class and slot offset become immediates); `shift` moves the slot by whole body iterations.
    auto val = [&](double (&v)[3], int e, int shift = 0) {
        const int ec = e + G::C0;
        src.get(v, ec % B, base + ec / B + shift);
    };
    double acc[3][B];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[a][b] = 0.0;
    // head: e = 0 .. B-2 (not yet in every window)
#pragma unroll
    for (int e = 0; e < B - 1; ++e) {
        double v[3];
        val(v, e);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (b <= e) acc[a][b] = acc[a][b] + v[a];
    }
    // body: e = B-1 .. 40 lies in all B windows; partially unrolled on purpose (a fully unrolled body lets the
    // scheduler hoist every LDS read and spill)
    constexpr int NBODY = 2 * TW + 1 - (B - 1); // 42 - B values
    constexpr int STEP = B == 3 ? 3 : 4;
    static_assert(STEP % B == 0, "a body iteration must advance every class by whole slots");
#pragma unroll 2
    for (int it = 0; it < NBODY / STEP; ++it) {
        const int shift = it * (STEP / B);
        double v[STEP][3];
#pragma unroll
        for (int u = 0; u < STEP; ++u) val(v[u], (B - 1) + u, shift);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int u = 0; u < STEP; ++u) acc[a][b] = acc[a][b] + v[u][a];
    }
#pragma unroll
    for (int e = (B - 1) + (NBODY / STEP) * STEP; e <= 2 * TW; ++e) { // body remainder
        double v[3];
        val(v, e);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[a][b] = acc[a][b] + v[a];
    }
    // tail: e = 41 .. 40+B-1 (already past the first windows)
#pragma unroll
    for (int e = 2 * TW + 1; e < NE; ++e) {
        double v[3];
        val(v, e);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (e <= b + 2 * TW) acc[a][b] = acc[a][b] + v[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b) s[a][b] = acc[a][b];
}

// K-B blocks are ONE wave: a wave's LDS operations complete in program order, so stage hand-overs through LDS
// need no s_barrier and no counter drain, only a point the compiler does not move LDS accesses across
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int KB_PROTEINS_PER_BLOCK = 32; // blocks retire regularly, so the K-A kernels' blocks keep getting slots

// residue codes as the first-level windows see them: 0..21 real, 22 = any other byte (scored as X),
// 23 = a proline that PAPA skips (second P of PP / PxP: hydropathy and LLR of P, log-odds 0), 24 = no residue
// (outside the protein: every table holds +0.0, which leaves the fixed-order sums unchanged)
constexpr int KC_DUP = 23, KC_NONE = 24, KC_ROWS = 25;
constexpr int PRE_MIRROR = 48; // >= 2 * TW + 1 + 3
typedef double kb_d2 __attribute__((ext_vector_type(2)));
// Window-code tables, 16-byte rows: (hyd, llr) is ONE aligned 16-byte LDS read, lod sits at the same row offset of
// a second table. 16-byte rows put 16 different codes on 16 different bank quads (only codes c and c+16 collide).
constexpr uint32_t KB_ROW_BYTES = 16;
struct KbShared {
    alignas(16) kb_d2 t_hl[KC_ROWS];  // (hydropathy, llr)
    alignas(16) kb_d2 t_lod[KC_ROWS]; // (PAPA log-odds, unused)
    alignas(16) double ring[3 * RING_DOUBLES]; // weight * first-level FoldIndex / llr / papa
    // pre[q & 511] = charge sum of positions < q; the first PRE_MIRROR entries are repeated behind the end so that the
    // stream form can read pre[base + constant] without wrapping every index
    int pre[512 + PRE_MIRROR];
    int t_chg[KC_ROWS];
    // window-code ring (same class/slot geometry as the value rings); an entry is the code's BYTE OFFSET into t_hl,
    // so a lookup is one 16-bit read and two table reads without any address arithmetic
    uint16_t cring[RING_DOUBLES + 32];
};

// value sources of window_sums3
template <int B>
struct SrcCodes { // first level: residue code -> three table values
    const KbShared &S;
    __device__ __forceinline__ void get(double (&v)[3], int cls, int slot) const {
        const uint32_t o = S.cring[cls * KbGeom<B>::SUB + slot];
        const kb_d2 hl = *reinterpret_cast<const kb_d2 *>(
            __builtin_assume_aligned(reinterpret_cast<const char *>(S.t_hl) + o, 16));
        v[0] = hl.x;
        v[1] = hl.y;
        v[2] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(S.t_lod) + o);
    }
};
template <int B>
struct SrcRings { // second level: the three weighted rings
    const double *__restrict__ R;
    __device__ __forceinline__ void get(double (&v)[3], int cls, int slot) const {
        v[0] = R[0 * RING_DOUBLES + cls * KbGeom<B>::SUB + slot];
        v[1] = R[1 * RING_DOUBLES + cls * KbGeom<B>::SUB + slot];
        v[2] = R[2 * RING_DOUBLES + cls * KbGeom<B>::SUB + slot];
    }
};

struct KbConst {
    const uint8_t *codes, *cend;
    int ww1, ww2;
    bool adjust;
    double cc0, cc1, cc2;
};

// one protein with B positions per lane. `nolag` (n < 64*B): the whole protein fits one iteration, so the three
// stages run back to back over the same positions instead of as a lagged pipeline (no fill/drain iteration).
template <int B, bool TRACKS>
__device__ __forceinline__ void tracks20_protein(KbShared &S, const KbConst &K, const uint8_t *__restrict__ x,
                                                 uint64_t off, int n, bool nolag, plaac_row *__restrict__ row,
                                                 TrackPtrs tr) {
    using G = KbGeom<B>;
    const int lag1 = nolag ? 0 : G::LAG1, lag2 = nolag ? 0 : G::LAG2;
    const int lane = threadIdx.x;
    double *__restrict__ ring = S.ring;
    int *__restrict__ pre = S.pre;
    for (int i = lane; i < 3 * RING_DOUBLES; i += 64) ring[i] = 0.0;
    for (int i = lane; i < RING_DOUBLES + 32; i += 64) S.cring[i] = (uint16_t)(KC_NONE * KB_ROW_BYTES);
    for (int i = lane; i < 512; i += 64) pre[i] = 0;

    const int we = n - 1 < TW ? n - 1 : TW; // w = ww/2 clamped to n-1 (:2588-2589)
    int halfw = (K.ww1 - 1) / 2;            // FoldIndex run scan domain (:5010-5013)
    if (halfw > n / 2) halfw = n / 2;
    const int dlo = halfw, dhi = n - halfw - 1;
    const int plo = (K.ww2 - 1) / 2, phi = n - (K.ww2 - 1) / 2; // PAPA centres k in [plo, phi) (:4942)
    const double cc0 = K.cc0, cc1 = K.cc1, cc2 = K.cc2;

    double pbest = -INFINITY, pfi = 0.0, pll2 = 0.0;
    int pcen = -1;
    int numaa = 0, maxlen = 0, carry = 0;   // numaa / maxlen: per-lane partials, reduced at the end
    int last_zero = -1, last_flag = 0;      // wave-uniform carries of the FoldIndex run scan (position -1 is
                                            // unflagged: a run can start at 0 at the earliest)
    wave_sync();

    const int nchunks = (n + lag2 + G::C - 1) / G::C;
    // ring slot of entry number e (= position / B): e mod RB; per iteration the entry numbers advance by 64
    auto wrap = [](int s) { return s >= G::RB ? s - G::RB : s; };
    auto neg_slot = [](int e) { return ((e % G::RB) + G::RB) % G::RB; };
    int slot_in = 0;                                             // entry 64k            (stage 0 writes)
    int slot_l1 = neg_slot(-(lag1 + TW + G::C0) / B);            // entry of position 64Bk - LAG1 - 20 - C0
    int slot_w1 = neg_slot(-lag1 / B);                           // entry of position 64Bk - LAG1 (stage 1 writes)
    int slot_l2 = neg_slot(-(lag2 + TW + G::C0) / B);            // entry of position 64Bk - LAG2 - 20 - C0
    for (int k = 0; k < nchunks; ++k) {
        // ---- stage 0: residues -> mapped inputs at q0 = 64Bk + B*lane .. q0 + B-1
        {
            const int q0 = G::C * k + B * lane;
            uint32_t kc[B]; // window codes of the lane's B positions
            int ch[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                kc[j] = (uint32_t)KC_NONE;
                ch[j] = 0;
            }
            if (q0 < n) {
                uint32_t cb[B + 2]; // residues q0-2 .. q0+B-1
                const uint32_t wa = load4(x + q0 - 2, K.codes, K.cend);
                cb[0] = q0 >= 2 ? (wa & 0xffu) : 255u;
                cb[1] = q0 >= 1 ? ((wa >> 8) & 0xffu) : 255u;
                cb[2] = (wa >> 16) & 0xffu;
                cb[3] = wa >> 24;
                if (B > 2) {
                    const uint32_t wb = load4(x + q0 + 2, K.codes, K.cend);
#pragma unroll
                    for (int m = 4; m < B + 2; ++m) cb[m] = (wb >> (8 * (m - 4))) & 0xffu;
                }
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const bool in = q0 + j < n;
                    const uint32_t c = cb[2 + j] < 22u ? cb[2 + j] : 22u;
                    // only the first P of PP / PxP scores (:2653-2654); absolute neighbours p-1, p-2
                    const bool dup = K.adjust && c == 13u && (cb[1 + j] == 13u || cb[j] == 13u);
                    kc[j] = in ? (dup ? (uint32_t)KC_DUP : c) : (uint32_t)KC_NONE;
                    ch[j] = in ? S.t_chg[c] : 0;
                }
            }
            const int idx = wrap(slot_in + lane);
#pragma unroll
            for (int j = 0; j < B; ++j) S.cring[j * G::SUB + idx] = (uint16_t)(kc[j] * KB_ROW_BYTES);
            if (idx < G::MIR) { // mirrored head
#pragma unroll
                for (int j = 0; j < B; ++j)
                    S.cring[j * G::SUB + idx + G::RB] = (uint16_t)(kc[j] * KB_ROW_BYTES);
            }
            // charge prefix counts: inclusive wave scan of the per-lane sums
            int lsum = 0;
#pragma unroll
            for (int j = 0; j < B; ++j) lsum += ch[j];
            int s = lsum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(s, d);
                if (lane >= d) s += o;
            }
            int run = carry + s - lsum;
#pragma unroll
            for (int j = 0; j < B; ++j) {
                run += ch[j];
                pre[(q0 + j + 1) & 511] = run;
            }
            carry += bcast_lane(s, 63);
        }
        wave_sync();
        // ---- stage 1: first-level tracks at i0 = 64Bk + B*lane - LAG1 .. i0 + B-1
        {
            const int i0 = G::C * k + B * lane - lag1;
            double sums[3][B];
            window_sums3<B>(SrcCodes<B>{S}, wrap(slot_l1 + lane), sums);
            double wfi[B], wll[B], wpa[B];
            int zpos[B]; // position if FoldIndex >= 0 there (or outside the scan domain), else "none"
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int i = i0 + j;
                const bool live = i >= 0 && i < n;
                const int lo = imax(i - TW, 0), hi = imin(i + TW, n - 1);
                const SharedDiv div(live ? (double)(hi - lo + 1) : 1.0);
                const int csum = pre[(hi + 1) & 511] - pre[lo & 511];
                const double hydro = div(sums[0][j]);
                const double charge = div((double)csum);
                const double fi = (cc0 * hydro + cc1 * fabs(charge)) + cc2; // axpbypc (:2050)
                const double llr1 = div(sums[1][j]);
                const double papa = div(sums[2][j]);
                // weight 0 outside the protein: +-0.0 in the ring leaves every fixed-order sum unchanged
                const double wt = (double)(live ? 1 + imin(i, we) + imin(n - i - 1, we) : 0);
                wfi[j] = wt * fi;
                wll[j] = wt * llr1;
                wpa[j] = wt * papa;
                const bool neg = live && (fi < 0.0) && i >= dlo && i <= dhi;
                zpos[j] = neg ? INT_MIN : i;
                if (TRACKS && live) {
                    tr.charge[off + i] = charge;
                    tr.hydro[off + i] = hydro;
                    tr.fi[off + i] = fi;
                    tr.plaacllr[off + i] = llr1;
                    tr.papa[off + i] = papa;
                }
            }
            const int idx = wrap(slot_w1 + lane);
#pragma unroll
            for (int j = 0; j < B; ++j) {
                ring[RG_WF * RING_DOUBLES + j * G::SUB + idx] = wfi[j];
                ring[RG_WL * RING_DOUBLES + j * G::SUB + idx] = wll[j];
                ring[RG_WP * RING_DOUBLES + j * G::SUB + idx] = wpa[j];
            }
            if (idx < G::MIR) {
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    ring[RG_WF * RING_DOUBLES + j * G::SUB + idx + G::RB] = wfi[j];
                    ring[RG_WL * RING_DOUBLES + j * G::SUB + idx + G::RB] = wll[j];
                    ring[RG_WP * RING_DOUBLES + j * G::SUB + idx + G::RB] = wpa[j];
                }
            }
            // FoldIndex<0 runs (:5020-5058), position-parallel: a run is accounted where it ENDS, i.e. at a
            // position q without the flag whose predecessor has it; its start is one past the last unflagged
            // position before q (running max of unflagged positions: in-lane, wave max-scan, carry).
            {
                int lanemax = zpos[0];
#pragma unroll
                for (int j = 1; j < B; ++j) lanemax = imax(lanemax, zpos[j]);
                int sc = lanemax;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(sc, d);
                    if (lane >= d) sc = imax(sc, o);
                }
                int before = __shfl_up(sc, 1); // last unflagged position before this lane's first position
                if (lane == 0) before = INT_MIN;
                before = imax(before, last_zero);
                int prevflag = __shfl_up(zpos[B - 1] == INT_MIN ? 1 : 0, 1);
                if (lane == 0) prevflag = last_flag;
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const bool flagged = zpos[j] == INT_MIN;
                    // a run [before+1, q-1] just ended here iff this position is unflagged and its predecessor was
                    int rs = before + 1, re = i0 + j - 1;
                    rs = rs == dlo ? 0 : rs;
                    re = re == dhi ? n - 1 : re;
                    const int len = re - rs + 1;
                    const int cnt = (!flagged && prevflag != 0 && len >= 5) ? len : 0;
                    numaa += cnt;
                    maxlen = imax(maxlen, cnt);
                    before = imax(before, zpos[j]);
                    prevflag = flagged ? 1 : 0;
                }
                last_zero = imax(last_zero, bcast_lane(sc, 63));
                last_flag = bcast_lane(prevflag, 63);
            }
        }
        wave_sync();
        // ---- stage 2: weighted second smoothing at i0 = 64Bk + B*lane - LAG2 .. i0 + B-1, PAPA arg-max
        {
            const int i0 = G::C * k + B * lane - lag2;
            double sums[3][B];
            window_sums3<B>(SrcRings<B>{ring}, wrap(slot_l2 + lane), sums);
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int i = i0 + j;
                const bool valid = i >= we && i <= n - we - 1; // implies 0 <= i < n
                const int den = (2 * we + 1) + window_weight_side(i, we) + window_weight_side(n - 1 - i, we);
                SharedDiv div(valid ? (double)den : 1.0);
                div.poison_unless(valid); // NaN outside [w, n-w-1] (:2597-2600)
                const double pax2 = div(sums[2][j]);
                if (TRACKS && i >= 0 && i < n) {
                    tr.fix2[off + i] = div(sums[0][j]);
                    tr.plaacllrx2[off + i] = div(sums[1][j]);
                    tr.papax2[off + i] = pax2;
                }
                // papamode 1 (:4942-4948): NaNs fail the comparison. fix2 < 0 <=> its numerator < 0 (the positive
                // denominator is at most 1681, a quotient cannot underflow to zero); the two other quotients of the
                // winning position are formed once, after the scan, from the kept numerators.
                const bool upd = i >= plo && i < phi && (pax2 > pbest) && (sums[0][j] < 0.0);
                pbest = upd ? pax2 : pbest;
                pcen = upd ? i : pcen;
                pfi = upd ? sums[0][j] : pfi;
                pll2 = upd ? sums[1][j] : pll2;
            }
        }
        slot_in = wrap(slot_in + 64);
        slot_l1 = wrap(slot_l1 + 64);
        slot_w1 = wrap(slot_w1 + 64);
        slot_l2 = wrap(slot_l2 + 64);
    }
    // wave arg-max: largest papax2, smallest centre among equals (first max of the serial loop); the lane that
    // owns the winning centre then hands over the numerators it kept
    {
        const int mine = pcen;
        for (int d = 32; d >= 1; d >>= 1) {
            const double ob = __shfl_xor(pbest, d);
            const int oc = __shfl_xor(pcen, d);
            const bool take = (oc >= 0) && (pcen < 0 || ob > pbest || (ob == pbest && oc < pcen));
            pbest = take ? ob : pbest;
            pcen = take ? oc : pcen;
        }
        if (pcen >= 0) { // wave-uniform
            const int src = __builtin_ctzll(__ballot(mine == pcen));
            const double s0 = bcast_lane(pfi, src), s1 = bcast_lane(pll2, src);
            const int den = (2 * we + 1) + window_weight_side(pcen, we) + window_weight_side(n - 1 - pcen, we);
            const SharedDiv div((double)den);
            pfi = div(s0);
            pll2 = div(s1);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { // FoldIndex run statistics: sum / max over lanes
        numaa += __shfl_xor(numaa, d);
        maxlen = imax(maxlen, __shfl_xor(maxlen, d));
    }
    // PAPAllr = first-level PLAAC-LLR at the centre, recomputed: lanes fetch the 41 taps in parallel and park them
    // in LDS (the rings are free now), the fixed-order sum reads them back as broadcasts (out-of-range taps add +0.0)
    double papallr = __builtin_nan("");
    if (pcen >= 0) {
        const int q = pcen - TW + lane;
        wave_sync();
        ring[lane] = (lane <= 2 * TW && q >= 0 && q < n) ? S.t_hl[ld_code(x, (uint32_t)q)].y : 0.0;
        wave_sync();
        double s = 0.0;
#pragma unroll
        for (int j = 0; j <= 2 * TW; ++j) s = s + ring[j];
        const int lo = imax(pcen - TW, 0), hi = imin(pcen + TW, n - 1);
        papallr = s / (double)(hi - lo + 1);
    }
    if (lane == 0) {
        row->fi_numaa = numaa;
        row->fi_maxrun = maxlen;
        row->papa_cen = pcen;
        if (pcen >= 0) {
            row->papa_combo = pbest;
            row->papa_prop = pbest;
            row->papa_fi = pfi;
            row->papa_llr = papallr;
            row->papa_llr2 = pll2;
        } else {
            row->papa_combo = -INFINITY;
            row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = __builtin_nan("");
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K-B, stream form: the 16 proteins of a block are laid end to end on ONE position axis (segment k starts at a
// multiple of 4, S[k+1] = S[k] + n_k + gap, gap = 20..23 positions without residues), and the three pipelined
// stages of tracks20_protein<4> run over that stream instead of over one protein at a time:
//  * no fill / drain iteration and no partly empty last iteration per protein (UniRef-shaped lengths: 83 % ->
//    90 % of the position slots carry a residue), and every iteration has 4 positions per lane, the variant with
//    the fewest LDS reads per add (the per-protein kernel needs 2 or 3 for short proteins);
//  * rings, prefix counts and code ring are initialised once per block, not once per protein.
// A gap of >= 20 empty positions (window code "none", ring value 0.0, charge 0) isolates neighbours exactly: every
// window of a position reaches at most 20 positions beyond its protein. What was wave-uniform per protein (n, the
// clamped half width, the scan domains) becomes per lane: a lane's 4 positions always lie in one segment. Per-protein
// results: FoldIndex run statistics go to LDS accumulators (a run is accounted at the position where it ends, which
// lies in the same segment); the PAPA arg-max keeps one candidate per lane for the lane's current segment and one for
// its previous segment, and a protein is finalised (wave arg-max over the lanes tagged with it, PAPAllr, row write)
// in the iteration in which stage 2 passes its last residue - by then no lane has moved on by more than one segment.
// ------------------------------------------------------------------------------------------------
// Wave-wide inclusive scans and a one-lane shift in DPP form (GFX9 row_shr / row_bcast / wave_shr modifiers): six
// data-parallel moves instead of six ds_bpermute round trips through the LDS crossbar with their address arithmetic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_scan_add(int s) {
    s += dpp_move<0x111, 0xf>(0, s); // row_shr:1
    s += dpp_move<0x112, 0xf>(0, s); // row_shr:2
    s += dpp_move<0x114, 0xf>(0, s); // row_shr:4
    s += dpp_move<0x118, 0xf>(0, s); // row_shr:8  -> inclusive within each row of 16
    s += dpp_move<0x142, 0xa>(0, s); // row_bcast:15 into rows 1 and 3
    s += dpp_move<0x143, 0xc>(0, s); // row_bcast:31 into rows 2 and 3
    return s;
}
__device__ __forceinline__ int wave_scan_max(int s) {
    s = imax(s, dpp_move<0x111, 0xf>(INT_MIN, s));
    s = imax(s, dpp_move<0x112, 0xf>(INT_MIN, s));
    s = imax(s, dpp_move<0x114, 0xf>(INT_MIN, s));
    s = imax(s, dpp_move<0x118, 0xf>(INT_MIN, s));
    s = imax(s, dpp_move<0x142, 0xa>(INT_MIN, s));
    s = imax(s, dpp_move<0x143, 0xc>(INT_MIN, s));
    return s;
}
__device__ __forceinline__ int wave_shr1(int fill, int v) { return dpp_move<0x138, 0xf>(fill, v); } // lane l <- lane l-1
// wave-wide maximum of a double / minimum of an int, returned wave-uniform (lane 63 of the scan)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double old, double v) {
    const long long o = __builtin_bit_cast(long long, old), u = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)dpp_move<CTRL, ROW_MASK>((int)(unsigned)(o & 0xffffffffll), (int)(unsigned)(u & 0xffffffffll));
    const unsigned hi = (unsigned)dpp_move<CTRL, ROW_MASK>((int)(unsigned)((unsigned long long)o >> 32),
                                                            (int)(unsigned)((unsigned long long)u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_max_f64(double s) { // values are never NaN here
    const double ninf = -INFINITY;
    s = __builtin_fmax(s, dpp_move_f64<0x111, 0xf>(ninf, s));
    s = __builtin_fmax(s, dpp_move_f64<0x112, 0xf>(ninf, s));
    s = __builtin_fmax(s, dpp_move_f64<0x114, 0xf>(ninf, s));
    s = __builtin_fmax(s, dpp_move_f64<0x118, 0xf>(ninf, s));
    s = __builtin_fmax(s, dpp_move_f64<0x142, 0xa>(ninf, s));
    s = __builtin_fmax(s, dpp_move_f64<0x143, 0xc>(ninf, s));
    return bcast_lane(s, 63);
}
__device__ __forceinline__ float wave_max_f32(float s) { // values are never NaN here; returned wave-uniform
    const int ninf = (int)0xff800000u;
    auto step = [&](float v, int moved) { return __builtin_fmaxf(v, __builtin_bit_cast(float, moved)); };
    s = step(s, dpp_move<0x111, 0xf>(ninf, __builtin_bit_cast(int, s)));
    s = step(s, dpp_move<0x112, 0xf>(ninf, __builtin_bit_cast(int, s)));
    s = step(s, dpp_move<0x114, 0xf>(ninf, __builtin_bit_cast(int, s)));
    s = step(s, dpp_move<0x118, 0xf>(ninf, __builtin_bit_cast(int, s)));
    s = step(s, dpp_move<0x142, 0xa>(ninf, __builtin_bit_cast(int, s)));
    s = step(s, dpp_move<0x143, 0xc>(ninf, __builtin_bit_cast(int, s)));
    return __builtin_bit_cast(float, bcast_lane(__builtin_bit_cast(int, s), 63));
}
__device__ __forceinline__ int wave_min_i32(int s) {
    s = imin(s, dpp_move<0x111, 0xf>(INT_MAX, s));
    s = imin(s, dpp_move<0x112, 0xf>(INT_MAX, s));
    s = imin(s, dpp_move<0x114, 0xf>(INT_MAX, s));
    s = imin(s, dpp_move<0x118, 0xf>(INT_MAX, s));
    s = imin(s, dpp_move<0x142, 0xa>(INT_MAX, s));
    s = imin(s, dpp_move<0x143, 0xc>(INT_MAX, s));
    return bcast_lane(s, 63);
}

constexpr int KS_GAP = TW; // empty positions after every protein (rounded up so that segments start at multiples of 4)
struct KsShared {
    KbShared kb;
    int segS[KB_PROTEINS_PER_BLOCK + 1]; // stream start of every segment; [16] = end of the stream
    int segN[KB_PROTEINS_PER_BLOCK];
    uint32_t segP[KB_PROTEINS_PER_BLOCK];
    uint64_t segOff[KB_PROTEINS_PER_BLOCK];
    int acc_numaa[KB_PROTEINS_PER_BLOCK], acc_maxlen[KB_PROTEINS_PER_BLOCK];
    double bc[48]; // the 41 taps of the PAPAllr recompute
};

struct KsCand { // PAPA arg-max candidate of one lane for one segment
    int tag, cen;
    double best, s0, s1;
};

template <bool TRACKS>
__global__ __launch_bounds__(64) void k_tracks20s(const uint8_t *__restrict__ codes, const uint4 *__restrict__ order,
                                                  uint32_t nprot, uint64_t total, const DevTables *__restrict__ T,
                                                  const KbDivTab *__restrict__ DT, plaac_row *__restrict__ rows,
                                                  TrackPtrs tr, const uint32_t *__restrict__ huge) {
    static_assert(TW_CONST == TW, "KbDivTab is built for the fast path's half window");
    if (*huge) return; // a protein too long for the int32 stream axis: k_tracks20 scores this batch
    constexpr int B = 4;
    using G = KbGeom<B>;
    constexpr int NP = KB_PROTEINS_PER_BLOCK;
    __shared__ KsShared Z;
    KbShared &S = Z.kb;
    const int lane = threadIdx.x;
    double *__restrict__ ring = S.ring;
    int *__restrict__ pre = S.pre;
    if (lane < KC_ROWS) {
        const int k = lane < NAA ? lane : (lane == KC_DUP ? 13 : 0); // 22 -> X, 23 -> P
        const bool none = lane == KC_NONE;
        S.t_hl[lane].x = none ? 0.0 : T->hyd[k];
        S.t_hl[lane].y = none ? 0.0 : T->llr[k];
        S.t_lod[lane].x = (none || lane == KC_DUP) ? 0.0 : T->lod[k];
        S.t_lod[lane].y = 0.0;
        S.t_chg[lane] = none ? 0 : T->chg[k];
    }
    for (int i = lane; i < 3 * RING_DOUBLES; i += 64) ring[i] = 0.0;
    for (int i = lane; i < RING_DOUBLES + 32; i += 64) S.cring[i] = (uint16_t)(KC_NONE * KB_ROW_BYTES);
    for (int i = lane; i < 512 + PRE_MIRROR; i += 64) pre[i] = 0;
    // ---- segment table: protein k of this block is plan item blockIdx.x + k * gridDim.x (every block gets the same
    //      mix of long and short proteins of the descending-length plan)
    {
        uint4 it = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t idx = blockIdx.x + (uint32_t)lane * gridDim.x;
        const bool have = lane < NP && idx < nprot;
        if (have) it = order[idx];
        const int n = have ? (int)it.z : 0;
        const int span = n > 0 ? ((n + KS_GAP + 3) & ~3) : 0;
        int incl = span;
#pragma unroll
        for (int d = 1; d < NP; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane < NP) {
            Z.segS[lane] = incl - span;
            Z.segN[lane] = n;
            Z.segP[lane] = it.w;
            Z.segOff[lane] = ((uint64_t)it.y << 32) | it.x;
            Z.acc_numaa[lane] = 0;
            Z.acc_maxlen[lane] = 0;
            if (lane == NP - 1) Z.segS[NP] = incl;
            if (have && n == 0) { // skipped record (:762): zero the fields this kernel owns
                plaac_row *row = rows + it.w;
                row->papa_combo = row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = 0.0;
                row->fi_numaa = row->fi_maxrun = row->papa_cen = 0;
            }
        }
    }
    wave_sync();
    const int stream_end = Z.segS[NP]; // wave-uniform
    if (stream_end == 0) return;
    const uint8_t *cend = codes + total;
    const int ww1 = T->ww1, ww2 = T->ww2;
    const bool adjust = T->adjustprolines != 0;
    const double cc0 = T->cc[0], cc1 = T->cc[1], cc2 = T->cc[2];
    const int plo = (ww2 - 1) / 2;

    // segment of the 4 positions starting at stream position s (s may be negative: segment 0); `cur` = wave-uniform
    // segment of the iteration's first position, `last` = stream position of the iteration's last lane
    auto seg_of = [&](int s, int cur, int last) {
        int k = cur;
        for (int kk = cur + 1; kk < NP; ++kk) { // wave-uniform trip count: segments that start inside the iteration
            const int sk = __builtin_amdgcn_readfirstlane(Z.segS[kk]);
            if (sk > last) break;
            k = s >= sk ? kk : k;
        }
        return k;
    };
    auto advance = [&](int cur, int first) { // largest k with segS[k] <= first
        while (cur + 1 < NP && __builtin_amdgcn_readfirstlane(Z.segS[cur + 1]) <= first) ++cur;
        return cur;
    };

    KsCand cur{-1, -1, -INFINITY, 0.0, 0.0}, prv{-1, -1, -INFINITY, 0.0, 0.0};
    int carry = 0, last_zero = -1, last_flag = 0;
    int c0 = 0, c1 = 0, c2 = 0, fin = 0; // segment cursors of the three stages; next protein to finalise
    auto wrap = [](int s) { return s >= G::RB ? s - G::RB : s; };
    auto neg_slot = [](int e) { return ((e % G::RB) + G::RB) % G::RB; };
    int slot_in = 0;
    int slot_l1 = neg_slot(-(G::LAG1 + TW + G::C0) / B);
    int slot_w1 = neg_slot(-G::LAG1 / B);
    int slot_l2 = neg_slot(-(G::LAG2 + TW + G::C0) / B);
    // the last residue sits at most at stream_end - KS_GAP - 1; stage 2 (lag LAG2) has to pass it
    const int nchunks = (stream_end - KS_GAP + G::LAG2 + G::C - 1) / G::C;
    for (int c = 0; c < nchunks; ++c) {
        // ---- stage 0: residues -> window codes, charge prefix counts
        {
            const int s = G::C * c + B * lane;
            c0 = advance(c0, G::C * c);
            const int k = seg_of(s, c0, G::C * c + G::C - 1);
            const int n = Z.segN[k], i0 = s - Z.segS[k];
            uint32_t kc[B];
            int ch[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                kc[j] = (uint32_t)KC_NONE;
                ch[j] = 0;
            }
            if (i0 < n) {
                const uint8_t *x = codes + Z.segOff[k];
                uint32_t cb[B + 2]; // residues i0-2 .. i0+3
                const uint32_t wa = load4(x + i0 - 2, codes, cend);
                cb[0] = i0 >= 2 ? (wa & 0xffu) : 255u;
                cb[1] = i0 >= 1 ? ((wa >> 8) & 0xffu) : 255u;
                cb[2] = (wa >> 16) & 0xffu;
                cb[3] = wa >> 24;
                const uint32_t wb = load4(x + i0 + 2, codes, cend);
#pragma unroll
                for (int m = 4; m < B + 2; ++m) cb[m] = (wb >> (8 * (m - 4))) & 0xffu;
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const bool in = i0 + j < n;
                    const uint32_t cd = cb[2 + j] < 22u ? cb[2 + j] : 22u;
                    const bool dup = adjust && cd == 13u && (cb[1 + j] == 13u || cb[j] == 13u); // (:2653-2654)
                    kc[j] = in ? (dup ? (uint32_t)KC_DUP : cd) : (uint32_t)KC_NONE;
                    ch[j] = in ? S.t_chg[cd] : 0;
                }
            }
            const int idx = wrap(slot_in + lane);
#pragma unroll
            for (int j = 0; j < B; ++j) S.cring[j * G::SUB + idx] = (uint16_t)(kc[j] * KB_ROW_BYTES);
            if (idx < G::MIR) {
#pragma unroll
                for (int j = 0; j < B; ++j) S.cring[j * G::SUB + idx + G::RB] = (uint16_t)(kc[j] * KB_ROW_BYTES);
            }
            int lsum = 0;
#pragma unroll
            for (int j = 0; j < B; ++j) lsum += ch[j];
            const int sc = wave_scan_add(lsum);
            int run = carry + sc - lsum;
#pragma unroll
            for (int j = 0; j < B; ++j) {
                run += ch[j];
                const int pi = (s + j + 1) & 511;
                pre[pi] = run;
                if (pi < PRE_MIRROR) pre[512 + pi] = run;
            }
            carry += bcast_lane(sc, 63);
        }
        wave_sync();
        // ---- stage 1: first-level tracks, LAG1 positions behind
        {
            const int s = G::C * c + B * lane - G::LAG1;
            c1 = advance(c1, G::C * c - G::LAG1);
            const int k = seg_of(s, c1, G::C * c + G::C - 1 - G::LAG1);
            const int n = Z.segN[k], sg = Z.segS[k], i0 = s - sg;
            const int we = n - 1 < TW ? n - 1 : TW; // (:2588-2589)
            int halfw = (ww1 - 1) / 2;              // (:5010-5013)
            halfw = halfw > n / 2 ? n / 2 : halfw;
            const int dlo = halfw, dhi = n - halfw - 1;
            double sums[3][B];
            window_sums3<B>(SrcCodes<B>{S}, wrap(slot_l1 + lane), sums);
            double wfi[B], wll[B], wpa[B];
            int zpos[B]; // STREAM position if FoldIndex >= 0 there (or outside the scan domain), else "none"
            // charge sum of a window = difference of two prefix counts, 41 positions apart on the stream: positions
            // outside the protein are empty for at least 20 positions on either side, so no clamping is needed
            const int *__restrict__ pw = pre + ((s - TW) & 511);
            uint64_t off = 0;
            if (TRACKS) off = Z.segOff[k];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int i = i0 + j;
                const bool live = i >= 0 && i < n;
                const int lo = imax(i - TW, 0), hi = imin(i + TW, n - 1);
                // The reciprocal is FETCHED AFTER the window sums (the empty asm ties the index to a sum): fetched
                // before them, the four table entries stay live across the sums and cost the registers that let a
                // fourth wave of another kernel share the SIMD.
                int c1 = live ? hi - lo + 1 : 1;
                asm volatile("" : "+v"(c1) : "v"(sums[0][j]));
                const double2 e1 = DT->first[c1];
                const SharedDiv div(e1.x, e1.y);
                const int csum = pw[j + 2 * TW + 1] - pw[j];
                const double hydro = div(sums[0][j]);
                const double charge = div((double)csum);
                const double fi = (cc0 * hydro + cc1 * fabs(charge)) + cc2; // axpbypc (:2050)
                const double llr1 = div(sums[1][j]);
                const double papa = div(sums[2][j]);
                const double wt = (double)(live ? 1 + imin(i, we) + imin(n - i - 1, we) : 0);
                wfi[j] = wt * fi;
                wll[j] = wt * llr1;
                wpa[j] = wt * papa;
                const bool neg = live && (fi < 0.0) && i >= dlo && i <= dhi;
                zpos[j] = neg ? INT_MIN : s + j;
                if (TRACKS && live) {
                    tr.charge[off + i] = charge;
                    tr.hydro[off + i] = hydro;
                    tr.fi[off + i] = fi;
                    tr.plaacllr[off + i] = llr1;
                    tr.papa[off + i] = papa;
                }
            }
            const int idx = wrap(slot_w1 + lane);
#pragma unroll
            for (int j = 0; j < B; ++j) {
                ring[RG_WF * RING_DOUBLES + j * G::SUB + idx] = wfi[j];
                ring[RG_WL * RING_DOUBLES + j * G::SUB + idx] = wll[j];
                ring[RG_WP * RING_DOUBLES + j * G::SUB + idx] = wpa[j];
            }
            if (idx < G::MIR) {
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    ring[RG_WF * RING_DOUBLES + j * G::SUB + idx + G::RB] = wfi[j];
                    ring[RG_WL * RING_DOUBLES + j * G::SUB + idx + G::RB] = wll[j];
                    ring[RG_WP * RING_DOUBLES + j * G::SUB + idx + G::RB] = wpa[j];
                }
            }
            // FoldIndex<0 runs (:5020-5058) as in tracks20_protein, on stream positions; gap positions are unflagged,
            // so no run crosses a segment, and a run ends (is accounted) inside its own segment
            {
                int lanemax = zpos[0];
#pragma unroll
                for (int j = 1; j < B; ++j) lanemax = imax(lanemax, zpos[j]);
                const int sc = wave_scan_max(lanemax);
                const int before0 = imax(wave_shr1(INT_MIN, sc), last_zero);
                int before = before0;
                int prevflag = wave_shr1(last_flag, zpos[B - 1] == INT_MIN ? 1 : 0);
                int numaa = 0, maxlen = 0;
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const bool flagged = zpos[j] == INT_MIN;
                    int rs = before + 1 - sg, re = i0 + j - 1; // the run that ends here, in protein coordinates
                    rs = rs == dlo ? 0 : rs;
                    re = re == dhi ? n - 1 : re;
                    const int len = re - rs + 1;
                    const int cnt = (!flagged && prevflag != 0 && len >= 5) ? len : 0;
                    numaa += cnt;
                    maxlen = imax(maxlen, cnt);
                    before = imax(before, zpos[j]);
                    prevflag = flagged ? 1 : 0;
                }
                if (numaa) {
                    atomicAdd(&Z.acc_numaa[k], numaa);
                    atomicMax(&Z.acc_maxlen[k], maxlen);
                }
                last_zero = imax(last_zero, bcast_lane(sc, 63));
                last_flag = bcast_lane(prevflag, 63);
            }
        }
        wave_sync();
        // ---- stage 2: weighted second smoothing, LAG2 positions behind; PAPA candidates
        {
            const int s = G::C * c + B * lane - G::LAG2;
            c2 = advance(c2, G::C * c - G::LAG2);
            const int k = seg_of(s, c2, G::C * c + G::C - 1 - G::LAG2);
            const int n = Z.segN[k], i0 = s - Z.segS[k];
            const int we = n - 1 < TW ? n - 1 : TW;
            const int phi = n - plo; // PAPA centres in [plo, phi) (:4942)
            if (k != cur.tag) { // this lane has moved on to another protein: keep the old candidate one more segment
                prv = cur;
                cur = KsCand{k, -1, -INFINITY, 0.0, 0.0};
            }
            double sums[3][B];
            window_sums3<B>(SrcRings<B>{ring}, wrap(slot_l2 + lane), sums);
            uint64_t off = 0;
            if (TRACKS) off = Z.segOff[k];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const int i = i0 + j;
                const bool valid = i >= we && i <= n - we - 1; // implies 0 <= i < n; NaN elsewhere (:2597-2600)
                // a protein of at most 20 residues (w clamped to n-1 < 20) has a valid position only for n = 1: i = 0,
                // w = 0, denominator 1 - the table's last entry
                const int ml = imax(0, 2 * TW - i), mr = imax(0, 2 * TW - (n - 1 - i));
                int c2 = valid ? (we == TW ? ml * (2 * TW + 1) + mr : (2 * TW + 1) * (2 * TW + 1) + 1)
                               : (2 * TW + 1) * (2 * TW + 1);
                asm volatile("" : "+v"(c2) : "v"(sums[2][j])); // fetch after the sums, see stage 1
                const double2 e2 = DT->second[c2];
                const SharedDiv div(e2.x, e2.y);
                const double pax2 = div(sums[2][j]);
                if (TRACKS && i >= 0 && i < n) {
                    tr.fix2[off + i] = div(sums[0][j]);
                    tr.plaacllrx2[off + i] = div(sums[1][j]);
                    tr.papax2[off + i] = pax2;
                }
                const bool upd = i >= plo && i < phi && (pax2 > cur.best) && (sums[0][j] < 0.0); // (:4942-4948)
                cur.best = upd ? pax2 : cur.best;
                cur.cen = upd ? i : cur.cen;
                cur.s0 = upd ? sums[0][j] : cur.s0;
                cur.s1 = upd ? sums[1][j] : cur.s1;
            }
        }
        // ---- proteins whose last residue stage 2 has passed in this iteration
        const int passed = G::C * c + G::C - 1 - G::LAG2;
        while (fin < NP) { // wave-uniform
            const int n = __builtin_amdgcn_readfirstlane(Z.segN[fin]);
            const int sg = __builtin_amdgcn_readfirstlane(Z.segS[fin]);
            if (n > 0 && sg + n - 1 > passed) break;
            if (n > 0) {
                const bool mc = cur.tag == fin, mp = prv.tag == fin;
                const double mbest = mc ? cur.best : (mp ? prv.best : -INFINITY);
                const int mine = mc ? cur.cen : (mp ? prv.cen : -1);
                const double ms0 = mc ? cur.s0 : prv.s0, ms1 = mc ? cur.s1 : prv.s1;
                // wave arg-max: the largest papax2 among the lanes that hold a candidate, then the smallest centre
                // among the lanes that hold that value (first maximum of the serial loop)
                const double pbest = wave_max_f64(mine >= 0 ? mbest : -INFINITY);
                const int cmin = wave_min_i32((mine >= 0 && mbest == pbest) ? mine : INT_MAX);
                const int pcen = cmin == INT_MAX ? -1 : cmin;
                const int we = n - 1 < TW ? n - 1 : TW;
                double pfi = 0.0, pll2 = 0.0, papallr = __builtin_nan("");
                if (pcen >= 0) { // wave-uniform
                    const int src = __builtin_ctzll(__ballot(mine == pcen));
                    const double s0 = bcast_lane_dyn(ms0, src), s1 = bcast_lane_dyn(ms1, src);
                    const int den = (2 * we + 1) + window_weight_side(pcen, we) + window_weight_side(n - 1 - pcen, we);
                    const SharedDiv div((double)den);
                    pfi = div(s0);
                    pll2 = div(s1);
                    // PAPAllr = first-level PLAAC-LLR at the centre, recomputed (41 taps through LDS broadcasts)
                    const uint8_t *x = codes + Z.segOff[fin];
                    const int q = pcen - TW + lane;
                    wave_sync();
                    if (lane <= 2 * TW) Z.bc[lane] = (q >= 0 && q < n) ? S.t_hl[ld_code(x, (uint32_t)q)].y : 0.0;
                    wave_sync();
                    double sm = 0.0;
#pragma unroll
                    for (int j = 0; j <= 2 * TW; ++j) sm = sm + Z.bc[j];
                    const int lo = imax(pcen - TW, 0), hi = imin(pcen + TW, n - 1);
                    papallr = sm / (double)(hi - lo + 1);
                }
                if (lane == 0) {
                    plaac_row *row = rows + Z.segP[fin];
                    row->fi_numaa = Z.acc_numaa[fin];
                    row->fi_maxrun = Z.acc_maxlen[fin];
                    row->papa_cen = pcen;
                    if (pcen >= 0) {
                        row->papa_combo = pbest;
                        row->papa_prop = pbest;
                        row->papa_fi = pfi;
                        row->papa_llr = papallr;
                        row->papa_llr2 = pll2;
                    } else {
                        row->papa_combo = -INFINITY;
                        row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = __builtin_nan("");
                    }
                }
            }
            ++fin;
        }
        slot_in = wrap(slot_in + 64);
        slot_l1 = wrap(slot_l1 + 64);
        slot_w1 = wrap(slot_w1 + 64);
        slot_l2 = wrap(slot_l2 + 64);
    }
}

// ------------------------------------------------------------------------------------------------
// K-B, summary mode: FILTER + EXACT REFINE  (k_tracks20f -> k_refine_centres -> k_tracks20 over a list)
//
// The summary row keeps no window track, only decisions made on them and four values at one position:
//   * the sign of FoldIndex at every position (run statistics, :5010-5059),
//   * which position has the largest doubly smoothed PAPA score among those with negative doubly smoothed
//     FoldIndex (:4932-4948),
//   * papax2 / fix2 / plaacllr / plaacllrx2 AT that position.
// A fixed-order 41-term sum cannot be shared between neighbouring positions, but a DECISION does not need the
// reference's bits, only a value with a rigorous error bound that excludes the other outcome. So the filter kernel
// computes every window sum as a difference of two running prefix sums (one fp64 wave scan per 448 positions instead
// of 41 adds per position), carries a bound on |approximate - reference| through both smoothing levels, and decides
// with it. Whatever it cannot decide with certainty (a FoldIndex within the bound of zero, two PAPA candidates
// within the bound of each other - exact ties are typical of perfect repeats) sends the WHOLE protein to the exact
// kernel (k_tracks20 over the fallback list), so correctness never rests on the filter, only speed does. The four
// values at the chosen centre are then computed in the reference's exact order by k_refine_centres (3 x 41 first-
// level windows + 3 second-level sums per protein instead of 6 x 41 adds per residue).
//
// Error bounds (u = 2^-53; every fp64 add/mul/fma rounds once; a summation tree of depth d over terms a_t has error
// <= d u sum|a_t| (1 + O(du))). P = running prefix over the block's stream, depth <= nch = iterations + 16 (in-lane
// chain 6, wave scan 7, carry chain <= iterations, two final adds), magnitude <= L*A (L = stream length, A = max
// |table value|):  E_P = nch * 2u * L * A   (2u instead of u: slack for the O(du) terms).
//   window sum   ~S = P[i+21] - P[i-20]:   |~S - S_ref| <= 2 E_P + 256u * 41 A           =: E_S   (reference: 40 u 41 A)
//   G = cc0 ~S_h + cc1 |C| + cc2 m  (= m * FoldIndex in real arithmetic; C, m exact integers):
//                 |G - m fi_ref| <= |cc0| E_S + 256u * 41 CC  =: E_G,   CC = |cc0| A_h + |cc1| + |cc2|
//                 so |G| > E_G fixes the sign of the reference's fi (m > 0), and wt*fi_ref = m fi_ref is G within 2 E_G
//   second level  ~T = Q[k+21] - Q[k-20] over f = G (resp. ~S_p):  E_T = 2 E_Q + 41 * 2 E_G + 256u * 1681 CC, with
//                 E_Q = nch * 2u * L * 41 CC (resp. A_p for CC);  fix2_ref < 0 <=> T_ref < 0 (positive denominator)
//   papax2        v = ~T_p / den, den >= 41:  E_v = E_Tp / 41 + 256u * A_p
// Candidates: positions whose fix2 is certainly negative; a unique one above (best - 2 E_v), no uncertain-fix2
// position above that mark either, is the reference's arg-max (strict >, first maximum).
// ------------------------------------------------------------------------------------------------
constexpr int FB = 7;                 // positions per lane
constexpr int FC = 64 * FB;           // positions per iteration (448)
constexpr int FLAG1 = TW + 1;         // stage 1 runs 21 positions behind stage 0: P[i+21] of a lane's positions are the
                                      // prefix values the lane itself produced in stage 0 (registers, no LDS read)
constexpr int FLAG2 = 2 * (TW + 1);   // stage 2 likewise 21 behind stage 1
constexpr int FRING = 511;            // ring slots = stream position mod 511 = 73 * FB: a lane's FB-aligned group of
                                      // slots never wraps, so every LDS access is base register + immediate
static_assert(FLAG1 % FB == 0 && FRING % FB == 0 && FC + 2 * FLAG1 <= FRING, "ring geometry");

// row_shr with bound_ctrl: lanes without a source lane read 0, no `old` operand to initialise
template <int CTRL>
__device__ __forceinline__ double dpp_shr0_f64(double v) {
    const long long u = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_scan_add_f64(double s) { // inclusive, lane order
    s = s + dpp_shr0_f64<0x111>(s);
    s = s + dpp_shr0_f64<0x112>(s);
    s = s + dpp_shr0_f64<0x114>(s);
    s = s + dpp_shr0_f64<0x118>(s);
    s = s + dpp_move_f64<0x142, 0xa>(0.0, s);
    s = s + dpp_move_f64<0x143, 0xc>(0.0, s);
    return s;
}
__device__ __forceinline__ double wave_shr1_f64(double fill, double v) { return dpp_move_f64<0x138, 0xf>(fill, v); }

struct KfShared {
    alignas(16) kb_d2 p1[FRING + 1]; // slot x-1: (P_h, P_p)[x], exclusive prefix sums of hydropathy / PAPA log-odds
    alignas(16) kb_d2 p2[FRING + 1]; // slot x-1: (Q_f, Q_p)[x], exclusive prefix sums of G (= m * FoldIndex) / PAPA sums
    int pre[FRING + 1];              // slot x-1: exclusive prefix count of the charge at x
    alignas(16) kb_d2 t_hp[KC_ROWS]; // (hydropathy, PAPA log-odds) by window code
    int t_chg[KC_ROWS];
    uint8_t lring[128];              // per lane slot (64 * iteration + lane) & 127: segment of the lane's positions
    int segS[KB_PROTEINS_PER_BLOCK + 1];
    int segN[KB_PROTEINS_PER_BLOCK];
    uint32_t segP[KB_PROTEINS_PER_BLOCK];
    uint64_t segOff[KB_PROTEINS_PER_BLOCK];
    int acc_numaa[KB_PROTEINS_PER_BLOCK], acc_maxlen[KB_PROTEINS_PER_BLOCK], seg_amb[KB_PROTEINS_PER_BLOCK];
    int res[KB_PROTEINS_PER_BLOCK]; // PAPA centre of a finished protein (-1: none, -2: undecided)
};

struct KfCand { // one lane's PAPA candidates of one segment: best and second best certain ones, best uncertain one
    int tag, cen;
    double best, runner, ambv;
};

__device__ __forceinline__ int ring_back(int slot, int by) { // (slot - by) mod FRING for 0 <= slot < FRING
    const int t = slot - by;
    return t < 0 ? t + FRING : t;
}

__global__ __launch_bounds__(64) void k_tracks20f(const uint8_t *__restrict__ codes, const uint4 *__restrict__ order,
                                                  uint32_t nprot, uint64_t total, const DevTables *__restrict__ T,
                                                  const KbDivTab *__restrict__ DT, plaac_row *__restrict__ rows,
                                                  const uint32_t *__restrict__ huge, uint2 *__restrict__ clist,
                                                  uint32_t *__restrict__ ccount, uint32_t *__restrict__ fblist,
                                                  uint32_t *__restrict__ fbcount) {
    constexpr int NP = KB_PROTEINS_PER_BLOCK;
    __shared__ KfShared Z;
    if (*huge) return; // a protein too long for the int32 stream axis: k_tracks20 scores the whole batch
    const int lane = threadIdx.x;
    double amax_h = 0.0, amax_p = 0.0;
    if (lane < KC_ROWS) {
        const int k = lane < NAA ? lane : (lane == KC_DUP ? 13 : 0); // 22 -> X, 23 -> P
        const bool none = lane == KC_NONE;
        const double vh = none ? 0.0 : T->hyd[k], vp = (none || lane == KC_DUP) ? 0.0 : T->lod[k];
        Z.t_hp[lane] = kb_d2{vh, vp};
        Z.t_chg[lane] = none ? 0 : T->chg[k];
        amax_h = fabs(vh);
        amax_p = fabs(vp);
    }
    for (int i = lane; i < FRING + 1; i += 64) {
        Z.p1[i] = kb_d2{0.0, 0.0};
        Z.p2[i] = kb_d2{0.0, 0.0};
        Z.pre[i] = 0;
    }
    // ---- segment table (as k_tracks20s; segments start at multiples of FB, followed by >= 20 empty positions)
    {
        uint4 it = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t idx = blockIdx.x + (uint32_t)lane * gridDim.x;
        const bool have = lane < NP && idx < nprot;
        if (have) it = order[idx];
        const int n = have ? (int)it.z : 0;
        const int span = n > 0 ? ((n + KS_GAP + FB - 1) / FB) * FB : 0;
        const int incl = wave_scan_add(span);
        if (lane < NP) {
            Z.segS[lane] = incl - span;
            Z.segN[lane] = n;
            Z.segP[lane] = it.w;
            Z.segOff[lane] = ((uint64_t)it.y << 32) | it.x;
            Z.acc_numaa[lane] = 0;
            Z.acc_maxlen[lane] = 0;
            Z.seg_amb[lane] = 0;
            Z.res[lane] = -2; // undecided until the stream has passed the protein
            if (lane == NP - 1) Z.segS[NP] = incl;
            if (have && n == 0) { // skipped record (:762): zero the fields this kernel owns
                plaac_row *row = rows + it.w;
                row->papa_combo = row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = 0.0;
                row->fi_numaa = row->fi_maxrun = row->papa_cen = 0;
            }
        }
    }
    wave_sync();
    const int stream_end = Z.segS[NP]; // wave-uniform
    if (stream_end == 0) {
        if (lane == 0) ccount[blockIdx.x] = 0u;
        return;
    }
    // positions before the stream start belong to "segment 0 at negative offsets": nothing there is live
    Z.lring[lane] = (uint8_t)0;
    Z.lring[64 + lane] = (uint8_t)0;
    const uint8_t *cend = codes + total;
    const int ww1 = T->ww1, ww2 = T->ww2;
    const bool adjust = T->adjustprolines != 0;
    const double cc0 = T->cc[0], cc1 = T->cc[1], cc2 = T->cc[2];
    const int plo = (ww2 - 1) / 2;
    const int nchunks = (stream_end - KS_GAP + FLAG2 + FC - 1) / FC;
    // ---- error bounds of this block (see the header comment)
    double E_G, E_T, E_v;
    {
        const double A_h = wave_max_f64(amax_h), A_p = wave_max_f64(amax_p);
        const double CC = fabs(cc0) * A_h + fabs(cc1) + fabs(cc2);
        const double u2 = 0x1p-52, slack = 0x1p-45; // 2u, 256u
        const double nch = (double)(nchunks + 16), L = (double)stream_end;
        const double E_Ph = nch * u2 * L * A_h, E_Pp = nch * u2 * L * A_p;
        const double E_Sh = 2.0 * E_Ph + slack * 41.0 * A_h, E_Sp = 2.0 * E_Pp + slack * 41.0 * A_p;
        E_G = fabs(cc0) * E_Sh + slack * 41.0 * CC;
        const double E_Qf = nch * u2 * L * 41.0 * CC, E_Qp = nch * u2 * L * 41.0 * A_p;
        E_T = 2.0 * E_Qf + 82.0 * E_G + slack * 1681.0 * CC;
        const double E_Tp = 2.0 * E_Qp + 41.0 * E_Sp + slack * 1681.0 * A_p;
        E_v = E_Tp * (1.0 / 41.0) + slack * A_p;
    }
    const double ninf = -INFINITY;
    wave_sync();

    // segment of stream position s (stage 0 only; the later stages take it from the lane ring)
    int c0 = 0;
    struct Fetch { // what stage 0 of one iteration needs from memory: issued one iteration ahead
        int sS, nk, i0;
        uint32_t wa, wb, wc;
    };
    auto fetch = [&](int c) {
        Fetch F;
        const int s = FC * c + FB * lane;
        while (c0 + 1 < NP && __builtin_amdgcn_readfirstlane(Z.segS[c0 + 1]) <= FC * c) ++c0; // wave-uniform
        int k = c0;
        for (int kk = c0 + 1; kk < NP; ++kk) { // wave-uniform trip count: segments that start inside the iteration
            const int sk = __builtin_amdgcn_readfirstlane(Z.segS[kk]);
            if (sk > FC * c + FC - 1) break;
            k = s >= sk ? kk : k;
        }
        const int n = Z.segN[k];
        F.sS = Z.segS[k];
        F.nk = n | (k << 16);
        F.i0 = s - F.sS;
        F.wa = F.wb = F.wc = 0u;
        if (F.i0 < n) {
            const uint8_t *x = codes + Z.segOff[k] + F.i0;
            F.wa = load4(x - 2, codes, cend); // residues i0-2 .. i0+1
            F.wb = load4(x + 2, codes, cend); //          i0+2 .. i0+5
            F.wc = load4(x + 6, codes, cend); //          i0+6
        }
        return F;
    };

    KfCand cur{-1, -1, ninf, ninf, ninf}, prv{-1, -1, ninf, ninf, ninf};
    double carry_h = 0.0, carry_p = 0.0, carry_f = 0.0, carry_q = 0.0;
    int carry_c = 0, last_zero = -1, last_flag = 0, fin = 0;
    int w0 = FB * lane; // ring slot of stream position s = FC * c + FB * lane
    Fetch nxt = fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        const int s = FC * c + FB * lane;
        const Fetch F = nxt;
        // ---- stage 0: residues -> table values and charges of the lane's FB positions; running prefix sums.
        //      hP[j] = (P_h, P_p)[s + j], cP[j] = charge prefix at s + j  (j = 0 .. FB; [0] = the lane's exclusive base)
        double hPh[FB + 1], hPp[FB + 1];
        int cP[FB + 1];
        {
            const int n = F.nk & 0xffff;
            double ah[FB], ap[FB];
            int ch[FB];
            uint32_t cb[FB + 2]; // residues i0-2 .. i0+6
            cb[0] = F.i0 >= 2 ? (F.wa & 0xffu) : 255u;
            cb[1] = F.i0 >= 1 ? ((F.wa >> 8) & 0xffu) : 255u;
            cb[2] = (F.wa >> 16) & 0xffu;
            cb[3] = F.wa >> 24;
            cb[4] = F.wb & 0xffu;
            cb[5] = (F.wb >> 8) & 0xffu;
            cb[6] = (F.wb >> 16) & 0xffu;
            cb[7] = F.wb >> 24;
            cb[8] = F.wc & 0xffu;
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const uint32_t cd = cb[2 + j] < 22u ? cb[2 + j] : 22u;
                const bool dup = adjust && cd == 13u && (cb[1 + j] == 13u || cb[j] == 13u); // (:2653-2654)
                const uint32_t kc = F.i0 + j < n ? (dup ? (uint32_t)KC_DUP : cd) : (uint32_t)KC_NONE;
                const kb_d2 hp = Z.t_hp[kc];
                ah[j] = hp.x;
                ap[j] = hp.y;
                ch[j] = Z.t_chg[kc];
            }
#pragma unroll
            for (int j = 1; j < FB; ++j) { // inclusive in-lane prefixes
                ah[j] = ah[j - 1] + ah[j];
                ap[j] = ap[j - 1] + ap[j];
                ch[j] = ch[j - 1] + ch[j];
            }
            const double sh = wave_scan_add_f64(ah[FB - 1]), sp = wave_scan_add_f64(ap[FB - 1]);
            const int sc = wave_scan_add(ch[FB - 1]);
            hPh[0] = carry_h + wave_shr1_f64(0.0, sh);
            hPp[0] = carry_p + wave_shr1_f64(0.0, sp);
            cP[0] = carry_c + wave_shr1(0, sc);
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                hPh[j + 1] = hPh[0] + ah[j];
                hPp[j + 1] = hPp[0] + ap[j];
                cP[j + 1] = cP[0] + ch[j];
                Z.p1[w0 + j] = kb_d2{hPh[j + 1], hPp[j + 1]}; // prefix at x = s + j + 1 -> slot x - 1
                Z.pre[w0 + j] = cP[j + 1];
            }
            Z.lring[(64 * c + lane) & 127] = (uint8_t)(F.nk >> 16);
            carry_h = carry_h + bcast_lane(sh, 63);
            carry_p = carry_p + bcast_lane(sp, 63);
            carry_c += bcast_lane(sc, 63);
        }
        if (c + 1 < nchunks) nxt = fetch(c + 1); // in flight during the rest of this iteration
        wave_sync();
        // ---- stage 2 operands that come from memory: the reciprocal of the second-level denominator of every position
        const int k2 = Z.lring[(64 * c + lane - 6) & 127];
        const int n2 = Z.segN[k2], i2 = s - FLAG2 - Z.segS[k2];
        int clo = imax(plo, TW), cspan = imin(n2 - plo - 1, n2 - TW - 1) - clo; // centres: in range and defined
        if (cspan < 0) {
            clo = 0x40000000;
            cspan = 0;
        }
        const bool any_cand = i2 + FB > clo && i2 <= clo + cspan;
        double rden[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j) rden[j] = 0.0;
        if (any_cand) {
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int i = i2 + j;
                const int ml = imin(imax(0, 2 * TW - i), 2 * TW), mr = imin(imax(0, 2 * TW - (n2 - 1 - i)), 2 * TW);
                rden[j] = DT->second[ml * (2 * TW + 1) + mr].y;
            }
        }
        // ---- stage 1: window sums 21 positions behind; FoldIndex sign, run statistics; second-level prefix sums
        double hQf[FB + 1], hQq[FB + 1];
        {
            const int k = Z.lring[(64 * c + lane - 3) & 127];
            const int n = Z.segN[k], sg = Z.segS[k], i0 = s - FLAG1 - sg;
            const int we = n - 1 < TW ? n - 1 : TW; // (:2588-2589)
            int halfw = (ww1 - 1) / 2;              // (:5010-5013)
            halfw = halfw > n / 2 ? n / 2 : halfw;
            const int dlo = halfw, dhi = n - halfw - 1;
            int dl = dlo, dsp = dhi - dlo; // FoldIndex scan domain as one unsigned comparison
            if (dsp < 0) {
                dl = 0x40000000;
                dsp = 0;
            }
            const int b1 = ring_back(w0, 2 * FLAG1); // slot of P[i - 20] for j = 0: position s - 42
            double gf[FB], gp[FB];
            uint32_t fm = 0u; // bit j: FoldIndex certainly negative at the lane's j-th position (inside the scan domain)
            bool amb = false;
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int i = i0 + j;
                const kb_d2 lo = Z.p1[b1 + j];
                const int csum = cP[j] - Z.pre[b1 + j];
                const double S_h = hPh[j] - lo.x, S_p = hPp[j] - lo.y;
                const int m = 1 + imin(i, we) + imin(n - i - 1, we); // residues under the window = weight (live positions)
                // G = m * FoldIndex in real arithmetic (axpbypc :2050 times the window count). Positions outside the
                // protein give bounded garbage that no used second-level window contains.
                const double G = __builtin_fma(cc0, S_h, __builtin_fma(cc1, fabs((double)csum), cc2 * (double)m));
                gf[j] = G;
                gp[j] = S_p;
                const bool dom = (unsigned)(i - dl) <= (unsigned)dsp;
                fm |= (dom && G < -E_G) ? (1u << j) : 0u;
                amb |= dom && fabs(G) <= E_G;
            }
            if (amb) Z.seg_amb[k] = 1; // a FoldIndex the bound cannot sign: the exact kernel scores this protein
#pragma unroll
            for (int j = 1; j < FB; ++j) { // second-level prefix sums
                gf[j] = gf[j - 1] + gf[j];
                gp[j] = gp[j - 1] + gp[j];
            }
            const double sf = wave_scan_add_f64(gf[FB - 1]), sq = wave_scan_add_f64(gp[FB - 1]);
            hQf[0] = carry_f + wave_shr1_f64(0.0, sf);
            hQq[0] = carry_q + wave_shr1_f64(0.0, sq);
            const int w1 = ring_back(w0, FLAG1);
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                hQf[j + 1] = hQf[0] + gf[j];
                hQq[j + 1] = hQq[0] + gp[j];
                Z.p2[w1 + j] = kb_d2{hQf[j + 1], hQq[j + 1]};
            }
            carry_f = carry_f + bcast_lane(sf, 63);
            carry_q = carry_q + bcast_lane(sq, 63);
            // FoldIndex<0 runs (:5020-5058): a run is accounted where it ENDS, i.e. at an unflagged position whose
            // predecessor is flagged; its start is one past the last unflagged position before it (in-lane from the
            // bit mask, else wave max-scan, else the carry). Only lanes that see a run end do the length arithmetic.
            {
                const int s1 = s - FLAG1;
                const uint32_t zm = ~fm & ((1u << FB) - 1u);
                const int lanemax = zm ? s1 + (31 - __builtin_clz(zm)) : INT_MIN;
                const int sc = wave_scan_max(lanemax);
                const int before0 = imax(wave_shr1(INT_MIN, sc), last_zero);
                const uint32_t prev = (uint32_t)wave_shr1(last_flag, (int)(fm >> (FB - 1)));
                uint32_t ends = zm & ((fm << 1) | prev);
                if (ends) {
                    int numaa = 0, maxlen = 0;
                    do {
                        const int j = __builtin_ctz(ends);
                        ends &= ends - 1u;
                        const uint32_t zb = j >= 1 ? (zm & ((1u << (j - 1)) - 1u)) : 0u; // unflagged ones below the run
                        const int before = zb ? s1 + (31 - __builtin_clz(zb)) : before0;
                        int rs = before + 1 - sg, re = i0 + j - 1; // the run that ends here, in protein coordinates
                        rs = rs == dlo ? 0 : rs;
                        re = re == dhi ? n - 1 : re;
                        const int len = re - rs + 1;
                        const int cnt = len >= 5 ? len : 0;
                        numaa += cnt;
                        maxlen = imax(maxlen, cnt);
                    } while (ends);
                    if (numaa) {
                        atomicAdd(&Z.acc_numaa[k], numaa);
                        atomicMax(&Z.acc_maxlen[k], maxlen);
                    }
                }
                last_zero = imax(last_zero, bcast_lane(sc, 63));
                last_flag = bcast_lane((int)(fm >> (FB - 1)), 63);
            }
        }
        wave_sync();
        // ---- stage 2: second smoothing another 21 positions behind; PAPA candidates with their certainty
        {
            if (k2 != cur.tag) {
                prv = cur;
                cur = KfCand{k2, -1, ninf, ninf, ninf};
            }
            if (any_cand) { // some position of this lane can be a centre
                const int b2 = ring_back(w0, 3 * FLAG1); // slot of Q[k - 20] for j = 0: position s - 63
#pragma unroll
                for (int j = 0; j < FB; ++j) {
                    const int i = i2 + j;
                    const bool cand = (unsigned)(i - clo) <= (unsigned)cspan;
                    const kb_d2 lo = Z.p2[b2 + j];
                    const double Tf = hQf[j] - lo.x, Tp = hQq[j] - lo.y;
                    const double v = Tp * rden[j];
                    const bool def = cand && Tf < -E_T;       // fix2 certainly negative
                    const bool unc = cand && fabs(Tf) <= E_T; // fix2 within the bound of zero
                    const double vv = def ? v : ninf;
                    cur.runner = __builtin_fmax(cur.runner, __builtin_fmin(cur.best, vv)); // second best (ties included)
                    cur.cen = vv > cur.best ? i : cur.cen;                               // strict >: first maximum
                    cur.best = __builtin_fmax(cur.best, vv);
                    cur.ambv = __builtin_fmax(cur.ambv, unc ? v : ninf);
                }
            }
        }
        // ---- proteins whose last residue stage 2 has passed in this iteration: which centre, or "undecided"
        const int passed = FC * c + FC - 1 - FLAG2;
        while (fin < NP) { // wave-uniform
            const int n = __builtin_amdgcn_readfirstlane(Z.segN[fin]);
            const int sg = __builtin_amdgcn_readfirstlane(Z.segS[fin]);
            if (n > 0 && sg + n - 1 > passed) break;
            if (n > 0) {
                const bool mc = cur.tag == fin, mp = prv.tag == fin;
                const int mine = mc ? cur.cen : (mp ? prv.cen : -1);
                const double mbest = mine >= 0 ? (mc ? cur.best : prv.best) : ninf;
                const double mrun = mc ? cur.runner : (mp ? prv.runner : ninf);
                const double mamb = mc ? cur.ambv : (mp ? prv.ambv : ninf);
                // The lane that holds the largest certain candidate, found on single-precision keys (rounding to float
                // is monotone: a lane whose key is below the unique largest key holds a smaller double); the exact
                // closeness test below is on the doubles. Two lanes with the same key: left to the exact kernel.
                const float key = (float)mbest;
                const float kmax = wave_max_f32(key);
                const unsigned long long top = __ballot(mine >= 0 && key == kmax);
                const int wl = top ? (int)__builtin_ctzll(top) : 0;
                const double g1 = top ? bcast_lane_dyn(mbest, wl) : ninf;
                const double mark = g1 - 2.0 * E_v; // -inf when there is no certain candidate
                const unsigned long long close = __ballot(mine >= 0 && mbest >= mark);
                const bool others = __ballot((mrun > ninf && mrun >= mark) || (mamb > ninf && mamb >= mark)) != 0ull;
                const bool undecided = __popcll(top) > 1 || __popcll(close) > 1 || others;
                const int pcen = top ? __builtin_amdgcn_readlane(mine, __builtin_amdgcn_readfirstlane(wl)) : -1;
                if (lane == 0) Z.res[fin] = undecided ? -2 : pcen;
            }
            ++fin;
        }
        w0 += FC - FRING; // FC mod FRING steps forward = 63 slots back
        w0 = w0 < 0 ? w0 + FRING : w0;
    }
    wave_sync();
    // ---- per-protein results, one lane per protein of the block: rows of the decided ones, the refine list (proteins
    //      with a centre), the fallback list (a FoldIndex or PAPA decision the bounds could not make)
    {
        const bool have = lane < NP && Z.segN[lane < NP ? lane : 0] > 0;
        const int res = have ? Z.res[lane] : -1;
        const bool fb = have && (res == -2 || Z.seg_amb[lane] != 0);
        const bool cen = have && !fb && res >= 0;
        const uint32_t pidx = blockIdx.x + (uint32_t)lane * gridDim.x; // plan index of this lane's protein
        const unsigned long long mfb = __ballot(fb), mcen = __ballot(cen);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (mfb) { // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(fbcount, (uint32_t)__popcll(mfb));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (fb) fblist[base + (uint32_t)__popcll(mfb & below)] = pidx;
        }
        if (cen) clist[(size_t)blockIdx.x * NP + (uint32_t)__popcll(mcen & below)] = make_uint2(pidx, (uint32_t)res);
        if (lane == 0) ccount[blockIdx.x] = (uint32_t)__popcll(mcen);
        if (have && !fb) {
            plaac_row *row = rows + Z.segP[lane];
            row->fi_numaa = Z.acc_numaa[lane];
            row->fi_maxrun = Z.acc_maxlen[lane];
            row->papa_cen = res;
            if (res < 0) {
                row->papa_combo = -INFINITY;
                row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = __builtin_nan("");
            }
        }
    }
}

// The four values at the PAPA centre in the reference's own operation order (disorderreport :4877-4905 at one
// position): first-level windows of hydropathy / charge / llr / PAPA log-odds at the 41 positions c-20 .. c+20
// (one thread each, 41 fixed-order taps; taps outside the protein add +0.0), FoldIndex, the weights, then the
// three weighted second-level sums over those 41 values. A block serves the centres of one k_tracks20f block,
// six proteins at a time (6 x 41 = 246 of 256 threads).
constexpr int RF_SLOTS = 6, RF_SPAN = 4 * TW + 1; // 81 residues under the two window levels
__global__ __launch_bounds__(256) void k_refine_centres(const uint8_t *__restrict__ codes, const uint4 *__restrict__ order,
                                                        uint64_t total, const DevTables *__restrict__ T,
                                                        plaac_row *__restrict__ rows, const uint32_t *__restrict__ huge,
                                                        const uint2 *__restrict__ clist,
                                                        const uint32_t *__restrict__ ccount) {
    __shared__ alignas(16) kb_d2 t_hl[KC_ROWS]; // (hydropathy, llr) by window code
    __shared__ alignas(16) kb_d2 t_pc[KC_ROWS]; // (PAPA log-odds, charge as a double: sums of -1/0/1 are exact)
    // per-position values of the 81 residues around the centre: a thread's 41 taps are consecutive entries, so the
    // reads of neighbouring threads are conflict-free 16-byte reads with immediate offsets (no code -> table hop per tap)
    __shared__ alignas(16) kb_d2 s_hl[RF_SLOTS][RF_SPAN + 1], s_pc[RF_SLOTS][RF_SPAN + 1];
    __shared__ double s_w[RF_SLOTS][3][2 * TW + 2];
    __shared__ double s_llr1[RF_SLOTS];
    __shared__ int s_n[RF_SLOTS], s_cen[RF_SLOTS];
    __shared__ uint32_t s_p[RF_SLOTS];
    if (*huge) return;
    const uint32_t cnt = ccount[blockIdx.x];
    if (cnt == 0u) return;
    const int tid = threadIdx.x;
    if (tid < KC_ROWS) {
        const int k = tid < NAA ? tid : (tid == KC_DUP ? 13 : 0);
        const bool none = tid == KC_NONE;
        t_hl[tid] = kb_d2{none ? 0.0 : T->hyd[k], none ? 0.0 : T->llr[k]};
        t_pc[tid] = kb_d2{(none || tid == KC_DUP) ? 0.0 : T->lod[k], none ? 0.0 : (double)T->chg[k]};
    }
    const bool adjust = T->adjustprolines != 0;
    const double cc0 = T->cc[0], cc1 = T->cc[1], cc2 = T->cc[2];
    const int slot = tid / (2 * TW + 1), l = tid - slot * (2 * TW + 1);
    const uint8_t *cend = codes + total;
    for (uint32_t r0 = 0; r0 < cnt; r0 += RF_SLOTS) {
        const bool act = slot < RF_SLOTS && r0 + (uint32_t)slot < cnt;
        int n = 0, cen = 0;
        uint32_t pidx = 0;
        const uint8_t *x = codes;
        if (act) {
            const uint2 e = clist[(size_t)blockIdx.x * KB_PROTEINS_PER_BLOCK + r0 + slot];
            const uint4 it = order[e.x];
            n = (int)it.z;
            cen = (int)e.y;
            x = codes + (((uint64_t)it.y << 32) | it.x);
            pidx = it.w;
        }
        __syncthreads(); // tables ready / previous round done with the staging arrays and the slot records
        if (act && l == 0) {
            s_n[slot] = n;
            s_cen[slot] = cen;
            s_p[slot] = pidx;
        }
        if (act) { // values of positions cen-40 .. cen+40 (two per thread); outside the protein: +0.0 everywhere
            for (int t = l; t < RF_SPAN; t += 2 * TW + 1) {
                const int q = cen - 2 * TW + t;
                uint32_t kc = (uint32_t)KC_NONE;
                if (q >= 0 && q < n) {
                    const uint8_t *xq = x + q;
                    const uint32_t c0 = (xq >= codes && xq < cend) ? *xq : 0u;
                    const uint32_t cd = c0 < 22u ? c0 : 22u;
                    const bool dup = adjust && cd == 13u && ((q >= 1 && xq[-1] == 13) || (q >= 2 && xq[-2] == 13));
                    kc = dup ? (uint32_t)KC_DUP : cd;
                }
                s_hl[slot][t] = t_hl[kc];
                s_pc[slot][t] = t_pc[kc];
            }
        }
        __syncthreads();
        if (act) { // first level at i = cen - 20 + l: fixed-order 41-term sums
            double sh = 0.0, sl = 0.0, sp = 0.0, sc = 0.0;
            const kb_d2 *__restrict__ vh = &s_hl[slot][l], *__restrict__ vp = &s_pc[slot][l];
#pragma unroll
            for (int t = 0; t <= 2 * TW; ++t) { // increasing position
                const kb_d2 a = vh[t], b = vp[t];
                sh = sh + a.x;
                sl = sl + a.y;
                sp = sp + b.x;
                sc = sc + b.y;
            }
            // quotients, FoldIndex, weights
            const int i = cen - TW + l;
            const int lo = imax(i - TW, 0), hi = imin(i + TW, n - 1);
            const SharedDiv div((double)(hi - lo + 1));
            const double hydro = div(sh), charge = div(sc);
            const double fi = (cc0 * hydro + cc1 * fabs(charge)) + cc2; // axpbypc (:2050)
            const double llr1 = div(sl), papa = div(sp);
            const double wt = (double)(1 + imin(i, TW) + imin(n - i - 1, TW));
            s_w[slot][0][l] = wt * fi;
            s_w[slot][1][l] = wt * llr1;
            s_w[slot][2][l] = wt * papa;
            if (l == TW) s_llr1[slot] = llr1;
        }
        __syncthreads();
        // second level at the centres: the 18 fixed-order sums (6 proteins x 3 tracks) on 18 lanes of ONE wave
        if (tid < 3 * RF_SLOTS) {
            const int sl2 = tid / 3, trk = tid - 3 * sl2;
            if (r0 + (uint32_t)sl2 < cnt) {
                double s2 = 0.0;
#pragma unroll 4
                for (int t = 0; t <= 2 * TW; ++t) s2 = s2 + s_w[sl2][trk][t];
                const int n2 = s_n[sl2], c2 = s_cen[sl2];
                const int den = (2 * TW + 1) + window_weight_side(c2, TW) + window_weight_side(n2 - 1 - c2, TW);
                const double q = s2 / (double)den;
                plaac_row *row = rows + s_p[sl2];
                if (trk == 0) {
                    row->papa_fi = q;
                } else if (trk == 1) {
                    row->papa_llr2 = q;
                    row->papa_llr = s_llr1[sl2];
                } else {
                    row->papa_combo = q;
                    row->papa_prop = q;
                }
            }
        }
    }
}

// positions-per-lane variant that wastes the fewest slots for a protein of n residues: iterations x
// (per-iteration cost ~ B + fixed part). A protein shorter than one iteration needs no pipeline lag.
__device__ __forceinline__ int kb_choose_b(int n) {
    auto cost = [n](int b, int lag2) {
        return n < 64 * b ? 10 * b + 7 : ((n + lag2 + 64 * b - 1) / (64 * b)) * (10 * b + 7);
    };
    const int c2 = cost(2, KbGeom<2>::LAG2), c3 = cost(3, KbGeom<3>::LAG2), c4 = cost(4, KbGeom<4>::LAG2);
    return (c4 <= c3 && c4 <= c2) ? 4 : (c3 <= c2 ? 3 : 2);
}


// ------------------------------------------------------------------------------------------------
// Sweeps: the window tracks depend on the background mix alpha only through the llr table, and the summary row
// keeps just two llr-derived values, both AT the PAPA centre (which itself does not depend on alpha). For every
// further alpha of a sweep K-B therefore shrinks to this kernel: one wave per protein takes the centre from the
// row the full kernel wrote for the first alpha, recomputes PAPAllr (first-level window, :4891-4897) and PAPAllr2
// (weighted second smoothing of the 2w+1 first-level values around it, :4903-4905) with this group's llr table -
// the same fixed-order sums, ~1/10 of the work - and copies the alpha-independent fields.
// ------------------------------------------------------------------------------------------------
constexpr int LLRAT_MAXW = 31; // 2w+1 first-level windows, one per lane
__global__ __launch_bounds__(256) void k_llr_at_centre(const uint8_t *__restrict__ codes,
                                                        const uint64_t *__restrict__ offsets,
                                                        const uint32_t *__restrict__ neff, uint32_t nprot,
                                                        const DevTables *__restrict__ T,
                                                        const plaac_row *__restrict__ src, plaac_row *__restrict__ dst) {
    __shared__ double s_llr[NAA];
    __shared__ uint8_t s_code[4][4 * LLRAT_MAXW + 4];
    __shared__ double s_val[4][64];
    if (threadIdx.x < NAA) s_llr[threadIdx.x] = T->llr[threadIdx.x];
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t p = blockIdx.x * 4u + (uint32_t)wv;
    if (p >= nprot) return;
    const int n = (int)neff[p];
    const plaac_row *r = src + p;
    plaac_row *o = dst + p;
    const int c = r->papa_cen;
    double llr1c = __builtin_nan(""), llx2 = __builtin_nan("");
    if (n > 0 && c >= 0) { // wave-uniform
        const int w = T->ww3 / 2;
        const int we = n - 1 < w ? n - 1 : w; // (:2588-2589)
        const uint8_t *x = codes + offsets[p];
        // residues c-2we .. c+2we (positions outside the protein: X, never used because of the range checks)
        for (int k = lane; k <= 4 * we; k += 64) {
            const int q = c - 2 * we + k;
            s_code[wv][k] = (q >= 0 && q < n) ? (uint8_t)ld_code(x, (uint32_t)q) : (uint8_t)0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double v = 0.0, l1 = 0.0;
        if (lane <= 2 * we) { // first-level window of position i = c - we + lane (inside the protein: c is a valid centre)
            const int i = c - we + lane;
            double sum = 0.0;
            for (int t = 0; t <= 2 * we; ++t) { // increasing position; taps outside the protein are skipped
                const int q = i - we + t;
                if (q >= 0 && q < n) sum = sum + s_llr[s_code[wv][lane + t]];
            }
            const int lo = imax(i - we, 0), hi = imin(i + we, n - 1);
            l1 = sum / (double)(hi - lo + 1);
            v = (double)(1 + imin(i, we) + imin(n - i - 1, we)) * l1;
        }
        s_val[wv][lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double s2 = 0.0;
        for (int t = 0; t <= 2 * we; ++t) s2 = s2 + s_val[wv][t];
        const int den = (2 * we + 1) + window_weight_side(c, we) + window_weight_side(n - 1 - c, we);
        // the centre comes from the PAPA window (ww2); for the llr window (ww3) it may lie in the NaN margin (:2597-2600)
        llx2 = (c >= we && c <= n - we - 1) ? s2 / (double)den : __builtin_nan("");
        llr1c = bcast_lane_dyn(l1, we);
    }
    if (lane == 0) {
        o->papa_combo = r->papa_combo;
        o->papa_prop = r->papa_prop;
        o->papa_fi = r->papa_fi;
        o->fi_numaa = r->fi_numaa;
        o->fi_maxrun = r->fi_maxrun;
        o->papa_cen = c;
        if (n > 0 && c >= 0) {
            o->papa_llr = llr1c;
            o->papa_llr2 = llx2;
        } else { // no centre (NaN) or skipped record (0.0): the same values for every alpha
            o->papa_llr = r->papa_llr;
            o->papa_llr2 = r->papa_llr2;
        }
    }
}

template <bool TRACKS>
__global__ __launch_bounds__(64) void k_tracks20(const uint8_t *__restrict__ codes,
                                                 const uint64_t *__restrict__ offsets,
                                                 const uint32_t *__restrict__ neff,
                                                 const uint4 *__restrict__ order, uint32_t nprot, uint64_t total,
                                                 const DevTables *__restrict__ T, plaac_row *__restrict__ rows,
                                                 TrackPtrs tr, const uint32_t *__restrict__ huge, uint32_t only_if_huge,
                                                 const uint32_t *__restrict__ list,
                                                 const uint32_t *__restrict__ list_count) {
    __shared__ KbShared S;
    // list == null: the whole plan (when the stream form cannot take the batch, or always with PLAAC_KB_PER_PROTEIN=1);
    // else: the plan items the filter kernel could not decide (k_tracks20f's fallback list)
    if (!list && only_if_huge && *huge == 0u) return;
    if (list && *huge != 0u) return;
    const uint32_t nitems = list ? *list_count : nprot;
    const int lane = threadIdx.x;
    if (lane < KC_ROWS) {
        const int k = lane < NAA ? lane : (lane == KC_DUP ? 13 : 0); // 22 -> X, 23 -> P
        const bool none = lane == KC_NONE;
        S.t_hl[lane].x = none ? 0.0 : T->hyd[k];
        S.t_hl[lane].y = none ? 0.0 : T->llr[k];
        S.t_lod[lane].x = (none || lane == KC_DUP) ? 0.0 : T->lod[k];
        S.t_lod[lane].y = 0.0;
        S.t_chg[lane] = none ? 0 : T->chg[k];
    }
    KbConst K;
    K.codes = codes;
    K.cend = codes + total;
    K.ww1 = T->ww1;
    K.ww2 = T->ww2;
    K.adjust = T->adjustprolines != 0;
    K.cc0 = T->cc[0];
    K.cc1 = T->cc[1];
    K.cc2 = T->cc[2];

    // metadata of the first protein of this block; the next one is prefetched while the current is scored
    uint32_t b = blockIdx.x;
    uint4 it_nxt = make_uint4(0u, 0u, 0u, 0u);
    auto item = [&](uint32_t k) { return order[list ? list[k] : k]; };
    if (b < nitems) it_nxt = item(b);

    for (; b < nitems; b += gridDim.x) {
        const uint32_t p = it_nxt.w;
        const int n = (int)it_nxt.z;
        const uint64_t off = ((uint64_t)it_nxt.y << 32) | it_nxt.x;
        if (b + gridDim.x < nitems) it_nxt = item(b + gridDim.x); // prefetch the next protein's plan item
        plaac_row *row = rows + p;
        if (n == 0) {
            if (lane == 0) {
                row->papa_combo = row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = 0.0;
                row->fi_numaa = row->fi_maxrun = row->papa_cen = 0;
            }
            continue;
        }
        wave_sync(); // the previous protein is done with the rings
        const uint8_t *__restrict__ x = codes + off;
        const int bsel = kb_choose_b(n); // wave-uniform
        const bool nolag = n < 64 * bsel;
        switch (bsel) {
        case 2: tracks20_protein<2, TRACKS>(S, K, x, off, n, nolag, row, tr); break;
        case 3: tracks20_protein<3, TRACKS>(S, K, x, off, n, nolag, row, tr); break;
        default: tracks20_protein<4, TRACKS>(S, K, x, off, n, nolag, row, tr); break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// histogram over valid records (:1698-1706, :1732-1739); one wave per record, grid-stride
// ------------------------------------------------------------------------------------------------
// residue codes of a host upload must be 0..21 (they index 22-row tables): one flag for the whole buffer
__global__ __launch_bounds__(256) void k_validate(const uint8_t *__restrict__ codes, uint64_t total,
                                                   uint32_t *__restrict__ flag) {
    const uint64_t nvec = total >> 4;
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nvec; i += (uint64_t)gridDim.x * 256u) {
        const uint4 v = reinterpret_cast<const uint4 *>(codes)[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) // any byte > 21 <=> (byte + 106) has bit 7 set, or the byte itself has
            bad |= (((w[k] & 0x7f7f7f7fu) + 0x6a6a6a6au) | w[k]) & 0x80808080u;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 15u)) bad |= codes[(nvec << 4) + threadIdx.x] > 21u;
    if (bad) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(256) void k_hist(const uint8_t *__restrict__ codes, const uint64_t *__restrict__ offsets,
                                              uint32_t nprot, unsigned long long *__restrict__ counts) {
    __shared__ unsigned int s_cnt[4][NAA + 2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < NAA + 2; i += 64) s_cnt[wave][i] = 0;
    unsigned long long acc = 0; // lane k (< 22) accumulates bin k for this wave
    const uint32_t nw = gridDim.x * 4u;
    for (uint32_t p = blockIdx.x * 4u + wave; p < nprot; p += nw) {
        const uint64_t b = offsets[p], e = offsets[p + 1];
        if (e <= b) continue;
        const uint64_t m = e - b;
        const uint8_t *__restrict__ x = codes + b;
        // validity: no X/* strictly inside, last residue not X (position 0 is not checked)
        bool bad = false;
        for (uint64_t i = lane; i < m; i += 64) {
            const uint32_t cq = ld_code(x, (uint32_t)i);
            if (i >= 1 && i + 1 < m && (cq == 0u || cq == 21u)) bad = true;
            if (i + 1 == m && cq == 0u) bad = true;
        }
        if (__ballot(bad) != 0ull) continue;
        for (uint64_t i = lane; i < m; i += 64) atomicAdd(&s_cnt[wave][ld_code(x, (uint32_t)i)], 1u);
        // flush per record so the 32-bit LDS bins cannot overflow
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < NAA) {
            acc += s_cnt[wave][lane];
            s_cnt[wave][lane] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane < NAA && acc) atomicAdd(&counts[lane], acc);
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C ABI — device half
// ------------------------------------------------------------------------------------------------
struct plaac_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    DevTables *d_tab = nullptr;   // tables of ctx->params
    DevTables *d_tabs = nullptr;  // tables of the groups of a sweep
    size_t cap_tabs = 0;
    std::vector<hipEvent_t> gev;  // per-group "forward pass done" events of a sweep
    std::vector<hipStream_t> gstreams; // side streams of the 2nd, 3rd ... group of a sweep (three each)
    std::vector<hipEvent_t> gjev;      // their join events
    hipEvent_t jev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // join events of the side streams
    plaac_params params;
    // plan / scratch buffers (grown on demand)
    uint32_t *d_neff = nullptr, *d_hist = nullptr, *d_bits = nullptr, *d_grow = nullptr;
    uint4 *d_order = nullptr; // the sorted plan: {offset lo, offset hi, effective length, protein index}
    uint4 *d_packed = nullptr;
    double2 *d_fwd = nullptr, *d_bwd = nullptr; // track mode: forward / backward pairs, group-interleaved
    uint32_t *h_pin = nullptr; // pinned words: [0] total packed rows of a call, [1] upload validation flag
    // pinned staging for the host-buffer entry points (pageable memcpy runs at a tenth of the link rate)
    static constexpr size_t STAGE_BYTES = 16u << 20;
    uint8_t *h_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    uint32_t *d_flag = nullptr;
    KbDivTab *d_divtab = nullptr; // reciprocal tables of the window kernel
    // summary-mode window kernel in filter form: centres per filter block, fallback list, [0] = its length
    uint2 *d_clist = nullptr;
    uint32_t *d_ccount = nullptr, *d_fblist = nullptr, *d_fbcount = nullptr;
    size_t cap_clist = 0, cap_ccount = 0, cap_fblist = 0;
    double *d_lat = nullptr; // latency forms: [0, nprot) lmarginalprob of hmm1, [nprot, 2 nprot) total of hmm0
    size_t cap_lat = 0;
    double *d_corep = nullptr; // latency forms: masked prefix sums of the long wave-groups, packed row numbering
    void *d_corepart = nullptr; // their per-row best windows
    size_t cap_corep = 0, cap_corepart = 0;
    bool kb_filter = true; // PLAAC_KB_FILTER=0: exact stream kernel (k_tracks20s) in summary mode too
    size_t cap_prot = 0, cap_order = 0, cap_bits = 0, cap_fwd = 0, cap_bwd = 0, cap_grow = 0, cap_packed = 0;
    // staging for the host-buffer entry points
    uint8_t *d_codes = nullptr;
    uint64_t *d_offsets = nullptr;
    plaac_row *d_rows = nullptr;
    uint8_t *d_trk8 = nullptr;
    double *d_trk64 = nullptr;
    unsigned long long *d_counts = nullptr;
    size_t cap_codes = 0, cap_offs = 0, cap_rows = 0, cap_trk = 0;
    static constexpr int EV_SETS = 32; // timings of the last 32 scored batches
    static constexpr int EV_PER = 15;  // start, planned, {begin,end} x {vit,fwd,win,tracks}, joined, {begin,end} x {pack,bwd}
    hipEvent_t ev[EV_SETS][EV_PER] = {};
    uint64_t ncalls = 0;
    hipStream_t aux[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // high-priority side streams of the K-A roles
    uint32_t vit_stop = 0; // DIAGNOSTIC, PLAAC_VIT_STOP=1|2: k_vit stops after that sweep (timing the sweeps; results are wrong)
    int latency_mode = -1; // PLAAC_LATENCY_MODE=0/1 forces the throughput / latency forms of the K-A kernels (-1: per batch)
    bool serial = false;                              // PLAAC_SERIAL_STREAMS=1: everything on one stream
    bool generic_tracks = false;                      // PLAAC_GENERIC_TRACKS=1: never use the ww=41 fast path
    bool per_protein_tracks = false;                  // PLAAC_KB_PER_PROTEIN=1: ww=41 fast path, one protein at a time
    std::string err;
};

namespace {

thread_local std::string g_create_err;

#define PL_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return e_ == hipErrorOutOfMemory ? PLAAC_ERR_NOMEM : PLAAC_ERR_DEVICE;                     \
        }                                                                                              \
    } while (0)

plaac_status fail(plaac_ctx *ctx, plaac_status st, const char *msg) {
    if (ctx) ctx->err = msg;
    return st;
}

void fill_tables(const plaac_params &P, DevTables &D) {
    std::memset(&D, 0, sizeof D);
    for (int k = 0; k < NAA; ++k) {
        D.row[k][R_LE0] = P.hmm1.le[0][k];
        D.row[k][R_LE1] = P.hmm1.le[1][k];
        D.row[k][R_LLR] = P.llr[k];
        D.row[k][R_HYD] = P.hydro2[k];
        D.row[k][R_LE0H] = P.hmm0.le[0][k];
        D.lod[k] = P.lodpapa[k];
        D.hyd[k] = P.hydro2[k];
        D.llr[k] = P.llr[k];
        D.chg[k] = (int32_t)P.charge[k];
    }
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 2; ++j) D.lt[i][j] = P.hmm1.lt[i][j];
        D.li[i] = P.hmm1.li[i];
        D.lf[i] = P.hmm1.lf[i];
    }
    D.h0_lt00 = P.hmm0.lt[0][0];
    D.h0_li0 = P.hmm0.li[0];
    D.h0_lf0 = P.hmm0.lf[0];
    for (int i = 0; i < 3; ++i) D.cc[i] = P.cc[i];
    D.big_neg = P.big_neg;
    D.corelength = P.corelength;
    D.ww1 = P.ww1;
    D.ww2 = P.ww2;
    D.ww3 = P.ww3;
    D.adjustprolines = P.adjustprolines;
    std::memcpy(D.loglut, P.loglut, sizeof P.loglut); // D.loglut[LUTLEN] stays 0.0
}

// The kernels exploit the structure of the reference's two models; refuse anything else loudly.
const char *check_params(const plaac_params &P) {
    if (P.corelength < 1) return "corelength must be >= 1";
    if (P.ww1 < 1 || P.ww2 < 1 || P.ww3 < 1) return "window sizes must be >= 1";
    if (P.ww1 / 2 > 256 || P.ww2 / 2 > 256 || P.ww3 / 2 > 256) return "window sizes above 513 are not supported";
    for (int k = 0; k < NAA; ++k) {
        if (!(std::isfinite(P.hmm1.le[0][k]) && std::isfinite(P.hmm1.le[1][k]) && std::isfinite(P.hmm0.le[0][k])))
            return "hmm emission log-probabilities must be finite";
        if (P.charge[k] != -1.0 && P.charge[k] != 0.0 && P.charge[k] != 1.0) return "charge table must be -1/0/1";
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            if (!std::isfinite(P.hmm1.lt[i][j])) return "hmm1 transitions must be finite in log space";
    // hmm0 must be the identity-transition null model started in state 0 (prionhmm0, :988-1001)
    if (!(P.hmm0.lt[0][1] == -INFINITY && P.hmm0.lt[1][0] == -INFINITY && P.hmm0.li[1] == -INFINITY &&
          std::isfinite(P.hmm0.lt[0][0]) && std::isfinite(P.hmm0.li[0]) && std::isfinite(P.hmm0.lf[0])))
        return "hmm0 must be the single-state null model";
    return nullptr;
}

template <class Tp>
plaac_status grow(plaac_ctx *ctx, Tp *&ptr, size_t &cap, size_t need) {
    if (need <= cap && ptr) return PLAAC_OK;
    if (ptr) PL_HIP(ctx, hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    size_t want = need + need / 8 + 64;
    PL_HIP(ctx, hipMalloc((void **)&ptr, want * sizeof(Tp)));
    cap = want;
    return PLAAC_OK;
}

} // namespace

extern "C" {

const char *plaac_last_error(const plaac_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int plaac_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

plaac_status plaac_ctx_create(const plaac_params *params, int device_id, plaac_ctx **out) {
    if (!params || !out) {
        g_create_err = "plaac_ctx_create: null argument";
        return PLAAC_ERR_ARG;
    }
    *out = nullptr;
    if (const char *why = check_params(*params)) {
        g_create_err = why;
        return PLAAC_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count=0");
        return PLAAC_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_err = "device_id out of range";
        return PLAAC_ERR_ARG;
    }
    plaac_ctx *ctx = new (std::nothrow) plaac_ctx();
    if (!ctx) {
        g_create_err = "out of host memory";
        return PLAAC_ERR_NOMEM;
    }
    ctx->device = device_id;
    ctx->params = *params;
    auto bail = [&](const char *what, hipError_t err) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(err);
        plaac_ctx_destroy(ctx);
        return PLAAC_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        plaac_ctx_destroy(ctx);
        return PLAAC_ERR_DEVICE;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        for (auto &a : ctx->aux)
            if ((e = hipStreamCreateWithPriority(&a, hipStreamNonBlocking, greatest)) != hipSuccess)
                return bail("hipStreamCreateWithPriority", e);
        for (auto &je : ctx->jev)
            if ((e = hipEventCreateWithFlags(&je, hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e);
        const char *ser = std::getenv("PLAAC_SERIAL_STREAMS");
        ctx->serial = ser && ser[0] == '1';
        const char *gen = std::getenv("PLAAC_GENERIC_TRACKS");
        ctx->generic_tracks = gen && gen[0] == '1';
        const char *ppt = std::getenv("PLAAC_KB_PER_PROTEIN");
        ctx->per_protein_tracks = ppt && ppt[0] == '1';
        const char *kbf = std::getenv("PLAAC_KB_FILTER");
        ctx->kb_filter = !(kbf && kbf[0] == '0');
        if (const char *vs = std::getenv("PLAAC_VIT_STOP")) ctx->vit_stop = (uint32_t)std::atoi(vs);
        const char *lat = std::getenv("PLAAC_LATENCY_MODE");
        if (lat && (lat[0] == '0' || lat[0] == '1')) ctx->latency_mode = lat[0] - '0';
    }
    for (auto &set : ctx->ev)
        for (auto &ev : set)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc((void **)&ctx->d_tab, sizeof(DevTables))) != hipSuccess) return bail("hipMalloc(tables)", e);
    if ((e = hipMalloc((void **)&ctx->d_hist, sizeof(uint32_t) * (LEN_BINS + 1))) != hipSuccess)
        return bail("hipMalloc(hist)", e);
    if ((e = hipMalloc((void **)&ctx->d_counts, sizeof(unsigned long long) * NAA)) != hipSuccess)
        return bail("hipMalloc(counts)", e);
    if ((e = hipHostMalloc((void **)&ctx->h_pin, 64, hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_divtab, sizeof(KbDivTab))) != hipSuccess) return bail("hipMalloc(divtab)", e);
    if ((e = hipMalloc((void **)&ctx->d_fbcount, sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(fbcount)", e);
    if ((e = hipMemset(ctx->d_fbcount, 0, sizeof(uint32_t))) != hipSuccess) return bail("hipMemset(fbcount)", e);
    hipLaunchKernelGGL(k_build_divtab, dim3(((2 * TW + 1) * (2 * TW + 1) + 256) / 256), dim3(256), 0, ctx->stream,
                       ctx->d_divtab);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail("k_build_divtab", e);
    plaac_status st = plaac_ctx_set_params(ctx, params);
    if (st != PLAAC_OK) {
        g_create_err = ctx->err;
        plaac_ctx_destroy(ctx);
        return st;
    }
    *out = ctx;
    return PLAAC_OK;
}

plaac_status plaac_ctx_set_params(plaac_ctx *ctx, const plaac_params *params) {
    if (!ctx || !params) return fail(ctx, PLAAC_ERR_ARG, "plaac_ctx_set_params: null argument");
    if (const char *why = check_params(*params)) return fail(ctx, PLAAC_ERR_ARG, why);
    PL_HIP(ctx, hipSetDevice(ctx->device));
    DevTables *h = new (std::nothrow) DevTables();
    if (!h) return fail(ctx, PLAAC_ERR_NOMEM, "out of host memory");
    fill_tables(*params, *h);
    // The last scored batch may still be reading the old tables: its kernels run on the caller's stream and on the
    // non-blocking side streams, none of which a null-stream copy waits for. Its join event (recorded on the caller's
    // stream after every side stream has been joined) covers all of them.
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && ctx->ncalls > 0)
        e = hipEventSynchronize(ctx->ev[(ctx->ncalls - 1) % plaac_ctx::EV_SETS][10 /* E_JOIN */]);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tab, h, sizeof(DevTables), hipMemcpyHostToDevice);
    delete h;
    if (e != hipSuccess) {
        ctx->err = std::string("table upload: ") + hipGetErrorString(e);
        return PLAAC_ERR_DEVICE;
    }
    ctx->params = *params;
    return PLAAC_OK;
}

void plaac_ctx_destroy(plaac_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_bwd) (void)hipFree(ctx->d_bwd);
    void *bufs[] = {ctx->d_tab,  ctx->d_neff,    ctx->d_order, ctx->d_hist, ctx->d_bits,  ctx->d_fwd,   ctx->d_codes,
                    ctx->d_offsets, ctx->d_rows, ctx->d_trk8,  ctx->d_trk64, ctx->d_counts, ctx->d_grow, ctx->d_packed};
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (int i = 0; i < 2; ++i) {
        if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    }
    if (ctx->d_flag) (void)hipFree(ctx->d_flag);
    if (ctx->d_divtab) (void)hipFree(ctx->d_divtab);
    for (void *b : {(void *)ctx->d_clist, (void *)ctx->d_ccount, (void *)ctx->d_fblist, (void *)ctx->d_fbcount,
                    (void *)ctx->d_lat, (void *)ctx->d_corep, ctx->d_corepart})
        if (b) (void)hipFree(b);
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (auto &set : ctx->ev)
        for (auto &ev : set)
            if (ev) (void)hipEventDestroy(ev);
    for (auto &a : ctx->aux)
        if (a) {
            (void)hipStreamSynchronize(a);
            (void)hipStreamDestroy(a);
        }
    for (hipEvent_t e : ctx->gev)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t a : ctx->gstreams)
        if (a) {
            (void)hipStreamSynchronize(a);
            (void)hipStreamDestroy(a);
        }
    for (hipEvent_t e : ctx->gjev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->jev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_tabs) (void)hipFree(ctx->d_tabs);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

plaac_status plaac_ctx_sync(plaac_ctx *ctx) {
    if (!ctx) return PLAAC_ERR_ARG;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PLAAC_OK;
}

// One planned pass over a resident batch for `npoints` parameter sets (npoints == 1: the ordinary call).
// Points whose tables differ only in the core length form a group: the plan, the packed copy, the forward pass,
// the window tracks and Viterbi + traceback run once per group; only the two prefix-sum window searches run per
// core length (inside the same kernels, up to MAXC at a time).
static plaac_status score_points(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets, uint32_t nprot,
                                 uint64_t total_residues, const plaac_params *points, uint32_t npoints,
                                 plaac_row *const *d_rows, const plaac_tracks *d_tracks, void *stream_) {
    if (nprot == 0 || npoints == 0) return PLAAC_OK;
    if (!d_offsets || !d_rows || !points || (!d_codes && total_residues))
        return fail(ctx, PLAAC_ERR_ARG, "null device buffer");
    if ((uintptr_t)d_codes & 15u) return fail(ctx, PLAAC_ERR_ARG, "d_codes must be 16-byte aligned");
    if (d_tracks && npoints != 1) return fail(ctx, PLAAC_ERR_ARG, "tracks are not available in sweeps");
    for (uint32_t i = 0; i < npoints; ++i) {
        if (!d_rows[i]) return fail(ctx, PLAAC_ERR_ARG, "null row array");
        if (const char *why = check_params(points[i])) return fail(ctx, PLAAC_ERR_ARG, why);
    }
    TrackPtrs tp{};
    if (d_tracks) {
        tp = TrackPtrs{d_tracks->vit,   d_tracks->map,  d_tracks->charge,     d_tracks->hydro,
                       d_tracks->fi,    d_tracks->plaacllr, d_tracks->papa,   d_tracks->fix2,
                       d_tracks->plaacllrx2, d_tracks->papax2, d_tracks->post0, d_tracks->post1};
        const void *all[] = {tp.vit, tp.map, tp.charge, tp.hydro, tp.fi, tp.plaacllr,
                             tp.papa, tp.fix2, tp.plaacllrx2, tp.papax2, tp.post0, tp.post1};
        for (const void *q : all)
            if (!q) return fail(ctx, PLAAC_ERR_ARG, "tracks struct has a null array");
    }
    PL_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;

    // ---- group the points: same tables up to the core length
    struct Group {
        uint32_t first;
        std::vector<uint32_t> members;
    };
    std::vector<Group> groups;
    for (uint32_t i = 0; i < npoints; ++i) {
        bool placed = false;
        for (Group &g : groups) {
            plaac_params a = points[g.first], b = points[i];
            a.corelength = b.corelength = 0;
            if (std::memcmp(&a, &b, sizeof a) == 0) {
                g.members.push_back(i);
                placed = true;
                break;
            }
        }
        if (!placed) groups.push_back(Group{i, {i}});
    }
    const size_t ng = groups.size();

    plaac_status rc;
    if ((rc = grow(ctx, ctx->d_neff, ctx->cap_prot, (size_t)nprot)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, ctx->d_order, ctx->cap_order, (size_t)nprot)) != PLAAC_OK) return rc;
    const uint32_t ngroups = (nprot + 63u) / 64u;
    if ((rc = grow(ctx, ctx->d_grow, ctx->cap_grow, (size_t)ngroups + 1)) != PLAAC_OK) return rc;
    if (!d_tracks && npoints == 1 && (rc = grow(ctx, ctx->d_lat, ctx->cap_lat, 2 * (size_t)nprot)) != PLAAC_OK) return rc;
    if (!d_tracks && ctx->kb_filter) { // lists of the filter form of the window kernel
        const size_t kb_blocks = ((size_t)nprot + KB_PROTEINS_PER_BLOCK - 1) / KB_PROTEINS_PER_BLOCK;
        if ((rc = grow(ctx, ctx->d_clist, ctx->cap_clist, kb_blocks * KB_PROTEINS_PER_BLOCK)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, ctx->d_ccount, ctx->cap_ccount, kb_blocks)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, ctx->d_fblist, ctx->cap_fblist, (size_t)nprot)) != PLAAC_OK) return rc;
    }
    // device tables: slot 0 keeps the ctx parameters (single-point calls), sweep groups use slots 1..ng
    const DevTables *gtab0 = ctx->d_tab;
    if (!(npoints == 1 && std::memcmp(&points[0], &ctx->params, sizeof(plaac_params)) == 0)) {
        if ((rc = grow(ctx, ctx->d_tabs, ctx->cap_tabs, ng)) != PLAAC_OK) return rc;
        std::vector<DevTables> host(ng);
        for (size_t g = 0; g < ng; ++g) fill_tables(points[groups[g].first], host[g]);
        PL_HIP(ctx, hipMemcpyAsync(ctx->d_tabs, host.data(), sizeof(DevTables) * ng, hipMemcpyHostToDevice, st));
        PL_HIP(ctx, hipStreamSynchronize(st)); // `host` is a temporary
        gtab0 = ctx->d_tabs;
    }
    while (ctx->gev.size() < ng) {
        hipEvent_t e = nullptr;
        PL_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->gev.push_back(e);
    }
    // The groups of a sweep are independent of each other: every group gets its own three side streams (and its
    // own traceback-bit buffer), so that the long serial chains of all groups advance together instead of one
    // group's tail after the other's.
    while (!ctx->serial && ctx->gstreams.size() < 3 * (ng - 1)) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        hipStream_t a = nullptr;
        PL_HIP(ctx, hipStreamCreateWithPriority(&a, hipStreamNonBlocking, greatest));
        ctx->gstreams.push_back(a);
        hipEvent_t e = nullptr;
        PL_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->gjev.push_back(e);
    }

    hipEvent_t *evs = ctx->ev[ctx->ncalls % plaac_ctx::EV_SETS];
    enum { E_START = 0, E_PLAN = 1, E_VIT = 2, E_FWD = 4, E_WIN = 6, E_TRK = 8, E_JOIN = 10, E_PACK = 11, E_BWD = 13 };
    hipStream_t sv = ctx->serial ? st : ctx->aux[0], sf = ctx->serial ? st : ctx->aux[1],
                sw = ctx->serial ? st : ctx->aux[2], sb = ctx->serial ? st : ctx->aux[3],
                sw2 = ctx->serial ? st : ctx->aux[4];

    // K-B base of a group: an earlier group whose window tracks differ only through the llr table (another alpha of
    // a sweep); such a group needs PAPAllr / PAPAllr2 at the known PAPA centre only (k_llr_at_centre)
    auto kb_base = [&](size_t g) -> long {
        const plaac_params &P = points[groups[g].first];
        if (P.ww3 / 2 > LLRAT_MAXW) return -1;
        for (size_t h = 0; h < g; ++h) {
            const plaac_params &Q = points[groups[h].first];
            if (P.ww1 == Q.ww1 && P.ww2 == Q.ww2 && P.ww3 == Q.ww3 && P.adjustprolines == Q.adjustprolines &&
                std::memcmp(P.cc, Q.cc, sizeof P.cc) == 0 && std::memcmp(P.hydro2, Q.hydro2, sizeof P.hydro2) == 0 &&
                std::memcmp(P.charge, Q.charge, sizeof P.charge) == 0 &&
                std::memcmp(P.lodpapa, Q.lodpapa, sizeof P.lodpapa) == 0)
                return (long)h;
        }
        return -1;
    };
    auto launch_tracks = [&](size_t g) -> plaac_status { // K-B: needs only the order, not the packed copy
        if (!d_tracks) {
            const long base = kb_base(g);
            if (base >= 0) {
                hipLaunchKernelGGL(k_llr_at_centre, dim3((nprot + 3u) / 4u), dim3(256), 0, st, d_codes, d_offsets,
                                   ctx->d_neff, nprot, gtab0 + g, d_rows[groups[(size_t)base].first],
                                   d_rows[groups[g].first]);
                return PLAAC_OK;
            }
        }
        const plaac_params &P = points[groups[g].first];
        const DevTables *tab = gtab0 + g;
        plaac_row *rows = d_rows[groups[g].first];
        const int wmax = std::max(P.ww1 / 2, std::max(P.ww2 / 2, P.ww3 / 2));
        const bool fast20 = P.ww1 / 2 == TW && P.ww2 / 2 == TW && P.ww3 / 2 == TW && !ctx->generic_tracks;
        if (g == 0) PL_HIP(ctx, hipEventRecord(evs[E_TRK], st));
#define LAUNCH_KB(RING)                                                                                            \
    do {                                                                                                           \
        if (d_tracks)                                                                                              \
            hipLaunchKernelGGL((k_tracks<RING, true>), dim3(nprot), dim3(64), 0, st, d_codes, d_offsets,           \
                               ctx->d_neff, ctx->d_order, nprot, tab, rows, tp);                                   \
        else                                                                                                       \
            hipLaunchKernelGGL((k_tracks<RING, false>), dim3(nprot), dim3(64), 0, st, d_codes, d_offsets,          \
                               ctx->d_neff, ctx->d_order, nprot, tab, rows, tp);                                   \
    } while (0)
        // the stream form keeps 32 proteins on one int32 position axis; a batch with a protein of >= 65535 residues is
        // left to the one-protein-at-a-time form. The planner raises a device flag for such a batch and both kernels are
        // enqueued: the one the flag rules out returns at once (no host round trip before the window kernel starts).
        if (fast20) {
            const unsigned kb_grid = (nprot + KB_PROTEINS_PER_BLOCK - 1) / KB_PROTEINS_PER_BLOCK;
            const uint32_t *huge = ctx->d_hist + LEN_BINS;
            const uint32_t only_if_huge = ctx->per_protein_tracks ? 0u : 1u; // PLAAC_KB_PER_PROTEIN=1: always this form
            if (!ctx->per_protein_tracks) {
                if (d_tracks) {
                    hipLaunchKernelGGL(k_tracks20s<true>, dim3(kb_grid), dim3(64), 0, st, d_codes, ctx->d_order, nprot,
                                       total_residues, tab, ctx->d_divtab, rows, tp, huge);
                } else if (ctx->kb_filter) {
                    // summary mode: decisions from error-bounded prefix sums, exact values at the chosen centre only,
                    // whatever the bounds cannot decide goes to the exact per-protein kernel through the fallback list
                    PL_HIP(ctx, hipMemsetAsync(ctx->d_fbcount, 0, sizeof(uint32_t), st));
                    hipLaunchKernelGGL(k_tracks20f, dim3(kb_grid), dim3(64), 0, st, d_codes, ctx->d_order, nprot,
                                       total_residues, tab, ctx->d_divtab, rows, huge, ctx->d_clist, ctx->d_ccount,
                                       ctx->d_fblist, ctx->d_fbcount);
                    hipLaunchKernelGGL(k_refine_centres, dim3(kb_grid), dim3(256), 0, st, d_codes, ctx->d_order,
                                       total_residues, tab, rows, huge, ctx->d_clist, ctx->d_ccount);
                    hipLaunchKernelGGL(k_tracks20<false>, dim3(std::min(kb_grid, 4096u)), dim3(64), 0, st, d_codes,
                                       d_offsets, ctx->d_neff, ctx->d_order, nprot, total_residues, tab, rows, tp, huge,
                                       0u, ctx->d_fblist, ctx->d_fbcount);
                } else {
                    hipLaunchKernelGGL(k_tracks20s<false>, dim3(kb_grid), dim3(64), 0, st, d_codes, ctx->d_order, nprot,
                                       total_residues, tab, ctx->d_divtab, rows, tp, huge);
                }
            }
            if (d_tracks)
                hipLaunchKernelGGL(k_tracks20<true>, dim3(kb_grid), dim3(64), 0, st, d_codes, d_offsets, ctx->d_neff,
                                   ctx->d_order, nprot, total_residues, tab, rows, tp, huge, only_if_huge, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
            else
                hipLaunchKernelGGL(k_tracks20<false>, dim3(kb_grid), dim3(64), 0, st, d_codes, d_offsets, ctx->d_neff,
                                   ctx->d_order, nprot, total_residues, tab, rows, tp, huge, only_if_huge, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
        } else if (wmax <= 32) LAUNCH_KB(128);
        else if (wmax <= 96) LAUNCH_KB(256);
        else LAUNCH_KB(1024);
#undef LAUNCH_KB
        if (g == 0) PL_HIP(ctx, hipEventRecord(evs[E_TRK + 1], st));
        return PLAAC_OK;
    };

    // the ctx's plan / scratch buffers are shared by consecutive calls: order this call after the previous one even
    // when the caller hands in a different stream
    if (ctx->ncalls > 0)
        PL_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[(ctx->ncalls - 1) % plaac_ctx::EV_SETS][E_JOIN], 0));
    PL_HIP(ctx, hipEventRecord(evs[E_START], st));
    PL_HIP(ctx, hipMemsetAsync(ctx->d_fbcount, 0, sizeof(uint32_t), st)); // plaac_last_exact_fallbacks: this call's count
    PL_HIP(ctx, hipMemsetAsync(ctx->d_hist, 0, sizeof(uint32_t) * (LEN_BINS + 1), st));
    const unsigned pb = (nprot + 255u) / 256u;
    const unsigned plb = (nprot + PLAN_THREADS * PLAN_ITEMS - 1) / (PLAN_THREADS * PLAN_ITEMS);
    hipLaunchKernelGGL(k_plan_lengths, dim3(plb), dim3(PLAN_THREADS), 0, st, d_codes, d_offsets, nprot, ctx->d_neff,
                       ctx->d_hist);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, ctx->d_hist);
    hipLaunchKernelGGL(k_plan_scatter, dim3(plb), dim3(PLAN_THREADS), 0, st, ctx->d_neff, nprot, ctx->d_hist,
                       d_offsets, ctx->d_order);
    PL_HIP(ctx, hipEventRecord(evs[E_PLAN], st));
    // The three K-A roles and K-B are independent given the plan: fork the K-A side onto high-priority streams
    // so the long serial chains (which set the wall time) overlap each other and the throughput-bound window
    // kernel. K-B starts right away on the caller's stream; the packing of the K-A input runs beside it.
    if (!ctx->serial) {
        if ((rc = launch_tracks(0)) != PLAAC_OK) return rc;
        PL_HIP(ctx, hipStreamWaitEvent(sv, evs[E_PLAN], 0));
    }
    // group rows of the interleaved copy; their total is the one value the host needs back (buffer sizes)
    PL_HIP(ctx, hipEventRecord(evs[E_PACK], sv));
    hipLaunchKernelGGL(k_group_rows, dim3((ngroups + 255u) / 256u), dim3(256), 0, sv, ctx->d_neff, ctx->d_order, nprot,
                       ngroups, ctx->d_grow);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(256), 0, sv, ctx->d_grow, ngroups);
    PL_HIP(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_grow + ngroups, sizeof(uint32_t), hipMemcpyDeviceToHost, sv));
    PL_HIP(ctx, hipMemcpyAsync(ctx->h_pin + 2, ctx->d_grow + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, sv));
    PL_HIP(ctx, hipStreamSynchronize(sv));
    const size_t total_rows = ctx->h_pin[0];
    // Is this batch bound by the serial chain of its longest protein (16-residue rows of the first wave-group x ~150 ns
    // per residue) rather than by throughput (~12 ps per residue)? Then the lane-per-protein kernels take the forms
    // that shorten one wave's chain (k_win as two kernels) at the price of a few more instructions in total.
    const bool latency_mode =
        !ctx->serial && !d_tracks && npoints == 1 &&
        (ctx->latency_mode >= 0 ? ctx->latency_mode == 1 : (uint64_t)ctx->h_pin[2] * 384000ull > total_residues);
    if (latency_mode && ctx->h_pin[2] >= CORE_LONG_ROWS) { // scratch of k_core_*: the rows of the long wave-groups
        const size_t lrows = std::min<size_t>(total_rows, (size_t)CORE_MAX_GROUPS * ctx->h_pin[2]);
        if ((rc = grow(ctx, ctx->d_corep, ctx->cap_corep, lrows * 1024u)) != PLAAC_OK) return rc;
        char *&cp = reinterpret_cast<char *&>(ctx->d_corepart);
        if ((rc = grow(ctx, cp, ctx->cap_corepart, lrows * 64u * sizeof(CorePart))) != PLAAC_OK) return rc;
    }
    if ((rc = grow(ctx, ctx->d_packed, ctx->cap_packed, total_rows * 64u + 64u)) != PLAAC_OK) return rc;
    const size_t bits_stride = total_rows * 64u + 64u; // one traceback-bit buffer per group
    if ((rc = grow(ctx, ctx->d_bits, ctx->cap_bits, bits_stride * ng)) != PLAAC_OK) return rc;
    if (d_tracks) {
        if ((rc = grow(ctx, ctx->d_fwd, ctx->cap_fwd, total_rows * 16u * 64u + 64u)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, ctx->d_bwd, ctx->cap_bwd, total_rows * 16u * 64u + 64u)) != PLAAC_OK) return rc;
    }
    hipLaunchKernelGGL(k_pack, dim3((nprot + 15u) / 16u), dim3(256), 0, sv, d_codes, d_offsets, ctx->d_neff,
                       ctx->d_order, nprot, total_residues, ctx->d_grow, ctx->d_packed);
    PL_HIP(ctx, hipEventRecord(evs[E_PACK + 1], sv));
    if (!ctx->serial) {
        for (hipStream_t a : {sf, sw, sb, sw2}) PL_HIP(ctx, hipStreamWaitEvent(a, evs[E_PACK + 1], 0));
        for (size_t k = 0; k < 3 * (ng - 1); ++k) PL_HIP(ctx, hipStreamWaitEvent(ctx->gstreams[k], evs[E_PACK + 1], 0));
    }
    const unsigned ab = (nprot + KA_THREADS - 1) / KA_THREADS;
    // track mode: the backward recurrence is a chain of its own, beside the forward one
    PL_HIP(ctx, hipEventRecord(evs[E_BWD], sb));
    if (d_tracks)
        hipLaunchKernelGGL(k_bwd, dim3(ab), dim3(KA_THREADS), 0, sb, d_offsets, ctx->d_neff, ctx->d_order, nprot, gtab0,
                           ctx->d_packed, ctx->d_grow, ctx->d_bwd);
    PL_HIP(ctx, hipEventRecord(evs[E_BWD + 1], sb));

    const hipStream_t sv0 = sv, sf0 = sf, sw0 = sw;
    for (size_t g = 0; g < ng; ++g) {
        const Group &G = groups[g];
        const DevTables *tab = gtab0 + g;
        plaac_row *rows0 = d_rows[G.first];
        const bool timed = g == 0;
        // streams of this group
        hipStream_t sv = sv0, sf = sf0, sw = sw0;
        if (g > 0 && !ctx->serial) {
            sv = ctx->gstreams[3 * (g - 1)];
            sf = ctx->gstreams[3 * (g - 1) + 1];
            sw = ctx->gstreams[3 * (g - 1) + 2];
        }
        uint32_t *gbits = ctx->d_bits + bits_stride * g;
        // K-B of this group (group 0 was launched before the host round trip; serialised mode launches it last)
        if (g > 0 && !ctx->serial) {
            if ((rc = launch_tracks(g)) != PLAAC_OK) return rc;
        }
        // forward pass: once per group
        if (timed) PL_HIP(ctx, hipEventRecord(evs[E_FWD], sf));
        if (latency_mode)
            hipLaunchKernelGGL(k_fwd_pair, dim3((nprot + KA_THREADS / 2 - 1) / (KA_THREADS / 2)), dim3(KA_THREADS), 0, sf,
                               ctx->d_order, nprot, tab, ctx->d_packed, ctx->d_grow, ctx->d_lat);
        else if (d_tracks)
            hipLaunchKernelGGL(k_fwd<true>, dim3(ab), dim3(KA_THREADS), 0, sf, d_codes, d_offsets, ctx->d_neff,
                               ctx->d_order, nprot, tab, ctx->d_packed, ctx->d_grow, rows0, ctx->d_fwd);
        else
            hipLaunchKernelGGL(k_fwd<false>, dim3(ab), dim3(KA_THREADS), 0, sf, d_codes, d_offsets, ctx->d_neff,
                               ctx->d_order, nprot, tab, ctx->d_packed, ctx->d_grow, rows0, (double2 *)nullptr);
        if (timed) PL_HIP(ctx, hipEventRecord(evs[E_FWD + 1], sf));
        // Viterbi / windows: up to MAXC core lengths per launch
        for (size_t m0 = 0; m0 < G.members.size(); m0 += MAXC) {
            const int nc = (int)std::min<size_t>(MAXC, G.members.size() - m0);
            SweepTargets tg{};
            for (int k = 0; k < MAXC; ++k) {
                const uint32_t idx = G.members[m0 + (size_t)std::min(k, nc - 1)];
                tg.c[k] = (uint32_t)points[idx].corelength;
                tg.rows[k] = d_rows[idx];
            }
            const bool t0 = timed && m0 == 0;
            tg.stop_after = ctx->vit_stop;
            if (t0) PL_HIP(ctx, hipEventRecord(evs[E_VIT], sv));
#define LAUNCH_VIT(NC)                                                                                             \
    hipLaunchKernelGGL((k_vit<NC>), dim3(ab), dim3(KA_THREADS), 0, sv, d_codes, d_offsets, ctx->d_neff, ctx->d_order, \
                       nprot, tab, ctx->d_packed, ctx->d_grow, gbits, tg)
            if (latency_mode) {
                hipLaunchKernelGGL((k_vit<1, true>), dim3(ab), dim3(KA_THREADS), 0, sv, d_codes, d_offsets, ctx->d_neff,
                                   ctx->d_order, nprot, tab, ctx->d_packed, ctx->d_grow, gbits, tg);
                if (tg.stop_after == 0u && ctx->h_pin[2] >= CORE_LONG_ROWS) { // some group is long: its core window
                    const unsigned lg = std::min<unsigned>(ngroups, CORE_MAX_GROUPS);
                    hipLaunchKernelGGL(k_core_chain, dim3(lg), dim3(64), 0, sv, ctx->d_order, nprot, tab, ctx->d_packed,
                                       ctx->d_grow, gbits, ctx->d_corep);
                    const size_t lrows = std::min<size_t>(total_rows, (size_t)CORE_MAX_GROUPS * ctx->h_pin[2]);
                    hipLaunchKernelGGL(k_core_eval, dim3((unsigned)lrows), dim3(64), 0, sv, ctx->d_order, nprot, ngroups,
                                       ctx->d_grow, ctx->d_corep, (CorePart *)ctx->d_corepart, tg.c[0]);
                    hipLaunchKernelGGL(k_core_reduce, dim3(lg), dim3(64), 0, sv, d_codes, ctx->d_order, nprot, tab,
                                       ctx->d_grow, gbits, (const CorePart *)ctx->d_corepart, tg.rows[0], tg.c[0]);
                }
            } else
            switch (nc) {
            case 1: LAUNCH_VIT(1); break;
            case 2: LAUNCH_VIT(2); break;
            case 3: LAUNCH_VIT(3); break;
            default: LAUNCH_VIT(4); break;
            }
#undef LAUNCH_VIT
            if (t0) PL_HIP(ctx, hipEventRecord(evs[E_VIT + 1], sv));
            if (t0) PL_HIP(ctx, hipEventRecord(evs[E_WIN], sw));
#define LAUNCH_WIN(NC, ROLE, STREAM)                                                                               \
    hipLaunchKernelGGL((k_win<NC, ROLE>), dim3(ab), dim3(KA_THREADS), 0, STREAM, d_codes, d_offsets, ctx->d_neff,  \
                       ctx->d_order, nprot, tab, ctx->d_packed, ctx->d_grow, tg,                                   \
                       ctx->d_lat ? ctx->d_lat + nprot : (double *)nullptr)
            if (latency_mode) { // three thirds side by side (LLR window | MW window | means + hmm0's running sum)
                LAUNCH_WIN(1, 2, sw);
                LAUNCH_WIN(1, 4, sw2);
                LAUNCH_WIN(1, 5, sb); // the backward stream is idle in summary mode
            } else {
                switch (nc) {
                case 1: LAUNCH_WIN(1, 0, sw); break;
                case 2: LAUNCH_WIN(2, 0, sw); break;
                case 3: LAUNCH_WIN(3, 0, sw); break;
                default: LAUNCH_WIN(4, 0, sw); break;
                }
            }
#undef LAUNCH_WIN
            if (t0) PL_HIP(ctx, hipEventRecord(evs[E_WIN + 1], sw));
        }
        if (ctx->serial) {
            if ((rc = launch_tracks(g)) != PLAAC_OK) return rc;
        }
        // fields that do not depend on the core length: copy from the group's first row array to the others
        if (G.members.size() > 1) {
            if (!ctx->serial) {
                PL_HIP(ctx, hipEventRecord(ctx->gev[g], sf));
                PL_HIP(ctx, hipStreamWaitEvent(st, ctx->gev[g], 0));
            }
            for (size_t m0 = 1; m0 < G.members.size(); m0 += MAXC - 1) {
                const int nd = (int)std::min<size_t>(MAXC - 1, G.members.size() - m0);
                SweepTargets tg{};
                for (int k = 1; k <= nd; ++k) tg.rows[k] = d_rows[G.members[m0 + (size_t)k - 1]];
                hipLaunchKernelGGL(k_replicate, dim3(pb), dim3(256), 0, st, rows0, tg, nd, nprot);
            }
        }
    }
    if (!ctx->serial) {
        // join: everything enqueued on the side streams so far
        PL_HIP(ctx, hipEventRecord(ctx->jev[0], sv));
        PL_HIP(ctx, hipEventRecord(ctx->jev[1], sf));
        PL_HIP(ctx, hipEventRecord(ctx->jev[2], sw));
        PL_HIP(ctx, hipEventRecord(ctx->jev[3], sb));
        PL_HIP(ctx, hipEventRecord(ctx->jev[4], sw2));
        for (hipEvent_t e : ctx->jev) PL_HIP(ctx, hipStreamWaitEvent(st, e, 0));
        for (size_t k = 0; k < 3 * (ng - 1); ++k) {
            PL_HIP(ctx, hipEventRecord(ctx->gjev[k], ctx->gstreams[k]));
            PL_HIP(ctx, hipStreamWaitEvent(st, ctx->gjev[k], 0));
        }
    }
    if (latency_mode)
        hipLaunchKernelGGL(k_finish, dim3(pb), dim3(256), 0, st, d_rows[0], ctx->d_lat, ctx->d_lat + nprot, nprot);
    if (d_tracks && total_rows) // posteriors, MAP, Viterbi bytes: needs k_fwd, k_bwd and the path bits (k_vit)
        hipLaunchKernelGGL(k_post, dim3((unsigned)total_rows), dim3(64), 0, st, d_offsets, ctx->d_neff, ctx->d_order,
                           nprot, ngroups, ctx->d_grow, gtab0, ctx->d_fwd, ctx->d_bwd, ctx->d_bits, tp);
    PL_HIP(ctx, hipEventRecord(evs[E_JOIN], st));
    PL_HIP(ctx, hipGetLastError());
    ctx->ncalls++;
    return PLAAC_OK;
}

plaac_status plaac_score_device(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets, uint32_t nprot,
                                uint64_t total_residues, plaac_row *d_rows, const plaac_tracks *d_tracks,
                                void *stream_) {
    if (!ctx) return PLAAC_ERR_ARG;
    plaac_row *rows1[1] = {d_rows};
    return score_points(ctx, d_codes, d_offsets, nprot, total_residues, &ctx->params, 1, rows1, d_tracks, stream_);
}

plaac_status plaac_score_sweep_device(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets,
                                      uint32_t nprot, uint64_t total_residues, const plaac_params *points,
                                      uint32_t npoints, plaac_row *const *d_rows, void *stream_) {
    if (!ctx) return PLAAC_ERR_ARG;
    return score_points(ctx, d_codes, d_offsets, nprot, total_residues, points, npoints, d_rows, nullptr, stream_);
}

plaac_status plaac_timings_mean(plaac_ctx *ctx, uint32_t ncalls, float ms[8]) {
    if (!ctx || !ms) return PLAAC_ERR_ARG;
    if (ctx->ncalls == 0) return fail(ctx, PLAAC_ERR_ARG, "no scored batch to time yet");
    if (ncalls == 0) ncalls = 1;
    if (ncalls > plaac_ctx::EV_SETS) ncalls = plaac_ctx::EV_SETS;
    if (ncalls > ctx->ncalls) ncalls = (uint32_t)ctx->ncalls;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    static const int pairs[8][2] = {{0, 10}, {0, 1}, {2, 3}, {4, 5}, {6, 7}, {8, 9}, {11, 12}, {13, 14}};
    for (uint32_t k = 0; k < ncalls; ++k) {
        hipEvent_t *evs = ctx->ev[(ctx->ncalls - 1 - k) % plaac_ctx::EV_SETS];
        PL_HIP(ctx, hipEventSynchronize(evs[10]));
        for (int i = 0; i < 8; ++i) {
            float t = 0.f;
            PL_HIP(ctx, hipEventElapsedTime(&t, evs[pairs[i][0]], evs[pairs[i][1]]));
            acc[i] += t;
        }
    }
    for (int i = 0; i < 8; ++i) ms[i] = (float)(acc[i] / ncalls);
    return PLAAC_OK;
}

plaac_status plaac_last_timings(plaac_ctx *ctx, float ms[8]) { return plaac_timings_mean(ctx, 1, ms); }

plaac_status plaac_last_exact_fallbacks(plaac_ctx *ctx, uint32_t *count) {
    if (!ctx || !count) return PLAAC_ERR_ARG;
    *count = 0;
    if (ctx->ncalls == 0) return PLAAC_OK;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_HIP(ctx, hipEventSynchronize(ctx->ev[(ctx->ncalls - 1) % plaac_ctx::EV_SETS][10 /* E_JOIN */]));
    PL_HIP(ctx, hipMemcpy(count, ctx->d_fbcount, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PLAAC_OK;
}

plaac_status plaac_histogram_device(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets,
                                    uint32_t nprot, int64_t *d_counts, void *stream_) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (!d_counts || (nprot && (!d_codes || !d_offsets))) return fail(ctx, PLAAC_ERR_ARG, "null device buffer");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    PL_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int64_t) * NAA, st));
    if (nprot) {
        unsigned blocks = (nprot + 3u) / 4u;
        if (blocks > 256u * 16u) blocks = 256u * 16u;
        hipLaunchKernelGGL(k_hist, dim3(blocks), dim3(256), 0, st, d_codes, d_offsets, nprot,
                           (unsigned long long *)d_counts);
        PL_HIP(ctx, hipGetLastError());
    }
    return PLAAC_OK;
}

static unsigned copy_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("PLAAC_THREADS")) n = (unsigned)std::max(1, std::atoi(e));
    return std::max(1u, std::min(n, 4u));
}

static void parallel_memcpy(void *dst, const void *src, size_t bytes) {
    const unsigned nt = bytes < (4u << 20) ? 1u : copy_threads();
    if (nt == 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t b = bytes * t / nt, e = bytes * (t + 1) / nt;
        pool.emplace_back([=] { std::memcpy((char *)dst + b, (const char *)src + b, e - b); });
    }
    for (auto &th : pool) th.join();
}

static plaac_status ensure_stage(plaac_ctx *ctx) {
    for (int i = 0; i < 2; ++i) {
        if (!ctx->h_stage[i]) PL_HIP(ctx, hipHostMalloc((void **)&ctx->h_stage[i], plaac_ctx::STAGE_BYTES, hipHostMallocDefault));
        if (!ctx->stage_ev[i]) PL_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
    }
    return PLAAC_OK;
}

// host (pageable) -> device through two pinned buffers: the copy into pinned memory of chunk k+1 overlaps the
// DMA of chunk k. Small copies go directly.
static plaac_status copy_in(plaac_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes < (1u << 20)) {
        PL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return PLAAC_OK;
    }
    plaac_status rc = ensure_stage(ctx);
    if (rc != PLAAC_OK) return rc;
    size_t done = 0;
    for (int k = 0; done < bytes; ++k) {
        const int b = k & 1;
        const size_t n = std::min(plaac_ctx::STAGE_BYTES, bytes - done);
        if (k >= 2) PL_HIP(ctx, hipEventSynchronize(ctx->stage_ev[b])); // DMA out of this buffer has finished
        parallel_memcpy(ctx->h_stage[b], (const char *)src + done, n);
        PL_HIP(ctx, hipMemcpyAsync((char *)dst + done, ctx->h_stage[b], n, hipMemcpyHostToDevice, st));
        PL_HIP(ctx, hipEventRecord(ctx->stage_ev[b], st));
        done += n;
    }
    PL_HIP(ctx, hipStreamSynchronize(st)); // the staging buffers are free again when this returns
    return PLAAC_OK;
}

// device -> host (pageable), same scheme; returns with the data in place
static plaac_status copy_out(plaac_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes < (1u << 20)) {
        PL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
        PL_HIP(ctx, hipStreamSynchronize(st));
        return PLAAC_OK;
    }
    plaac_status rc = ensure_stage(ctx);
    if (rc != PLAAC_OK) return rc;
    const size_t nchunks = (bytes + plaac_ctx::STAGE_BYTES - 1) / plaac_ctx::STAGE_BYTES;
    auto issue = [&](size_t k) -> plaac_status {
        const size_t o = k * plaac_ctx::STAGE_BYTES, n = std::min(plaac_ctx::STAGE_BYTES, bytes - o);
        PL_HIP(ctx, hipMemcpyAsync(ctx->h_stage[k & 1], (const char *)src + o, n, hipMemcpyDeviceToHost, st));
        PL_HIP(ctx, hipEventRecord(ctx->stage_ev[k & 1], st));
        return PLAAC_OK;
    };
    if ((rc = issue(0)) != PLAAC_OK) return rc;
    for (size_t k = 0; k < nchunks; ++k) {
        if (k + 1 < nchunks && (rc = issue(k + 1)) != PLAAC_OK) return rc; // other buffer: drained in iteration k-1
        PL_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k & 1]));
        const size_t o = k * plaac_ctx::STAGE_BYTES, n = std::min(plaac_ctx::STAGE_BYTES, bytes - o);
        parallel_memcpy((char *)dst + o, ctx->h_stage[k & 1], n);
    }
    return PLAAC_OK;
}

static plaac_status stage_in(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                             uint64_t *total_out) {
    if (!offsets) return fail(ctx, PLAAC_ERR_ARG, "null offsets");
    if (offsets[0] != 0) return fail(ctx, PLAAC_ERR_ARG, "offsets[0] must be 0");
    for (uint32_t p = 0; p < nprot; ++p)
        if (offsets[p + 1] < offsets[p]) return fail(ctx, PLAAC_ERR_ARG, "offsets must be non-decreasing");
    const uint64_t total = offsets[nprot];
    if (total && !codes) return fail(ctx, PLAAC_ERR_ARG, "null codes");
    plaac_status rc;
    if ((rc = grow(ctx, ctx->d_codes, ctx->cap_codes, (size_t)total + 64)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, ctx->d_offsets, ctx->cap_offs, (size_t)nprot + 1)) != PLAAC_OK) return rc;
    if (total && (rc = copy_in(ctx, ctx->d_codes, codes, total, ctx->stream)) != PLAAC_OK) return rc;
    if ((rc = copy_in(ctx, ctx->d_offsets, offsets, sizeof(uint64_t) * ((size_t)nprot + 1), ctx->stream)) != PLAAC_OK)
        return rc;
    // codes must be 0..21 (they index the kernels' tables): checked on the device, the upload is there anyway
    if (total) {
        if (!ctx->d_flag) PL_HIP(ctx, hipMalloc((void **)&ctx->d_flag, sizeof(uint32_t)));
        PL_HIP(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(uint32_t), ctx->stream));
        const unsigned vb = (unsigned)std::min<uint64_t>(((total >> 4) + 255u) / 256u + 1u, 2048u);
        hipLaunchKernelGGL(k_validate, dim3(vb), dim3(256), 0, ctx->stream, ctx->d_codes, total, ctx->d_flag);
        PL_HIP(ctx, hipMemcpyAsync(ctx->h_pin + 1, ctx->d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_pin[1]) return fail(ctx, PLAAC_ERR_ARG, "residue code > 21");
    }
    *total_out = total;
    return PLAAC_OK;
}

plaac_status plaac_histogram(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                             int64_t counts[PLAAC_NAA]) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (!counts) return fail(ctx, PLAAC_ERR_ARG, "null counts");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t total = 0;
    plaac_status rc = stage_in(ctx, codes, offsets, nprot, &total);
    if (rc != PLAAC_OK) return rc;
    rc = plaac_histogram_device(ctx, ctx->d_codes, ctx->d_offsets, nprot, (int64_t *)ctx->d_counts, ctx->stream);
    if (rc != PLAAC_OK) return rc;
    PL_HIP(ctx, hipMemcpyAsync(counts, ctx->d_counts, sizeof(int64_t) * NAA, hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PLAAC_OK;
}

// score device-resident codes/offsets and bring rows (+ tracks) back to host buffers
static plaac_status score_resident_to_host(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets,
                                           uint32_t nprot, uint64_t total, plaac_row *rows,
                                           const plaac_tracks *tracks) {
    plaac_status rc;
    if ((rc = grow(ctx, ctx->d_rows, ctx->cap_rows, (size_t)nprot)) != PLAAC_OK) return rc;
    plaac_tracks dt{};
    if (tracks) {
        double *hd[10] = {tracks->charge, tracks->hydro,      tracks->fi,     tracks->plaacllr, tracks->papa,
                          tracks->fix2,   tracks->plaacllrx2, tracks->papax2, tracks->post0,    tracks->post1};
        for (double *q : hd)
            if (!q) return fail(ctx, PLAAC_ERR_ARG, "tracks struct has a null array");
        if (!tracks->vit || !tracks->map) return fail(ctx, PLAAC_ERR_ARG, "tracks struct has a null array");
        const size_t need = (size_t)total + 8;
        if (need > ctx->cap_trk || !ctx->d_trk8) {
            if (ctx->d_trk8) PL_HIP(ctx, hipFree(ctx->d_trk8));
            if (ctx->d_trk64) PL_HIP(ctx, hipFree(ctx->d_trk64));
            ctx->d_trk8 = nullptr;
            ctx->d_trk64 = nullptr;
            ctx->cap_trk = 0;
            PL_HIP(ctx, hipMalloc((void **)&ctx->d_trk8, 2 * need));
            PL_HIP(ctx, hipMalloc((void **)&ctx->d_trk64, 10 * need * sizeof(double)));
            ctx->cap_trk = need;
        }
        const size_t s = ctx->cap_trk;
        dt.vit = ctx->d_trk8;
        dt.map = ctx->d_trk8 + s;
        double *b = ctx->d_trk64;
        dt.charge = b;
        dt.hydro = b + s;
        dt.fi = b + 2 * s;
        dt.plaacllr = b + 3 * s;
        dt.papa = b + 4 * s;
        dt.fix2 = b + 5 * s;
        dt.plaacllrx2 = b + 6 * s;
        dt.papax2 = b + 7 * s;
        dt.post0 = b + 8 * s;
        dt.post1 = b + 9 * s;
        // entries of trimmed stops are unspecified: make them deterministic (0 / NaN)
        PL_HIP(ctx, hipMemsetAsync(ctx->d_trk8, 0, 2 * s, ctx->stream));
        PL_HIP(ctx, hipMemsetAsync(ctx->d_trk64, 0xff, 10 * s * sizeof(double), ctx->stream));
    }
    rc = plaac_score_device(ctx, d_codes, d_offsets, nprot, total, ctx->d_rows, tracks ? &dt : nullptr, ctx->stream);
    if (rc != PLAAC_OK) return rc;
    if ((rc = copy_out(ctx, rows, ctx->d_rows, sizeof(plaac_row) * (size_t)nprot, ctx->stream)) != PLAAC_OK) return rc;
    if (tracks && total) {
        const size_t nb = (size_t)total;
        uint8_t *h8[2] = {tracks->vit, tracks->map};
        uint8_t *d8[2] = {dt.vit, dt.map};
        for (int i = 0; i < 2; ++i)
            if ((rc = copy_out(ctx, h8[i], d8[i], nb, ctx->stream)) != PLAAC_OK) return rc;
        double *hd[10] = {tracks->charge, tracks->hydro,      tracks->fi,     tracks->plaacllr, tracks->papa,
                          tracks->fix2,   tracks->plaacllrx2, tracks->papax2, tracks->post0,    tracks->post1};
        double *dd[10] = {dt.charge, dt.hydro, dt.fi, dt.plaacllr, dt.papa, dt.fix2, dt.plaacllrx2, dt.papax2, dt.post0,
                          dt.post1};
        for (int i = 0; i < 10; ++i)
            if ((rc = copy_out(ctx, hd[i], dd[i], nb * sizeof(double), ctx->stream)) != PLAAC_OK) return rc;
    }
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PLAAC_OK;
}

plaac_status plaac_score(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                         plaac_row *rows, const plaac_tracks *tracks) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (nprot == 0) return PLAAC_OK;
    if (!rows) return fail(ctx, PLAAC_ERR_ARG, "null rows");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t total = 0;
    plaac_status rc = stage_in(ctx, codes, offsets, nprot, &total);
    if (rc != PLAAC_OK) return rc;
    return score_resident_to_host(ctx, ctx->d_codes, ctx->d_offsets, nprot, total, rows, tracks);
}

struct plaac_batch {
    plaac_ctx *ctx;
    uint8_t *d_codes;
    uint64_t *d_offsets;
    uint32_t nprot;
    uint64_t total;
};

plaac_status plaac_batch_upload(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                plaac_batch **out) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (!out) return fail(ctx, PLAAC_ERR_ARG, "null out");
    *out = nullptr;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t total = 0;
    // stage through the ctx buffers' validation, then hand the device copies over to the batch
    plaac_status rc = stage_in(ctx, codes, offsets, nprot, &total);
    if (rc != PLAAC_OK) return rc;
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    plaac_batch *b = new (std::nothrow) plaac_batch{ctx, ctx->d_codes, ctx->d_offsets, nprot, total};
    if (!b) return fail(ctx, PLAAC_ERR_NOMEM, "out of host memory");
    ctx->d_codes = nullptr; // ownership moves to the batch; the ctx allocates fresh staging buffers on demand
    ctx->d_offsets = nullptr;
    ctx->cap_codes = ctx->cap_offs = 0;
    *out = b;
    return PLAAC_OK;
}

plaac_status plaac_batch_histogram(plaac_batch *b, int64_t counts[PLAAC_NAA]) {
    if (!b || !b->ctx) return PLAAC_ERR_ARG;
    plaac_ctx *ctx = b->ctx;
    if (!counts) return fail(ctx, PLAAC_ERR_ARG, "null counts");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    plaac_status rc = plaac_histogram_device(ctx, b->d_codes, b->d_offsets, b->nprot, (int64_t *)ctx->d_counts, ctx->stream);
    if (rc != PLAAC_OK) return rc;
    PL_HIP(ctx, hipMemcpyAsync(counts, ctx->d_counts, sizeof(int64_t) * NAA, hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PLAAC_OK;
}

plaac_status plaac_batch_score(plaac_batch *b, plaac_row *rows, const plaac_tracks *tracks) {
    if (!b || !b->ctx) return PLAAC_ERR_ARG;
    plaac_ctx *ctx = b->ctx;
    if (b->nprot == 0) return PLAAC_OK;
    if (!rows) return fail(ctx, PLAAC_ERR_ARG, "null rows");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    return score_resident_to_host(ctx, b->d_codes, b->d_offsets, b->nprot, b->total, rows, tracks);
}

plaac_status plaac_batch_sweep(plaac_batch *b, const plaac_params *points, uint32_t npoints, plaac_row *const *rows) {
    if (!b || !b->ctx) return PLAAC_ERR_ARG;
    plaac_ctx *ctx = b->ctx;
    if (b->nprot == 0 || npoints == 0) return PLAAC_OK;
    if (!points || !rows) return fail(ctx, PLAAC_ERR_ARG, "null argument");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    plaac_status rc;
    if ((rc = grow(ctx, ctx->d_rows, ctx->cap_rows, (size_t)b->nprot * npoints)) != PLAAC_OK) return rc;
    std::vector<plaac_row *> drows(npoints);
    for (uint32_t i = 0; i < npoints; ++i) {
        if (!rows[i]) return fail(ctx, PLAAC_ERR_ARG, "null row array");
        drows[i] = ctx->d_rows + (size_t)i * b->nprot;
    }
    rc = score_points(ctx, b->d_codes, b->d_offsets, b->nprot, b->total, points, npoints, drows.data(), nullptr,
                      ctx->stream);
    if (rc != PLAAC_OK) return rc;
    for (uint32_t i = 0; i < npoints; ++i)
        PL_HIP(ctx, hipMemcpyAsync(rows[i], drows[i], sizeof(plaac_row) * (size_t)b->nprot, hipMemcpyDeviceToHost,
                                   ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PLAAC_OK;
}

void plaac_batch_free(plaac_batch *b) {
    if (!b) return;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
    }
    if (b->d_codes) (void)hipFree(b->d_codes);
    if (b->d_offsets) (void)hipFree(b->d_offsets);
    delete b;
}

} // extern "C"
