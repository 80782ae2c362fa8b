// plaac_kernels.hip — gfx950 (CDNA4) kernels + the device half of the C ABI (include/plaac_native.h).
//
// MUST be compiled with -ffp-contract=off: the reference (Java) never fuses a*b+c and the
// tie-breaks of the window searches depend on the exact rounding of every add.
//
// One translation unit; the device code lives in section files included below (inside one anonymous namespace):
//   kernels_tables_plan.hip.inc     DevTables; k_plan_lengths / _scan / _scatter: descending-length counting sort, so that
//                                   the 64 lanes of a wave run chains of similar length and the longest start first
//   kernels_chains.hip.inc          "K-A", ONE LANE PER PROTEIN: the order-sensitive serial fp64 chains (reference = plaac.java)
//                                     Viterbi + traceback                hmm.viterbidecodel :3077-3121
//                                     forward with the LUT log-sum-exp   hmm.posteriorl :3354-3375, logeapeb :1024-1047
//                                     hmm0 (degenerates to a running sum)  :795, SURVEY H4
//                                     MW / LLR fixed-width windows over prefix sums   hss2 :1206-1257 via :767-783
//                                     masked core window, PRD expansion, PRD score    :816-880
//                                     mean hydropathy / charge / FoldIndex            :4877-4885
//                                   k_group_rows / k_scan_u32 / k_pack (group-interleaved residue copy), k_vit, k_fwd, k_win,
//                                   k_bwd / k_post (track mode); latency forms for batches bound by their longest protein:
//                                   k_fwd_pair (two lanes per protein), k_win roles, k_core_chain / _eval / _reduce, k_finish
//   kernels_windows_exact.hip.inc   "K-B", position-parallel, exact: the window tracks of disorderreport (:4866-5068) as
//                                   fixed-order 41-term sums: k_tracks20s (32 proteins end to end on one position axis),
//                                   k_tracks20 (one protein at a time: fallback tier, huge proteins), k_tracks<RING> (any
//                                   window size); k_llr_at_centre (sweeps)
//   kernels_windows_filter.hip.inc  "K-B", summary mode: k_tracks20f (decisions from error-bounded prefix sums) and
//                                   k_refine_centres (the reported values at the chosen centre in the reference's order)
//   kernels_windows_lane.hip.inc    "K-B", summary mode, the filter tier with one lane per protein and sliding windows (k_tracksL)
//   kernels_misc.hip.inc            k_hist (countaas / isvalidprotein :1698-1739), k_validate
// Below them: the device half of the C ABI (contexts, streams, the scheduling of a scoring call).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <climits>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "plaac_native.h"

namespace {

static inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
    __asm__ __volatile__("pause");
#endif
}
#include "kernels_tables_plan.hip.inc"
#include "kernels_chains.hip.inc"
#include "kernels_windows_exact.hip.inc"
#include "kernels_windows_filter.hip.inc"
#include "kernels_windows_lane.hip.inc"
#include "kernels_misc.hip.inc"
#include "fasta_device.hip.inc"
#include "format_device.hip.inc"
#include "schedule.hip.inc"

constexpr int E_JOIN_IDX = sched::E_JOIN; // index of a call's join event in its set of timing events

} // namespace

// the latency-form chain kernels built a second time under another instruction-scheduling strategy (plaac_kernels_lat.hip)
namespace plaac_lat {
void launch_long(dim3 grid, hipStream_t s, const uint64_t *offsets, const uint32_t *neff, const void *order, uint32_t nprot,
                 const void *tab, const void *packed, const uint32_t *grow, const void *tg_bytes, double *lmarg, double *h0);
void launch_fwd_pair(dim3 grid, hipStream_t s, const void *order, uint32_t nprot, const void *tab, const void *packed,
                     const uint32_t *grow, double *lmarg);
void launch_vit_lat_ext(dim3 grid, hipStream_t s, const uint8_t *codes, const uint64_t *offsets, const uint32_t *neff, const void *order,
                        uint32_t nprot, const void *tab, const void *packed, const uint32_t *grow, uint32_t *bits, const void *tg_bytes,
                        double *vend);
size_t sizeof_sweep_targets();
size_t sizeof_dev_tables();
} // namespace plaac_lat
namespace {
// (test hook LAT_UNIT=0, read when a context is created: the main unit's copies of those kernels - A/B)
bool lat_unit_usable() {
    std::string v;
    const bool off = sched::knob_value("LAT_UNIT", v) && !v.empty() && v[0] == '0';
    return !off && plaac_lat::sizeof_sweep_targets() == sizeof(SweepTargets) && plaac_lat::sizeof_dev_tables() == sizeof(DevTables);
}
} // namespace

// ------------------------------------------------------------------------------------------------
// C ABI — device half
// ------------------------------------------------------------------------------------------------
// a text batch uploaded and parsed ahead of its scoring call (plaac_text_upload): the device buffers a pending slot would hold
struct plaac_text_batch {
    plaac_ctx *ctx = nullptr;
    char *d_text = nullptr;
    uint64_t *d_starts = nullptr;
    uint32_t *d_len = nullptr;
    uint8_t *d_blank = nullptr;
    FaExtent *d_ext = nullptr;
    unsigned long long *d_total = nullptr, *d_bsum = nullptr;
    uint8_t *d_codes = nullptr;
    uint64_t *d_offsets = nullptr;
    size_t cap_text = 0, cap_starts = 0, cap_len = 0, cap_blank = 0, cap_ext = 0, cap_bsum = 0, cap_codes = 0, cap_offs = 0;
    uint32_t nrec = 0;
    uint64_t total = 0;
};

struct plaac_ctx {
    int device = 0;
    int num_cus = 256;
    bool value_tolerance = false; // plaac_ctx_set_value_tolerance: k_refine_centres<., SLIDE>
    bool lat_unit = true;         // the latency-form chain kernels of the max-ilp unit (plaac_kernels_lat.hip) are usable
    hipStream_t stream = nullptr;
    DevTables *d_tab = nullptr;   // tables of ctx->params
    DevTables *d_tabs = nullptr;  // tables of the groups of a sweep
    size_t cap_tabs = 0;
    std::vector<hipEvent_t> gev;  // per-group "forward pass done" events of a sweep
    std::vector<hipStream_t> gstreams; // side streams of the 2nd, 3rd ... group of a sweep (three each), high priority
    std::vector<hipStream_t> gstreams_n; // the same at normal priority (throughput-bound batches, see auxn)
    std::vector<hipEvent_t> gjev;      // their join events
    hipEvent_t jev[6] = {}; // join events of the side streams
    hipEvent_t fev[2] = {nullptr, nullptr};                            // k_finish waits for the forward / window streams
    plaac_params params;
    // plan / scratch buffers (grown on demand)
    // The plan of a call - effective lengths, length histogram, the sorted plan {offset lo, offset hi, effective length,
    // protein index}, row offsets of the wave-groups, the packed copy - exists TWICE, used by alternate calls: with
    // overlapping calls (plaac_ctx_set_overlap) the planning and packing of call k+1 then depend on call k-1 only and run
    // beside the whole of call k.
    struct PlanBufs {
        uint32_t *neff = nullptr, *hist = nullptr, *grow = nullptr;
        uint4 *order = nullptr, *packed = nullptr;
        // [0, nprot) lmarginalprob of hmm1, [nprot, 2 nprot) total of hmm0 (single-point calls whose kernels leave the terms to
        // k_finish: latency / mixed forms, track mode; throughput-bound calls write HMMall / HMMvit from k_fwd / k_vit). Per
        // call too: the forward / window kernels of the next call write theirs while k_finish of this one still reads.
        // [2 nprot, 3 nprot) end score of the Viterbi path (k_vit<.., EXT>; k_finish writes HMMvit).
        double *lat = nullptr;
        size_t cap_prot = 0, cap_order = 0, cap_grow = 0, cap_packed = 0, cap_lat = 0;
        // Everything else the lane-per-protein kernels of a call write besides the rows (round 4: per call parity as well,
        // so that the chain kernels of call k+1 need nothing from call k and the long runs of consecutive calls - k_long on
        // alternating streams - run side by side): traceback / path bit words, core list and its count (k_vit<.., LIST> ->
        // k_core_list), scratch of the position-parallel core search of the long wave-groups (k_core_*).
        uint32_t *bits = nullptr, *corelist = nullptr, *corecount = nullptr, *coreflags = nullptr;
        double *corep = nullptr; // latency forms: masked prefix sums of the long wave-groups, packed row numbering
        void *corepart = nullptr; // their per-row best windows
        size_t cap_bits = 0, cap_corelist = 0, cap_corecount = 0, cap_corep = 0, cap_corepart = 0;
        // summary-mode window kernels in filter form (per call parity since the filter tier of call k+1 runs beside the
        // refine / exact tier of call k): refine list (offset lo, offset hi, length, centre) + row index, and the plan items
        // of the proteins the filter tier hands to the exact tier
        uint4 *clist = nullptr, *fblist = nullptr;
        uint32_t *crow = nullptr;
        size_t cap_clist = 0, cap_crow = 0, cap_fblist = 0;
    } pl[2];
    double2 *d_fwd = nullptr, *d_bwd = nullptr; // track mode: forward / backward pairs, group-interleaved
    sched::Knobs knobs; // the environment switches, read once at creation
    uint32_t *h_pin = nullptr; // pinned words: [0] total packed rows of a call, [1] upload validation flag, [2..5] see score_points
    uint32_t *d_hpin = nullptr; // the same words as the device sees them
    bool poll_ok = true;        // the host polls h_pin[5] for the plan words (false: stream synchronisation; PLAAC_POLL_PLAN=0)
    unsigned long polled = 0, synced = 0; // DIAGNOSTIC (PLAAC_STREAM_DEBUG): calls served either way
    // pinned staging for the host-buffer entry points (pageable memcpy runs at a tenth of the link rate)
    static constexpr size_t STAGE_BYTES = 16u << 20;
    uint8_t *h_stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    // plaac_text_upload (an uploader thread of the host beside the thread that scores): its own stream, pinned buffers and
    // recycled sets of device buffers
    hipStream_t up = nullptr;
    uint8_t *h_stage_up[2] = {nullptr, nullptr};
    hipEvent_t stage_ev_up[2] = {nullptr, nullptr};
    std::mutex up_mu;
    std::string up_err;           // message of the last failing plaac_text_upload (written by the uploader thread only)
    std::vector<plaac_text_batch *> up_pool;
    uint32_t *d_flag = nullptr;
    KbDivTab *d_divtab = nullptr; // reciprocal tables of the window kernel
    uint32_t *d_kbcnt = nullptr;   // four sets of list counters, used in turn (see CallData::counters)
    uint32_t *d_fbcount = nullptr; // = d_kbcnt (context creation uses the word as the reciprocal check's error count)
    size_t cap_ccount = 0;
    static constexpr int KB_MAXSEG = sched::KB_MAXSEG;
    // PLAAC_KB_CHUNKS (1..8): chunks of the lane-form filter, each refined on a second stream while the next is filtered.
    // Measured at 10 M sequences: 1 chunk (filter, refine, exact tier in a row on the caller's stream) 21.0 ms, 4 chunks
    // 21.1 ms, 8 chunks with a second hardware queue feeding the filter 20.5 ms; holding the chain kernels back until
    // the filter is through (so that the refine kernel does not end the step alone) 20.5 - 22.6 against 20.4 ms: the
    // step is bound by the sum of the kernels' issue time, not by their order, and the extra event hops cost the small
    // batches 0.3 ms. Default 1.
    hipEvent_t kbev[KB_MAXSEG] = {};
    static constexpr int TRK_MAXSEG = sched::TRK_MAXSEG; // track mode: runs of wave-groups (k_post of one run beside the chains of the next)
    hipEvent_t tfev[TRK_MAXSEG] = {}, tbev[TRK_MAXSEG] = {}, tpev = nullptr; // forward / backward of a run done; posteriors done
    hipEvent_t pkev[TRK_MAXSEG] = {}; // packed copy of a run done
    hipEvent_t rev[2 * TRK_MAXSEG] = {}; // EK_R
    // Consecutive calls may overlap (plaac_ctx_set_overlap): the planning and packing of a call - HBM-bound, they touch the
    // plan buffers only - run beside the "tail" of the previous call, the exact values at the PAPA centres and the exact
    // tier (k_refine_centres, k_tracks20 over the fallback list: instruction-bound, alone on the chip otherwise), which read
    // their own lists and this call's copy of the "huge" word only. tail_ev: recorded on the caller's stream before the tail.
    // With the plan double-buffered the head of call k+1 depends on call k-1 only (same buffers): on tail_ev2[p] (recorded on
    // the caller's stream before the tail of the call that used buffers p; tail_open2[p]: that call recorded one) and
    // ka_done[p] (its side streams joined). The chain kernels of call k+1 wait for ka_done of call k (shared scratch).
    bool overlap = false, tail_open2[2] = {false, false}, last_chain_bound = false, last_single_summary = false;
    hipEvent_t tail_ev2[2] = {nullptr, nullptr}, ka_done[2] = {nullptr, nullptr};
    uint32_t *d_huge = nullptr; // four words, used in turn: the tail of call k reads its word while call k+2 already plans
    // Chain-bound single-point summary calls in MIXED FORMS (round 4, PLAAC_MIXED=0: latency forms for every wave-group as
    // in round 3): the long wave-groups on two streams of the high class per call parity, every other wave-group in the
    // throughput forms on the normal class's role streams (schedule.hip.inc, decide_forms).
    bool last_mixed = false;
    hipEvent_t lev[2] = {nullptr, nullptr}; // long run of a call done (per parity)
    bool core_par_ok = false; // the tables in d_tab pass core_par_tables_ok
    bool fi_int = false;   // the tables in d_tab qualify for FoldIndex in integers (derive_fi_int)
    size_t cap_fwd = 0, cap_bwd = 0;
    // Pipelined host-buffer scoring (plaac_score_begin / _end): two batches in flight. Each slot owns its device copies of
    // the batch and its rows; uploads and downloads go through `xfer`, a copy stream of their own, so that the upload of
    // batch k+1 and the download of batch k run beside the kernels (which are on `stream` and the side streams).
    struct Slot {
        uint8_t *d_codes = nullptr;
        uint64_t *d_offsets = nullptr;
        plaac_row *d_rows = nullptr;
        unsigned long long *d_counts = nullptr; // plaac_score_begin_counting: the batch's 22 background counts
        bool counted = false;
        // plaac_score_begin_text: the batch's FASTA text, record starts, per-record lengths / blank-line flags (K1)
        char *d_text = nullptr;
        uint64_t *d_starts = nullptr;
        uint32_t *d_len = nullptr;
        uint8_t *d_blank = nullptr;
        FaExtent *d_ext = nullptr;
        unsigned long long *d_total = nullptr, *d_bsum = nullptr;
        size_t cap_text = 0, cap_starts = 0, cap_len = 0, cap_blank = 0, cap_ext = 0, cap_bsum = 0;
        // plaac_score_end_text_table: the batch's rows as text (format_device.hip.inc)
        uint64_t *d_toffs = nullptr;
        char *d_table = nullptr;
        uint32_t *d_fmtflags = nullptr;
        size_t cap_toffs = 0, cap_table = 0, cap_fmtflags = 0;
        uint64_t table_bytes = 0;
        bool table_sized = false;
        bool hist_only = false;        // plaac_histogram_begin_text: parsed and counted, not scored
        hipEvent_t hist_ev = nullptr;  // ... its histogram is through
        uint64_t nres = 0;
        bool from_text = false;
        size_t cap_codes = 0, cap_offs = 0, cap_rows = 0;
        uint32_t nprot = 0;
        uint64_t call_no = 0; // ncalls of the scoring call (its join event is ev[call_no % EV_SETS][E_JOIN]); ~0: nothing enqueued
        bool busy = false;
    } slot[2];
    unsigned slot_next = 0, slot_oldest = 0, slots_busy = 0;
    hipStream_t xfer = nullptr;
    int table_corelength = 60, table_ww2 = 41, table_prev_blank = 1; // (plaac_score_end_text_table_size -> _table)
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
    // Side streams, one per ROLE. The HIP runtime multiplexes the streams of a priority class onto FOUR hardware queues,
    // and kernels of different streams that share a queue run strictly one after the other (tools/queue_probe.hip,
    // profiles/r03_queue_probe.txt). Which roles share a queue therefore decides the step - measured on one box: the
    // forward kernel sharing with the window-track kernels 27.7 ms per 10 M sequences, nobody sharing with them 22.3 ms -
    // and which streams share one depends on everything the process has created and destroyed before (the probe shows
    // 0-7 1-6 2-5-9 3-4-8 for ten fresh streams, 0-3 1-4 after others had been destroyed). So the context MEASURES which of
    // its candidate streams collide with each other and with the caller's stream (assign_role_streams) and gives the
    // roles that run together streams that do not. The window-track kernels (KB) run on the caller's stream.
    enum Role { R_WIN = sched::R_WIN, R_VIT = sched::R_VIT, R_FWD = sched::R_FWD, R_KB = sched::R_KB, R_BWD = sched::R_BWD,
                R_WIN2 = sched::R_WIN2, NROLES = sched::NROLES };
    static constexpr int NCAND = 8;
    hipStream_t cand[2][NCAND] = {}; // [0] normal, [1] high priority: every side stream the context created
    int ncand[2] = {0, 0};
    bool cand_hit[2][NCAND][NCAND] = {}; // measured: the two candidates are serialised against each other
    hipStream_t roles_for = nullptr;     // the caller's stream the normal-class roles were assigned for
    bool roles_assigned = false;
    long long *d_qprobe = nullptr;
    int role_cost[2] = {-1, -1}; // weighted collisions left after the assignment (0: the roles that meet are all apart)
    hipStream_t aux[NROLES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // high priority (chain-bound batches)
    // the same roles at normal priority, for throughput-bound batches: there the step is the sum of all kernels' issue
    // time, and high-priority chain kernels only keep the window kernel's waves out of the SIMDs until they are done
    // (chains, then windows, one after the other); at equal priority the two mix and hide each other's latencies
    hipStream_t auxn[NROLES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::string err;
};

namespace {

thread_local std::string g_create_err;

// Where a failing call leaves its message: the context's slot (plaac_last_error) - or, inside plaac_text_upload, which may run
// on a second host thread beside the scoring calls of the same context, the uploader's own slot (plaac_text_upload_error):
// two threads never write one std::string.
thread_local std::string *t_err_redirect = nullptr;
struct ErrRedirect {
    std::string *prev;
    explicit ErrRedirect(std::string *to) : prev(t_err_redirect) { t_err_redirect = to; }
    ~ErrRedirect() { t_err_redirect = prev; }
};
inline std::string &err_slot(plaac_ctx *ctx);
#define PL_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            err_slot(ctx) = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return e_ == hipErrorOutOfMemory ? PLAAC_ERR_NOMEM : PLAAC_ERR_DEVICE;                     \
        }                                                                                              \
    } while (0)

plaac_status fail(plaac_ctx *ctx, plaac_status st, const char *msg);
inline std::string &err_slot(plaac_ctx *ctx) { return t_err_redirect ? *t_err_redirect : ctx->err; }
plaac_status fail(plaac_ctx *ctx, plaac_status st, const char *msg) {
    if (ctx) err_slot(ctx) = msg;
    return st;
}

void derive_fi_int(const plaac_params &P, DevTables &D, int32_t *info = nullptr);

// May the forward / backward kernels take lse_lut<CLAMPED> (kernels_chains.hip.inc) for these tables? Its argument: the larger
// operand of every log-sum-exp of the recurrences is a sum of transition / initial / final log-probabilities (all <= 0
// here) and at least one emission log-probability (all <= -0.125 here), so it is <= -0.125, and the table's last entry
// (the most a difference in [40, 40.01) can add) is below half the spacing of the doubles there, 2^-57.
// Test hook LSE_CLAMP=0: never.
bool lse_clamp_ok(const plaac_params &P) {
    {
        std::string v; // (test hook LSE_CLAMP, read whenever tables are built: the tests switch it)
        if (sched::knob_value("LSE_CLAMP", v) && !v.empty() && v[0] == '0') return false;
    }
    auto nonpos = [](double v) { return v <= 0.0; }; // (false for NaN)
    for (int i = 0; i < 2; ++i) {
        if (!nonpos(P.hmm1.li[i]) || !nonpos(P.hmm1.lf[i])) return false;
        for (int j = 0; j < 2; ++j)
            if (!nonpos(P.hmm1.lt[i][j])) return false;
        for (int k = 0; k < NAA; ++k)
            if (!(P.hmm1.le[i][k] <= -0.125)) return false;
    }
    const double last = P.loglut[LUTLEN - 1];
    return last >= 0.0 && last < 0x1p-57;
}

// May they take lse_lut<2>, the form without any clamp? Then every log-sum-exp of the forward and backward recurrences
// (plaac.java:3360-3368, :3383-3391) must have operands less than 40 apart - the reference's own in-range test (:1027) - for
// ANY residue sequence. The two states of the chain cannot drift apart further than the tables allow:
//   forward:  a_j(t) = LSE_i(lt[i][j] + a_i(t-1)) + le_j(x_t), and max_i(.) <= LSE_i(.) <= max_i(.) + ln 2, so
//             |a_0 - a_1| <= Df := max(max_i lt[i][0] - min_i lt[i][1], max_i lt[i][1] - min_i lt[i][0]) + ln 2 + E,
//             E := max_x |le_0(x) - le_1(x)| (also at t = 0: |li_0 - li_1| + E); the operands lt[0][j] + a_0, lt[1][j] + a_1
//             of the next step differ by at most Df + max_j |lt[0][j] - lt[1][j]|;
//   backward: b_i(t) = LSE_j((lt[i][j] + b_j(t+1)) + le_j(x_{t+1})): |b_0 - b_1| <= Db := max_j |lt[0][j] - lt[1][j]| + ln 2
//             (also at the end: |lf_0 - lf_1|); the operands differ by at most max_i |lt[i][0] - lt[i][1]| + Db + E.
// One unit of slack covers the table's interpolation error (3e-6 per step, not accumulating: the bound is per step) and
// the roundings. The reference's tables: 18.1 / 18.0. test hook LSE_CLAMP=0 or 1: never.
bool lse_range_ok(const plaac_params &P) {
    {
        std::string v;
        if (sched::knob_value("LSE_CLAMP", v) && !v.empty() && (v[0] == '0' || v[0] == '1')) return false;
    }
    const plaac_hmm &H = P.hmm1;
    double E = 0.0;
    for (int k = 0; k < NAA; ++k) {
        if (!std::isfinite(H.le[0][k]) || !std::isfinite(H.le[1][k])) return false;
        E = std::max(E, std::fabs(H.le[0][k] - H.le[1][k]));
    }
    for (int i = 0; i < 2; ++i) {
        if (!std::isfinite(H.li[i]) || !std::isfinite(H.lf[i])) return false;
        for (int j = 0; j < 2; ++j)
            if (!std::isfinite(H.lt[i][j])) return false;
    }
    const double ln2 = 0.6931471805599453;
    const double col = std::max(std::max(H.lt[0][0], H.lt[1][0]) - std::min(H.lt[0][1], H.lt[1][1]),
                                std::max(H.lt[0][1], H.lt[1][1]) - std::min(H.lt[0][0], H.lt[1][0]));
    const double Df = std::max(col + ln2 + E, std::fabs(H.li[0] - H.li[1]) + E);
    const double rows = std::max(std::fabs(H.lt[0][0] - H.lt[1][0]), std::fabs(H.lt[0][1] - H.lt[1][1])); // same target state
    const double cols = std::max(std::fabs(H.lt[0][0] - H.lt[0][1]), std::fabs(H.lt[1][0] - H.lt[1][1])); // same source state
    const double Of = Df + rows;
    const double Db = std::max(rows + ln2, std::fabs(H.lf[0] - H.lf[1]));
    const double Ob = cols + Db + E;
    return std::max(Of, Ob) + 1.0 < 39.9;
}

void fill_tables(const plaac_params &P, DevTables &D) {
    std::memset(&D, 0, sizeof D);
    for (int k = 0; k < NAA; ++k) {
        D.row[k][R_LE0] = P.hmm1.le[0][k];
        D.row[k][R_LE1] = P.hmm1.le[1][k];
        D.row[k][R_LLR] = P.llr[k];
        D.row[k][R_HYD] = P.hydro2[k];
        D.row[k][R_LE0H] = P.hmm0.le[0][k];
        D.row[k][R_PAD] = win_row_words(k, (int)P.charge[k]); // (two integers: what k_win's packed counters add / take off)
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
    std::memcpy(D.loglut, P.loglut, sizeof P.loglut); // D.loglut[LUTLEN], [LUTLEN + 1] stay 0.0
    D.lse_clamp = lse_range_ok(P) ? 2 : (lse_clamp_ok(P) ? 1 : 0);
    derive_fi_int(P, D);
}

// FoldIndex in integers. The reference's hydropathy table is aahydro/9 + 0.5 with one-decimal aahydro (:90) and
// cc = {2.785, -1, -1.151} (:4885): rationals H_k/SH and c_i/SC with SH = 90, SC = 1000. In exact arithmetic
//   m * fi = cc0 S_h + cc1 |C| + cc2 m = (c0 K + c1 SH |C| + c2 SH m) / (SH SC) =: I / (SH SC),   K = sum of H over the window,
// an integer over a fixed denominator: either I = 0 or |m fi| >= 1/(SH SC), while the reference's fp64 evaluation is
// within E_G (~1e-12) of it. So for I != 0 the sign of the reference's FoldIndex is the sign of I, with certainty and
// without fp64; likewise the doubly smoothed FoldIndex (its weights are the window counts m, :2640-2646) has the sign
// of the 41-term sum of I. I = 0 (either level) is left to the exact kernel as before. This function looks for the
// denominators, checks every range the kernel relies on (bit fields of the packed prefix sum, 24-bit multiplier
// operands, int32 window sums) and the margin 1/(SH SC) >> E_T; tables that do not qualify keep the fp64 form.
void derive_fi_int(const plaac_params &P, DevTables &D, int32_t *info) {
    D.fi_int = 0;
    D.fi_A2 = D.fi_B2 = D.fi_C2 = 0;
    for (int k = 0; k < NAA; ++k) D.fi_iv[k] = 0u;
    constexpr long double TOL = 1e-14L;
    long H[NAA] = {0};
    long double dh = 0.0L, dc = 0.0L;
    int SH = 0, SC = 0;
    for (int s = 1; s <= 20000 && !SH; ++s) {
        bool ok = true;
        long double worst = 0.0L;
        for (int k = 0; k < NAA && ok; ++k) {
            const long double x = (long double)P.hydro2[k] * s, r = nearbyintl(x);
            ok = std::isfinite(P.hydro2[k]) && fabsl(x - r) <= TOL * s;
            H[k] = (long)r;
            worst = std::max(worst, fabsl(x - r) / s);
        }
        if (ok) {
            SH = s;
            dh = worst;
        }
    }
    long cI[3] = {0, 0, 0};
    for (int s = 1; s <= 100000 && !SC; ++s) {
        bool ok = true;
        long double worst = 0.0L;
        for (int i = 0; i < 3 && ok; ++i) {
            const long double x = (long double)P.cc[i] * s, r = nearbyintl(x);
            ok = std::isfinite(P.cc[i]) && fabsl(x - r) <= TOL * s;
            cI[i] = (long)r;
            worst = std::max(worst, fabsl(x - r) / s);
        }
        if (ok) {
            SC = s;
            dc = worst;
        }
    }
    if (!SH || !SC) return;
    long hmin = H[0], hmax = H[0];
    double A_h = 0.0;
    for (int k = 0; k < NAA; ++k) {
        hmin = std::min(hmin, H[k]);
        hmax = std::max(hmax, H[k]);
        A_h = std::max(A_h, std::fabs(P.hydro2[k]));
        if (P.charge[k] != -1.0 && P.charge[k] != 0.0 && P.charge[k] != 1.0) return;
    }
    constexpr long W = 2 * TW + 1;
    const long range = hmax - hmin;
    const long A2 = 2 * cI[0], B2 = 2 * (cI[0] * hmin + cI[2] * (long)SH), C2 = 2 * cI[1] * (long)SH;
    if (W * range >= (1L << 19)) return;                                   // K' field: bits 13..31
    if (std::labs(A2) >= (1L << 23) || std::labs(B2) >= (1L << 23) || std::labs(C2) >= (1L << 23)) return; // v_mad_i32_i24
    const long maxI2 = std::labs(A2) * W * range + std::labs(B2) * W + std::labs(C2) * W + 1;
    if (maxI2 * W >= (1L << 31) - 1024) return;                            // second-level window sums in int32
    const long double u = 0x1p-53L, CC = fabsl((long double)P.cc[0]) * A_h + fabsl((long double)P.cc[1]) + fabsl((long double)P.cc[2]);
    const long double E_S = 256 * u * W * A_h + W * dh;
    const long double E_G = fabsl((long double)P.cc[0]) * E_S + 256 * u * W * CC + W * dc * (A_h + 2.0L);
    const long double E_T = 2 * W * E_G + 256 * u * W * W * CC;
    if (1.0L / ((long double)SH * SC) < 1024.0L * E_T) return;             // margin between I = +-1 and the rounding noise
    if (info) {
        info[3] = SH;
        info[4] = SC;
        info[5] = (int32_t)hmin;
    }
    D.fi_int = 1;
    D.fi_A2 = (int32_t)A2;
    D.fi_B2 = (int32_t)B2;
    D.fi_C2 = (int32_t)C2;
    for (int k = 0; k < NAA; ++k)
        D.fi_iv[k] = 1u | ((uint32_t)((int)P.charge[k] + 1) << 6) | ((uint32_t)(H[k] - hmin) << 13);
}

// k_core_par replaces the serial masked prefix chain by exact sums on the grid of the binade of k * |big_neg| (see the
// kernel). That needs rn_q(llr) to be tie-free for every grid the chain can reach: q = 2^(e-52) for the binades e of
// k * |big_neg|, k = 1 .. 65535; a tie needs the lowest set bit of an llr value to be exactly q/2.
bool core_par_tables_ok(const plaac_params &P) {
    if (!(P.big_neg < 0.0) || !std::isfinite(P.big_neg)) return false;
    int e_lo = 0, e_hi = 0;
    (void)std::frexp(-P.big_neg, &e_lo);           // -big_neg = f * 2^e_lo, f in [0.5, 1): binade exponent e_lo - 1
    (void)std::frexp(-P.big_neg * 65535.0, &e_hi);
    e_lo -= 1;
    e_hi -= 1;
    if (e_lo < 8) return false; // |big_neg| must dwarf the llr values (the reference's is 1e6)
    for (int k = 0; k < NAA; ++k) {
        const double t = P.llr[k];
        if (!std::isfinite(t)) return false;
        if (t == 0.0) continue;
        int e = 0;
        const double f = std::frexp(std::fabs(t), &e); // |t| = f * 2^e
        const unsigned long long m = (unsigned long long)std::ldexp(f, 53); // 53-bit integer mantissa: |t| = m * 2^(e-53)
        const int low = e - 53 + __builtin_ctzll(m);                        // exponent of the lowest set bit
        if (low >= e_lo - 53 && low <= e_hi - 53) return false;             // = q/2 for some reachable q = 2^(eb-52)
    }
    return true;
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

// The window kernels divide by multiplying with a tabulated reciprocal and one correction step (SharedDiv), which is the
// correctly rounded quotient if and only if the reciprocal itself is correctly rounded. Checked at every context
// creation: the table k_build_divtab left on the device against the HOST's IEEE quotients, and the device-side count of
// k_check_recip (the in-kernel constructor over every denominator it can meet, against the device's IEEE division).
constexpr uint32_t RECIP_CHECK_MAX = 1u << 19;
const char *check_divtab(plaac_ctx *ctx) {
    static thread_local std::string msg;
    std::vector<KbDivTab> h(1);
    uint32_t bad = 0;
    if (hipMemcpy(h.data(), ctx->d_divtab, sizeof(KbDivTab), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&bad, ctx->d_fbcount, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemset(ctx->d_fbcount, 0, sizeof(uint32_t)) != hipSuccess)
        return "reciprocal table: device copy failed";
    if (bad) {
        msg = "SharedDiv: " + std::to_string(bad) + " in-kernel reciprocals differ from the IEEE quotient";
        return msg.c_str();
    }
    constexpr int W = TW, M = 2 * W + 1;
    auto off = [&](const double2 &e, double den, const char *what, int i) -> bool {
        volatile double q = 1.0 / den; // IEEE division on the host
        if (e.x == den && e.y == q) return false;
        msg = std::string("reciprocal table entry ") + what + "[" + std::to_string(i) + "] is not the IEEE quotient 1/den";
        return true;
    };
    for (int i = 0; i <= M; ++i)
        if (off(h[0].first[i], (double)(i > 0 ? i : 1), "first", i)) return msg.c_str();
    for (int i = 0; i < M * M; ++i) {
        const int ml = i / M, mr = i % M;
        const int den = M + (M * W - ((ml * (ml + 1)) >> 1)) + (M * W - ((mr * (mr + 1)) >> 1));
        if (off(h[0].second[i], (double)den, "second", i)) return msg.c_str();
    }
    if (!(h[0].second[M * M].x == 1.0 && std::isnan(h[0].second[M * M].y)) ||
        !(h[0].second[M * M + 1].x == 1.0 && h[0].second[M * M + 1].y == 1.0))
        return "reciprocal table: sentinel entries are off";
    return nullptr;
}

// Give every role a stream (see plaac_ctx::Role). Per priority class: start with as many candidate streams as the class
// has roles, measure which of them are serialised against each other - and, in the normal class, against the caller's
// stream `st`, which carries the window-track kernels - (a 100 us spin on one, a time stamp on the other; a collision is
// confirmed by a second run, because a late submission of the second kernel looks like one), take the assignment with the
// least weighted collision cost over the roles that run together, and add candidates (up to NCAND) while that cost is
// above the least possible one. As few streams as possible: the HIP runtime's own per-call cost grows with the number of live streams
// (measured: 17 streams per context 4.97 ms per 1.25 M-sequence step, 13 streams 4.06 ms). The high class is assigned
// at context creation (st == nullptr), the normal class at the first scoring call on a given caller stream.
// PLAAC_STREAM_PROBE=0: candidates in creation order, unmeasured.
const char *assign_role_streams(plaac_ctx *ctx, int cls, hipStream_t st) {
    constexpr int NMAX = plaac_ctx::NCAND;
    const bool probe = ctx->knobs.stream_probe;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    auto collide = [&](hipStream_t a, hipStream_t b, bool &out) -> bool { // false: HIP failure
        out = true;
        for (int rep = 0; rep < 2 && out; ++rep) {
            hipLaunchKernelGGL(k_qprobe_spin, dim3(1), dim3(64), 0, a, 10000ll /* 100 us */, ctx->d_qprobe);
            hipLaunchKernelGGL(k_qprobe_mark, dim3(1), dim3(64), 0, b, ctx->d_qprobe + 1);
            long long t[2] = {0, 0};
            if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess ||
                hipMemcpy(t, ctx->d_qprobe, sizeof t, hipMemcpyDeviceToHost) != hipSuccess)
                return false;
            out = t[1] >= t[0];
        }
        return true;
    };
    // roles of a class that get a stream of their own, and the cost of two of them sharing a queue (0: they never run
    // together). Normal class: KB is the caller's stream. Measured per 10 M sequences (profiles/r03_role_sharing.txt): all
    // apart 22.3 - 23.5 ms, the forward kernel with the window-track kernels 27.7, Viterbi with them 30.7, ...
    static const int roles[2][4] = {{plaac_ctx::R_WIN, plaac_ctx::R_VIT, plaac_ctx::R_FWD, plaac_ctx::R_BWD},
                                    {plaac_ctx::R_WIN, plaac_ctx::R_VIT, plaac_ctx::R_FWD, plaac_ctx::R_WIN2}};
    static const int W[2][4][4] = {{{0, 4, 4, 3}, {4, 0, 8, 6}, {4, 8, 0, 8}, {3, 6, 8, 0}},   // normal: WIN VIT FWD BWD
                                   {{0, 8, 8, 8}, {8, 0, 9, 8}, {8, 9, 0, 8}, {8, 8, 8, 0}}};  // high: WIN VIT FWD WIN2
    static const int Wst[4] = {6, 9, 9, 9}; // normal class: sharing a queue with the window-track kernels
    bool(&hit)[NMAX][NMAX] = ctx->cand_hit[cls];
    bool hit_st[NMAX] = {};
    int &n = ctx->ncand[cls];
    if (probe && st)
        for (int c = 0; c < n; ++c)
            if (!collide(st, ctx->cand[cls][c], hit_st[c])) return "stream probe failed";
    int best_cost = 1 << 30, best[4] = {0, 1, 2, 3};
    for (;;) {
        if (n >= 4) {
            int cur[4];
            best_cost = 1 << 30;
            auto search = [&](auto &&self, int r, unsigned used, int cost) -> void {
                if (r == 4) {
                    if (cost < best_cost) {
                        best_cost = cost;
                        for (int i = 0; i < 4; ++i) best[i] = cur[i];
                    }
                    return;
                }
                for (int c = 0; c < n; ++c) {
                    if (used & (1u << c)) continue;
                    int add = (cls == 0 && hit_st[c]) ? Wst[r] : 0;
                    for (int q = 0; q < r; ++q)
                        if (hit[cur[q]][c]) add += W[cls][q][r];
                    cur[r] = c;
                    self(self, r + 1, used | (1u << c), cost + add);
                }
            };
            search(search, 0, 0u, 0);
            if (!probe || best_cost <= (cls == 0 && st ? 3 : 0) || n >= NMAX) break; // (normal: five parties on four queues)
        }
        // one more candidate, probed against the earlier ones (and the caller's stream)
        if (hipStreamCreateWithPriority(&ctx->cand[cls][n], hipStreamNonBlocking, cls ? greatest : 0) != hipSuccess)
            return "hipStreamCreateWithPriority failed";
        for (int a = 0; probe && a < n; ++a) {
            bool h = false;
            if (!collide(ctx->cand[cls][a], ctx->cand[cls][n], h)) return "stream probe failed";
            hit[a][n] = hit[n][a] = h;
        }
        if (probe && st && !collide(st, ctx->cand[cls][n], hit_st[n])) return "stream probe failed";
        ++n;
    }
    hipStream_t *dst = cls ? ctx->aux : ctx->auxn;
    for (int i = 0; i < 4; ++i) dst[roles[cls][i]] = ctx->cand[cls][best[i]];
    // roles without a stream of their own: BWD (track mode) and WIN2 (latency forms, summary mode) never meet, so in the
    // high class BWD takes WIN2's stream; in the normal class WIN2 only carries a join event
    if (cls) ctx->aux[plaac_ctx::R_BWD] = ctx->aux[plaac_ctx::R_WIN2];
    else ctx->auxn[plaac_ctx::R_WIN2] = ctx->auxn[plaac_ctx::R_WIN];
    ctx->role_cost[cls] = probe ? best_cost : -1;
    if (ctx->knobs.stream_debug) { // DIAGNOSTIC: the measured collisions and the assignment, on stderr
        std::fprintf(stderr, "plaac: %s-priority streams, %d candidates, collisions:", cls ? "high" : "normal", n);
        for (int a = 0; a < n; ++a)
            for (int b = a + 1; b < n; ++b)
                if (hit[a][b]) std::fprintf(stderr, " %d-%d", a, b);
        for (int a = 0; a < n; ++a)
            if (hit_st[a]) std::fprintf(stderr, " caller-%d", a);
        std::fprintf(stderr, "; roles %s ->", cls ? "WIN VIT FWD WIN2" : "WIN VIT FWD BWD");
        for (int i = 0; i < 4; ++i) std::fprintf(stderr, " %d", best[i]);
        std::fprintf(stderr, "; cost %d\n", best_cost);
    }
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
const char *plaac_text_upload_error(const plaac_ctx *ctx) { return ctx ? ctx->up_err.c_str() : ""; }

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
    const bool ctx_timing = std::getenv("PLAAC_CTX_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { // PLAAC_CTX_TIMING=1: what bringing a context up is made of (stderr)
        if (!ctx_timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "plaac-ctx-timing: %-40s %8.3f ms\n", what, std::chrono::duration<double>(t - t_last).count() * 1e3);
        t_last = t;
    };
    hipError_t e = hipGetDeviceCount(&ndev);
    lap("hipGetDeviceCount (HIP comes up)");
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
    ctx->knobs = sched::read_knobs(); // every environment switch, once
    auto bail = [&](const char *what, hipError_t err) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(err);
        plaac_ctx_destroy(ctx);
        return PLAAC_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->lat_unit = lat_unit_usable();
    lap("hipSetDevice + properties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        plaac_ctx_destroy(ctx);
        return PLAAC_ERR_DEVICE;
    }
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e);
        if ((e = hipMalloc((void **)&ctx->d_qprobe, 2 * sizeof(long long))) != hipSuccess) return bail("hipMalloc(qprobe)", e);
        lap("first stream + first hipMalloc");
        if (const char *why = assign_role_streams(ctx, 1, nullptr)) {
            g_create_err = why;
            plaac_ctx_destroy(ctx);
            return PLAAC_ERR_DEVICE;
        }
        lap("high-priority role streams (create + probe)");
        for (auto &je : ctx->jev)
            if ((e = hipEventCreateWithFlags(&je, hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e);
        for (auto &fe : ctx->fev)
            if ((e = hipEventCreateWithFlags(&fe, hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate", e);
        for (auto *arr : {ctx->tfev, ctx->tbev, ctx->pkev})
            for (int k = 0; k < plaac_ctx::TRK_MAXSEG; ++k)
                if ((e = hipEventCreateWithFlags(&arr[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&ctx->tpev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        for (auto &re : ctx->rev)
            if ((e = hipEventCreateWithFlags(&re, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        for (auto *arr : {ctx->tail_ev2, ctx->ka_done, ctx->lev})
            for (int k = 0; k < 2; ++k)
                if ((e = hipEventCreateWithFlags(&arr[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipMalloc((void **)&ctx->d_huge, 4 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(huge)", e);
        ctx->overlap = ctx->knobs.overlap_default;
        for (auto &ke : ctx->kbev)
            if ((e = hipEventCreateWithFlags(&ke, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    for (auto &set : ctx->ev)
        for (auto &ev : set)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    lap("events");
    if ((e = hipMalloc((void **)&ctx->d_tab, sizeof(DevTables))) != hipSuccess) return bail("hipMalloc(tables)", e);
    for (auto &pb : ctx->pl)
        if ((e = hipMalloc((void **)&pb.hist, sizeof(uint32_t) * (LEN_BINS + 1))) != hipSuccess)
            return bail("hipMalloc(hist)", e);
    if ((e = hipMalloc((void **)&ctx->d_counts, sizeof(unsigned long long) * NAA)) != hipSuccess)
        return bail("hipMalloc(counts)", e);
    if ((e = hipHostMalloc((void **)&ctx->h_pin, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
        return bail("hipHostMalloc", e);
    std::memset(ctx->h_pin, 0, 64);
    if ((e = hipHostGetDevicePointer((void **)&ctx->d_hpin, ctx->h_pin, 0)) != hipSuccess)
        return bail("hipHostGetDevicePointer", e);
    ctx->poll_ok = ctx->knobs.poll_plan;
    if ((e = hipMalloc((void **)&ctx->d_divtab, sizeof(KbDivTab))) != hipSuccess) return bail("hipMalloc(divtab)", e);
    if ((e = hipMalloc((void **)&ctx->d_kbcnt, sizeof(uint32_t) * 4 * KB_COUNTER_WORDS)) != hipSuccess) return bail("hipMalloc(counters)", e);
    if ((e = hipMemset(ctx->d_kbcnt, 0, sizeof(uint32_t) * 4 * KB_COUNTER_WORDS)) != hipSuccess) return bail("hipMemset(counters)", e);
    ctx->d_fbcount = ctx->d_kbcnt;
    hipLaunchKernelGGL(k_build_divtab, dim3(((2 * TW + 1) * (2 * TW + 1) + 256) / 256), dim3(256), 0, ctx->stream,
                       ctx->d_divtab);
    hipLaunchKernelGGL(k_check_recip, dim3((RECIP_CHECK_MAX + 255u) / 256u), dim3(256), 0, ctx->stream, RECIP_CHECK_MAX,
                       ctx->d_fbcount);
    lap("small allocations");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail("k_build_divtab", e);
    lap("first kernels (code object load) + sync");
    if (const char *why = check_divtab(ctx)) { // SharedDiv is correctly rounded only if every reciprocal is
        g_create_err = why;
        plaac_ctx_destroy(ctx);
        return PLAAC_ERR_DEVICE;
    }
    plaac_status st = plaac_ctx_set_params(ctx, params);
    lap("tables");
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
    const bool fi_int = h->fi_int != 0;
    const bool cpar = core_par_tables_ok(*params);
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
    ctx->fi_int = fi_int;
    ctx->core_par_ok = cpar;
    return PLAAC_OK;
}

void plaac_ctx_destroy(plaac_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->knobs.stream_debug)
        std::fprintf(stderr, "plaac: plan words polled in %lu calls, stream-synchronised in %lu\n", ctx->polled, ctx->synced);
    if (ctx->d_bwd) (void)hipFree(ctx->d_bwd);
    void *bufs[] = {ctx->d_tab,        ctx->d_fwd,         ctx->d_codes,      ctx->d_offsets,
                    ctx->d_rows,       ctx->d_trk8,        ctx->d_trk64,       ctx->d_counts,     ctx->pl[0].neff,
                    ctx->pl[0].order,  ctx->pl[0].hist,    ctx->pl[0].grow,    ctx->pl[0].packed, ctx->pl[1].neff,
                    ctx->pl[1].order,  ctx->pl[1].hist,    ctx->pl[1].grow,    ctx->pl[1].packed};
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (int i = 0; i < 2; ++i) {
        if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    }
    if (ctx->d_flag) (void)hipFree(ctx->d_flag);
    for (auto &sl : ctx->slot)
        if (sl.hist_ev) (void)hipEventDestroy(sl.hist_ev);
    for (int i = 0; i < 2; ++i) {
        if (ctx->h_stage_up[i]) (void)hipHostFree(ctx->h_stage_up[i]);
        if (ctx->stage_ev_up[i]) (void)hipEventDestroy(ctx->stage_ev_up[i]);
    }
    for (plaac_text_batch *tb : ctx->up_pool) {
        for (void *b : {(void *)tb->d_text, (void *)tb->d_starts, (void *)tb->d_len, (void *)tb->d_blank, (void *)tb->d_ext, (void *)tb->d_total,
                        (void *)tb->d_bsum, (void *)tb->d_codes, (void *)tb->d_offsets})
            if (b) (void)hipFree(b);
        delete tb;
    }
    if (ctx->up) {
        (void)hipStreamSynchronize(ctx->up);
        (void)hipStreamDestroy(ctx->up);
    }
    for (auto &sl : ctx->slot)
        for (void *b : {(void *)sl.d_codes, (void *)sl.d_offsets, (void *)sl.d_rows, (void *)sl.d_counts, (void *)sl.d_text,
                        (void *)sl.d_starts, (void *)sl.d_len, (void *)sl.d_blank, (void *)sl.d_total, (void *)sl.d_ext,
                        (void *)sl.d_bsum, (void *)sl.d_toffs, (void *)sl.d_table, (void *)sl.d_fmtflags})
            if (b) (void)hipFree(b);
    if (ctx->xfer) {
        (void)hipStreamSynchronize(ctx->xfer);
        (void)hipStreamDestroy(ctx->xfer);
    }
    if (ctx->d_divtab) (void)hipFree(ctx->d_divtab);
    for (void *b : {(void *)ctx->d_kbcnt, (void *)ctx->pl[0].lat, (void *)ctx->pl[1].lat, (void *)ctx->pl[0].clist,
                    (void *)ctx->pl[1].clist, (void *)ctx->pl[0].crow, (void *)ctx->pl[1].crow, (void *)ctx->pl[0].fblist,
                    (void *)ctx->pl[1].fblist})
        if (b) (void)hipFree(b);
    for (auto &pb : ctx->pl)
        for (void *b : {(void *)pb.bits, (void *)pb.corelist, (void *)pb.corecount, (void *)pb.coreflags, (void *)pb.corep,
                        pb.corepart})
            if (b) (void)hipFree(b);
    for (hipEvent_t e : ctx->lev)
        if (e) (void)hipEventDestroy(e);
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (auto &set : ctx->ev)
        for (auto &ev : set)
            if (ev) (void)hipEventDestroy(ev);
    for (auto &cls : ctx->cand)
        for (auto &a : cls)
            if (a) {
                (void)hipStreamSynchronize(a);
                (void)hipStreamDestroy(a);
            }
    if (ctx->d_qprobe) (void)hipFree(ctx->d_qprobe);
    for (hipEvent_t e : ctx->gev)
        if (e) (void)hipEventDestroy(e);
    for (auto *gs : {&ctx->gstreams, &ctx->gstreams_n})
        for (hipStream_t a : *gs)
            if (a) {
                (void)hipStreamSynchronize(a);
                (void)hipStreamDestroy(a);
            }
    for (hipEvent_t e : ctx->gjev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->jev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tfev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tbev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->pkev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->rev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->tpev) (void)hipEventDestroy(ctx->tpev);
    for (auto *arr : {ctx->tail_ev2, ctx->ka_done})
        for (int k = 0; k < 2; ++k)
            if (arr[k]) (void)hipEventDestroy(arr[k]);
    if (ctx->d_huge) (void)hipFree(ctx->d_huge);
    for (hipEvent_t e : ctx->kbev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->fev)
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

// ------------------------------------------------------------------------------------------------
// A scoring call = the schedule of schedule.hip.inc + this launcher.
// ------------------------------------------------------------------------------------------------
namespace {

// the device side of one call: pointers and sizes the launcher hands to the kernels
struct CallData {
    const uint8_t *d_codes;
    const uint64_t *d_offsets;
    uint32_t nprot, ngroups;
    uint64_t total;
    const plaac_params *points;
    plaac_row *const *d_rows;
    TrackPtrs tp;
    bool tracks;
    const DevTables *gtab0;
    plaac_ctx::PlanBufs *PL;
    size_t bits_stride = 0, total_rows = 0;
    hipStream_t st;
    hipEvent_t *evs;
    uint32_t seq;
    const uint32_t *huge;
    // this call's set of list counters (k_plan_scan resets it): [0] fallback list, [1 .. KB_MAXSEG] centre-list segments,
    // [16 ..] core lists of the sweep groups
    uint32_t *counters, *fbcount, *ccount, *corecount;
    size_t core_lrows = 0;
};

hipStream_t slot_stream(plaac_ctx *ctx, const CallData &D, uint8_t slot) {
    using namespace sched;
    if (slot == S_CALLER) return D.st;
    if (slot < S_NO) return ctx->aux[slot - S_HI];
    if (slot < S_G) return ctx->auxn[slot - S_NO];
    if (slot < S_GN) return ctx->gstreams[slot - S_G];
    return ctx->gstreams_n[slot - S_GN];
}

hipEvent_t slot_event(plaac_ctx *ctx, const CallData &D, uint16_t e) {
    using namespace sched;
    const unsigned idx = e & 255u, par = (unsigned)(ctx->ncalls & 1u);
    switch (e >> 8) {
    case EK_T: return D.evs[idx];
    case EK_PK: return ctx->pkev[idx];
    case EK_J: return ctx->jev[idx];
    case EK_F: return ctx->fev[idx];
    case EK_L: return ctx->lev[par];
    case EK_KA: return ctx->ka_done[idx];
    case EK_TAIL: return ctx->tail_ev2[idx];
    case EK_R: return ctx->rev[idx];
    case EK_KB: return ctx->kbev[idx];
    case EK_TF: return ctx->tfev[idx];
    case EK_TB: return ctx->tbev[idx];
    case EK_TP: return ctx->tpev;
    case EK_G: return ctx->gev[idx];
    case EK_GJ: return ctx->gjev[idx];
    default: return ctx->ev[(ctx->ncalls - idx) % plaac_ctx::EV_SETS][E_JOIN_IDX]; // EK_PREVJOIN
    }
}

// Op -> HIP. One switch: the kernels' arguments are formed here and nowhere else.
plaac_status run_ops(plaac_ctx *ctx, const CallData &D, const sched::CallKind &C, const sched::Forms &F, const sched::PlanWords &W,
                     const std::deque<sched::Op> &ops, size_t from) {
    using namespace sched;
    const Knobs &K = ctx->knobs;
    plaac_ctx::PlanBufs &PL = *D.PL;
    const uint32_t nprot = D.nprot, ngroups = D.ngroups;
    auto seg_first = [&](const std::vector<uint32_t> &sb, size_t k) { return sb[k] * 64u; };
    auto seg_count = [&](const std::vector<uint32_t> &sb, size_t k) {
        return (uint32_t)(std::min<uint64_t>((uint64_t)sb[k + 1] * 64u, nprot) - sb[k] * 64u);
    };
    double *const lmarg = PL.lat, *const h0 = PL.lat ? PL.lat + nprot : nullptr, *const vend = PL.lat ? PL.lat + 2 * (size_t)nprot : nullptr;
    for (size_t i = from; i < ops.size(); ++i) {
        const Op &o = ops[i];
        const hipStream_t s = slot_stream(ctx, D, o.stream);
        if (o.kind == Op::RECORD) {
            PL_HIP(ctx, hipEventRecord(slot_event(ctx, D, o.event), s));
            if ((o.event >> 8) == EK_TAIL) ctx->tail_open2[o.event & 255u] = true;
            continue;
        }
        if (o.kind == Op::WAIT) {
            PL_HIP(ctx, hipStreamWaitEvent(s, slot_event(ctx, D, o.event), 0));
            continue;
        }
        if (K.skip_mask && ((K.skip_mask >> o.kern) & 1ull) && ctx->ncalls >= K.skip_from) continue; // (diagnostic)
        const GroupKind *G = o.group < C.groups.size() ? &C.groups[o.group] : nullptr;
        const DevTables *tab = D.gtab0 + o.group;
        plaac_row *rows0 = G ? D.d_rows[G->first_point] : D.d_rows[0];
        uint32_t *gbits = PL.bits ? PL.bits + D.bits_stride * o.group : nullptr;
        // the core lengths / row arrays of this launch (members m0 .. m0 + nc - 1 of the group; unused slots repeat the last)
        SweepTargets tg{};
        if (G) {
            for (int k = 0; k < MAXC; ++k) {
                const uint32_t idx = G->first_point + o.m0 + (uint32_t)std::min<int>(k, (int)o.nc - 1);
                tg.c[k] = (uint32_t)D.points[idx].corelength;
                tg.rows[k] = D.d_rows[idx];
            }
        }
        tg.stop_after = K.vit_stop;
        switch ((Kern)o.kern) {
        case K_MEMSET_HIST: PL_HIP(ctx, hipMemsetAsync(PL.hist, 0, sizeof(uint32_t) * (LEN_BINS + 1), s)); break;
        case K_MEMSET_COREFLAGS:
            if (!PL.coreflags) PL_HIP(ctx, hipMalloc((void **)&PL.coreflags, sizeof(uint32_t) * CORE_MAX_GROUPS * 64u));
            PL_HIP(ctx, hipMemsetAsync(PL.coreflags, 0, sizeof(uint32_t) * (size_t)W.long_groups * 64u, s));
            break;
        case K_PLAN_LENGTHS: {
            const unsigned plb = (nprot + PLAN_THREADS * PLAN_ITEMS - 1) / (PLAN_THREADS * PLAN_ITEMS);
            hipLaunchKernelGGL(k_plan_lengths, dim3(plb), dim3(PLAN_THREADS), 0, s, D.d_codes, D.d_offsets, nprot, PL.neff, PL.hist);
            break;
        }
        case K_PLAN_SCAN:
            hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(256), 0, s, PL.hist, ctx->d_huge + (ctx->ncalls & 3u), D.counters);
            break;
        case K_PLAN_SCATTER: {
            const unsigned plb = (nprot + PLAN_THREADS * PLAN_ITEMS - 1) / (PLAN_THREADS * PLAN_ITEMS);
            hipLaunchKernelGGL(k_plan_scatter, dim3(plb), dim3(PLAN_THREADS), 0, s, PL.neff, nprot, PL.hist, D.d_offsets, PL.order);
            break;
        }
        case K_GROUP_ROWS:
            hipLaunchKernelGGL(k_group_rows, dim3((ngroups + 255u) / 256u), dim3(256), 0, s, PL.neff, PL.order, nprot, ngroups, PL.grow);
            break;
        case K_SCAN_U32:
            // the scan writes the words the host needs (h_pin[0] total rows, [2] rows of the first wave-group, [3] the long
            // wave-groups, [4] their rows, [6..12] run boundaries) into pinned host memory and then this call's sequence
            // number into h_pin[5]: the host polls that word
            for (int k = 6; k < 6 + SCAN_SEGS - 1; ++k) ctx->h_pin[k] = 0xffffffffu; // (no rows: the scan leaves them alone)
            hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(SCAN_THREADS), 0, s, PL.grow, ngroups, CORE_LONG_ROWS, CORE_MAX_GROUPS,
                               ctx->d_hpin, D.seq);
            break;
        case K_PACK: {
            const uint32_t first = seg_first(F.segb, o.run), cnt = seg_count(F.segb, o.run);
            hipLaunchKernelGGL(k_pack, dim3((cnt + 15u) / 16u), dim3(256), 0, s, D.d_codes, D.d_offsets, PL.neff, PL.order + first,
                               cnt, D.total, PL.grow + F.segb[o.run], PL.packed);
            break;
        }
        case K_LONG: {
            const uint32_t lcnt = (uint32_t)std::min<uint64_t>((uint64_t)F.gl * 64u, nprot);
            if (ctx->lat_unit)
                plaac_lat::launch_long(dim3(4u * ((lcnt + KA_THREADS - 1) / KA_THREADS)), s, D.d_offsets, PL.neff, PL.order, lcnt, tab,
                                       PL.packed, PL.grow, &tg, lmarg, h0);
            else
                hipLaunchKernelGGL(k_long, dim3(4u * ((lcnt + KA_THREADS - 1) / KA_THREADS)), dim3(KA_THREADS), 0, s, D.d_offsets,
                                   PL.neff, PL.order, lcnt, tab, PL.packed, PL.grow, tg, lmarg, h0);
            break;
        }
        case K_VIT: {
            uint32_t first, cnt, g0;
            if (o.sel == 1) { // the long run of a call in mixed forms
                first = 0u, cnt = (uint32_t)std::min<uint64_t>((uint64_t)F.gl * 64u, nprot), g0 = 0u;
                tg.long_groups_elsewhere = F.core_long ? 1u : 0u;
            } else {
                first = seg_first(F.vsegb, o.run), cnt = seg_count(F.vsegb, o.run), g0 = F.vsegb[o.run];
                tg.long_groups_elsewhere = (F.core_long && !F.mixed) ? 1u : 0u; // (mixed forms: the long wave-groups are the long run's)
            }
            tg.first = first;
            const unsigned abk = (cnt + KA_THREADS - 1) / KA_THREADS;
            uint32_t *cl_list = C.single() ? PL.corelist : (PL.corelist ? PL.corelist + (size_t)nprot * o.group : nullptr);
            uint32_t *cl_count = D.corecount + o.group;
            if (!o.list) cl_list = cl_count = nullptr;
#define VIT_ARGS D.d_codes, D.d_offsets, PL.neff, PL.order + first, cnt, tab, PL.packed, PL.grow + g0, gbits, tg, cl_list, cl_count, vend
#define LAUNCH_VIT(NC, LAT, EXT, LIST) hipLaunchKernelGGL((k_vit<NC, LAT, EXT, LIST>), dim3(abk), dim3(KA_THREADS), 0, s, VIT_ARGS)
#define LAUNCH_VIT_NC(LAT, EXT, LIST)                                                                              \
    switch (o.nc) {                                                                                                \
    case 1: LAUNCH_VIT(1, LAT, EXT, LIST); break;                                                                  \
    case 2: LAUNCH_VIT(2, LAT, EXT, LIST); break;                                                                  \
    case 3: LAUNCH_VIT(3, LAT, EXT, LIST); break;                                                                  \
    default: LAUNCH_VIT(4, LAT, EXT, LIST); break;                                                                 \
    }
            if (o.lat && o.ext && ctx->lat_unit && !C.tracks)
                plaac_lat::launch_vit_lat_ext(dim3(abk), s, D.d_codes, D.d_offsets, PL.neff, PL.order + first, cnt, tab, PL.packed,
                                              PL.grow + g0, gbits, &tg, vend);
            else if (o.lat && o.ext) LAUNCH_VIT(1, true, true, false);
            else if (o.lat) LAUNCH_VIT_NC(true, false, false)
            else if (o.ext && o.list) LAUNCH_VIT(1, false, true, true);
            else if (o.list) LAUNCH_VIT_NC(false, false, true)
            else if (o.ext) LAUNCH_VIT(1, false, true, false);
            else LAUNCH_VIT_NC(false, false, false)
#undef LAUNCH_VIT_NC
#undef LAUNCH_VIT
#undef VIT_ARGS
            break;
        }
        case K_CORE_PAR:
        case K_CORE_CHAIN:
        case K_CORE_EVAL:
        case K_CORE_REDUCE: {
            const unsigned lg = W.long_groups; // (the kernels re-check every group: lengths >= 65535 are not ordered)
            double *corep = PL.corep + D.core_lrows * 1024u * o.a;
            CorePart *corepart = (CorePart *)PL.corepart + D.core_lrows * 64u * o.a;
            if (o.kern == K_CORE_PAR)
                hipLaunchKernelGGL(k_core_par, dim3(lg * 64u), dim3(64 * CP_WAVES), 0, s, D.d_codes, PL.order, nprot, tab, PL.grow,
                                   gbits, corep, PL.coreflags);
            else if (o.kern == K_CORE_CHAIN)
                hipLaunchKernelGGL(k_core_chain, dim3(lg), dim3(64), 0, s, PL.order, nprot, tab, PL.packed, PL.grow, gbits, corep,
                                   o.sel ? PL.coreflags : (const uint32_t *)nullptr);
            else if (o.kern == K_CORE_EVAL)
                hipLaunchKernelGGL(k_core_eval, dim3(W.long_rows), dim3(64), 0, s, PL.order, nprot, ngroups, PL.grow, corep, corepart,
                                   tg.c[0]);
            else
                hipLaunchKernelGGL(k_core_reduce, dim3(lg * 64u), dim3(64), 0, s, D.d_codes, PL.order, nprot, tab, PL.grow, gbits,
                                   (const CorePart *)corepart, tg.rows[0], tg.c[0]);
            break;
        }
        case K_CORE_LIST: {
            const unsigned lgrid = std::min((nprot + KA_THREADS - 1) / KA_THREADS, 2048u);
            uint32_t *cl_list = C.single() ? PL.corelist : PL.corelist + (size_t)nprot * o.group;
            uint32_t *cl_count = D.corecount + o.group;
            tg.long_groups_elsewhere = (F.core_long && !F.mixed) ? 1u : 0u;
            tg.first = 0u;
#define LAUNCH_CORE_LIST(NC)                                                                                       \
    hipLaunchKernelGGL(k_core_list<NC>, dim3(lgrid), dim3(KA_THREADS), 0, s, D.d_codes, D.total, PL.order, tab, PL.packed,  \
                       PL.grow, gbits, tg, cl_list, cl_count)
            switch (o.nc) {
            case 1: LAUNCH_CORE_LIST(1); break;
            case 2: LAUNCH_CORE_LIST(2); break;
            case 3: LAUNCH_CORE_LIST(3); break;
            default: LAUNCH_CORE_LIST(4); break;
            }
#undef LAUNCH_CORE_LIST
            break;
        }
        case K_FWD: {
            const uint32_t first = seg_first(F.segb, o.run), cnt = seg_count(F.segb, o.run);
#define LAUNCH_FWD(TRK, EXTF)                                                                                      \
    hipLaunchKernelGGL((k_fwd<TRK, EXTF>), dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_codes, \
                       D.d_offsets, PL.neff, PL.order + first, cnt, tab, PL.packed, PL.grow + F.segb[o.run], rows0, \
                       TRK ? ctx->d_fwd : (double2 *)nullptr, PL.lat, nprot)
#ifdef PLAAC_DIAG
            if (K.fwd_direct && !o.trk && !o.ext)
                hipLaunchKernelGGL(k_fwd_direct, dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_codes, D.total,
                                   D.d_offsets, PL.neff, PL.order + first, cnt, tab, rows0);
            else
#endif
            if (o.trk && o.ext) LAUNCH_FWD(true, true);
            else if (o.trk) LAUNCH_FWD(true, false);
            else if (o.ext) LAUNCH_FWD(false, true);
            else LAUNCH_FWD(false, false);
#undef LAUNCH_FWD
            break;
        }
        case K_FWD_PAIR: { // (o.b > o.a: the wave-groups [o.a, o.b) of the run instead of all of it)
            const bool part = o.b > o.a;
            const uint32_t g0 = part ? o.a : F.segb[o.run];
            const uint32_t first = part ? o.a * 64u : seg_first(F.segb, o.run);
            const uint32_t cnt = part ? (uint32_t)(std::min<uint64_t>((uint64_t)o.b * 64u, nprot) - first) : seg_count(F.segb, o.run);
            const dim3 grid((cnt + KA_THREADS / 2 - 1) / (KA_THREADS / 2));
            if (ctx->lat_unit && !o.trk)
                plaac_lat::launch_fwd_pair(grid, s, PL.order + first, cnt, tab, PL.packed, PL.grow + g0, PL.lat);
            else if (o.trk)
                hipLaunchKernelGGL(k_fwd_pair<true>, grid, dim3(KA_THREADS), 0, s, PL.order + first, cnt, tab, PL.packed,
                                   PL.grow + g0, PL.lat, ctx->d_fwd);
            else
                hipLaunchKernelGGL(k_fwd_pair<false>, grid, dim3(KA_THREADS), 0, s, PL.order + first, cnt, tab, PL.packed,
                                   PL.grow + g0, PL.lat, (double2 *)nullptr);
            break;
        }
        case K_BWD:
        case K_BWD_PAIR:
        case K_BWD_FWD_POST:
        case K_FWD_POST_LANE: { // (o.b > o.a: the wave-groups [o.a, o.b) of the run instead of all of it)
            const bool part = o.b > o.a;
            const uint32_t g0 = part ? o.a : F.segb[o.run];
            const uint32_t first = part ? o.a * 64u : seg_first(F.segb, o.run);
            const uint32_t cnt = part ? (uint32_t)(std::min<uint64_t>((uint64_t)o.b * 64u, nprot) - first) : seg_count(F.segb, o.run);
            if (o.kern == K_BWD && o.sel) // (checkpoints only: k_fwd_post<true> recomputes the pairs between them)
                hipLaunchKernelGGL(k_bwd<true>, dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, D.gtab0, PL.packed, PL.grow + g0, ctx->d_bwd);
            else if (o.kern == K_BWD)
                hipLaunchKernelGGL(k_bwd<false>, dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, D.gtab0, PL.packed, PL.grow + g0, ctx->d_bwd);
            else if (o.kern == K_BWD_PAIR)
                hipLaunchKernelGGL(k_bwd_pair, dim3((cnt + KA_THREADS / 2 - 1) / (KA_THREADS / 2)), dim3(KA_THREADS), 0, s,
                                   PL.order + first, cnt, D.gtab0, PL.packed, PL.grow + g0, ctx->d_bwd);
#ifdef PLAAC_DIAG // (forms measured slower, EXPERIMENTS.md: the one-pass kernel, the lane-store forms of the forward pass)
            else if (o.kern == K_BWD_FWD_POST)
                hipLaunchKernelGGL(k_bwd_fwd_post<true>, dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, tab, PL.packed, PL.grow + g0, PL.lat, h0, ctx->d_bwd, D.tp);
            else if (o.sel && K.track_post_form == 0 && K.track_post_occ3)
                hipLaunchKernelGGL((k_fwd_post<true, 3>), dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, tab, PL.packed, PL.grow + g0, PL.lat, h0, (const double2 *)ctx->d_bwd, D.tp);
            else if (o.sel && K.track_post_form == 0)
                hipLaunchKernelGGL(k_fwd_post<true>, dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, tab, PL.packed, PL.grow + g0, PL.lat, h0, (const double2 *)ctx->d_bwd, D.tp);
            else if (!o.sel)
                hipLaunchKernelGGL(k_fwd_post<false>, dim3((cnt + KF_THREADS - 1) / KF_THREADS), dim3(KF_THREADS), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, tab, PL.packed, PL.grow + g0, PL.lat, h0, (const double2 *)ctx->d_bwd, D.tp);
#endif
            else // (the forward pass with the posteriors behind a checkpointed backward pass: the one form the release library has)
                hipLaunchKernelGGL(k_fwd_post_t<256>, dim3((cnt + 255u) / 256u), dim3(256), 0, s, D.d_offsets, PL.neff,
                                   PL.order + first, cnt, tab, PL.packed, PL.grow + g0, PL.lat, h0, (const double2 *)ctx->d_bwd, D.tp);
            break;
        }
        case K_WIN: { // (o.b > o.a: the wave-groups [o.a, o.b) of the run instead of all of it)
            const bool part = o.b > o.a;
            const uint32_t wg0 = part ? o.a : F.segb[o.run];
            const uint32_t first = part ? o.a * 64u : seg_first(F.segb, o.run);
            const uint32_t cnt = part ? (uint32_t)(std::min<uint64_t>((uint64_t)o.b * 64u, nprot) - first) : seg_count(F.segb, o.run);
            const unsigned abk = (cnt + KA_THREADS - 1) / KA_THREADS;
#define LAUNCH_WIN(NC, ROLE)                                                                                       \
    hipLaunchKernelGGL((k_win<NC, ROLE>), dim3(abk), dim3(KA_THREADS), 0, s, D.d_codes, D.d_offsets, PL.neff,      \
                       PL.order + first, cnt, tab, PL.packed, PL.grow + wg0, tg, h0)
            if (o.role == 2) LAUNCH_WIN(1, 2);
            else if (o.role == 3) LAUNCH_WIN(1, 3);
            else
                switch (o.nc) {
                case 1: LAUNCH_WIN(1, 0); break;
                case 2: LAUNCH_WIN(2, 0); break;
                case 3: LAUNCH_WIN(3, 0); break;
                default: LAUNCH_WIN(4, 0); break;
                }
#undef LAUNCH_WIN
            break;
        }
        case K_FINISH: {
            const uint32_t lcnt = (uint32_t)std::min<uint64_t>((uint64_t)F.gl * 64u, nprot);
            const uint32_t f0 = o.sel ? 0u : o.a, cnt = o.sel ? lcnt : nprot - o.a;
            hipLaunchKernelGGL(k_finish, dim3((cnt + 255u) / 256u), dim3(256), 0, s, PL.order + f0, cnt, D.d_rows[0], lmarg, h0, vend);
            break;
        }
        case K_POST: {
            const unsigned post_grid = (unsigned)((D.total_rows + POST_ROWS - 1) / POST_ROWS);
#define POST_ARGS D.d_offsets, PL.neff, PL.order, nprot, ngroups, PL.grow, D.gtab0, ctx->d_fwd, ctx->d_bwd, PL.bits, D.tp, o.a, o.b
            if (o.sel == 1) hipLaunchKernelGGL((k_post<true, false>), dim3(post_grid), dim3(64), 0, s, POST_ARGS);
            else if (o.sel == 2) hipLaunchKernelGGL((k_post<false, true>), dim3(post_grid), dim3(64), 0, s, POST_ARGS);
            else hipLaunchKernelGGL((k_post<true, true>), dim3(post_grid), dim3(64), 0, s, POST_ARGS);
#undef POST_ARGS
            break;
        }
        case K_REPLICATE: {
            SweepTargets tr{};
            for (int k = 1; k <= o.nc; ++k) tr.rows[k] = D.d_rows[G->first_point + o.m0 + (uint32_t)k - 1];
            hipLaunchKernelGGL(k_replicate, dim3((nprot + 255u) / 256u), dim3(256), 0, s, rows0, tr, (int)o.nc, nprot);
            break;
        }
        // ---- K-B: window tracks ----
        case K_TRACKS_RING: {
#define LAUNCH_KB(RING)                                                                                            \
    do {                                                                                                           \
        if (o.trk)                                                                                                 \
            hipLaunchKernelGGL((k_tracks<RING, true>), dim3(nprot), dim3(64), 0, s, D.d_codes, D.d_offsets, PL.neff, PL.order, \
                               nprot, tab, rows0, D.tp);                                                           \
        else                                                                                                       \
            hipLaunchKernelGGL((k_tracks<RING, false>), dim3(nprot), dim3(64), 0, s, D.d_codes, D.d_offsets, PL.neff, PL.order, \
                               nprot, tab, rows0, D.tp);                                                           \
    } while (0)
            if (o.sel == 0) LAUNCH_KB(128);
            else if (o.sel == 1) LAUNCH_KB(256);
            else LAUNCH_KB(1024);
#undef LAUNCH_KB
            break;
        }
        case K_TRACKS20_WHOLE: {
            // the stream form keeps 32 proteins on one int32 position axis; a batch with a protein of >= 65535 residues is left
            // to the one-protein-at-a-time form. The planner raises a device flag for such a batch and both kernels are
            // enqueued: the one the flag rules out returns at once (no host round trip before the window kernel starts).
            const unsigned kb_grid = (nprot + KB_PROTEINS_PER_BLOCK - 1) / KB_PROTEINS_PER_BLOCK;
            if (o.trk)
                hipLaunchKernelGGL(k_tracks20<true>, dim3(kb_grid), dim3(64), 0, s, D.d_codes, D.d_offsets, PL.neff, PL.order, nprot,
                                   D.total, tab, rows0, D.tp, D.huge, (uint32_t)o.sel, (const uint4 *)nullptr, (const uint32_t *)nullptr);
            else
                hipLaunchKernelGGL(k_tracks20<false>, dim3(kb_grid), dim3(64), 0, s, D.d_codes, D.d_offsets, PL.neff, PL.order, nprot,
                                   D.total, tab, rows0, D.tp, D.huge, (uint32_t)o.sel, (const uint4 *)nullptr, (const uint32_t *)nullptr);
            break;
        }
        case K_TRACKS20_LIST: {
            const unsigned kb_grid = (nprot + KB_PROTEINS_PER_BLOCK - 1) / KB_PROTEINS_PER_BLOCK;
            hipLaunchKernelGGL(k_tracks20<false>, dim3(std::min(kb_grid, 4096u)), dim3(64), 0, s, D.d_codes, D.d_offsets, PL.neff,
                               PL.order, nprot, D.total, tab, rows0, D.tp, D.huge, 0u, PL.fblist, D.fbcount);
            break;
        }
        case K_TRACKS20S: {
            const unsigned kb_grid = (nprot + KB_PROTEINS_PER_BLOCK - 1) / KB_PROTEINS_PER_BLOCK;
            if (o.trk) {
                // (input order behind the long proteins: see the kernel's segment table)
                const unsigned na = std::min(kb_grid, K.track_consec);
                hipLaunchKernelGGL(k_tracks20s<true>, dim3(na + kb_grid), dim3(64), K.track_kb_lds, s, D.d_codes, PL.order, nprot, D.total, tab,
                                   ctx->d_divtab, rows0, D.tp, D.huge, D.d_offsets, PL.neff, (uint32_t)na);
                // (PAPAfi / PAPAllr / PAPAllr2 = the tracks' values at the centre the kernel above has just found)
                hipLaunchKernelGGL(k_papa_from_tracks, dim3((nprot + 255u) / 256u), dim3(256), 0, s, D.d_offsets, PL.neff, nprot, rows0, D.tp, D.huge);
            } else
                hipLaunchKernelGGL(k_tracks20s<false>, dim3(kb_grid), dim3(64), 0, s, D.d_codes, PL.order, nprot, D.total, tab,
                                   ctx->d_divtab, rows0, D.tp, D.huge, D.d_offsets, PL.neff, 0u);
            break;
        }
        case K_TRACKS20F: {
            const dim3 grid((o.a + o.b - 1) / o.b);
            if (o.sel)
                hipLaunchKernelGGL(k_tracks20f<true>, grid, dim3(64), 0, s, D.d_codes, PL.order, o.a, D.total, tab, ctx->d_divtab,
                                   rows0, D.huge, PL.clist, PL.crow, D.ccount, PL.fblist, D.fbcount);
            else
                hipLaunchKernelGGL(k_tracks20f<false>, grid, dim3(64), 0, s, D.d_codes, PL.order, o.a, D.total, tab, ctx->d_divtab,
                                   rows0, D.huge, PL.clist, PL.crow, D.ccount, PL.fblist, D.fbcount);
            break;
        }
        case K_TRACKSL: {
            const uint32_t g0 = o.a, g1 = o.b, base = g0 * 64u;
            hipLaunchKernelGGL(k_tracksL, dim3((g1 - g0 + KL_THREADS / 64 - 1) / (KL_THREADS / 64)), dim3(KL_THREADS), 0, s, PL.order,
                               nprot, g0, g1, tab, ctx->d_divtab, PL.packed, PL.grow, rows0, D.huge, PL.clist + base,
                               PL.crow + base, D.ccount + o.seg, PL.fblist, D.fbcount, F.kb_prio_lane);
            break;
        }
        case K_REFINE: {
            // a resident grid that strides over the list: 8 one-wave blocks of 20 KB per CU hold the LDS of every CU until the
            // list is through - 7 per CU when the next call may plan beside this kernel (its planning kernels take 8 / 16 KB
            // per block; with 32 KB they waited for this kernel's end: 3.3 ms)
            const unsigned rounds = (o.b + RF_SLOTS - 1) / RF_SLOTS;
            if (ctx->value_tolerance && o.sel)
                hipLaunchKernelGGL((k_refine_centres<true, true>), dim3(std::min(rounds, K.rf_grid)), dim3(64), 0, s, D.d_codes, D.total, tab,
                                   ctx->d_divtab, rows0, D.huge, PL.clist + o.a, PL.crow + o.a, D.ccount + o.seg, F.kb_prio_refine);
            else if (ctx->value_tolerance)
                hipLaunchKernelGGL((k_refine_centres<false, true>), dim3(std::min(rounds, F.tail_allowed ? K.rf_grid / 8u * 7u : K.rf_grid)),
                                   dim3(64), 0, s, D.d_codes, D.total, tab, ctx->d_divtab, rows0, D.huge, PL.clist + o.a,
                                   PL.crow + o.a, D.ccount + o.seg, F.kb_prio_refine);
            else if (o.sel)
                hipLaunchKernelGGL(k_refine_centres<true>, dim3(std::min(rounds, K.rf_grid)), dim3(64), 0, s, D.d_codes, D.total, tab,
                                   ctx->d_divtab, rows0, D.huge, PL.clist + o.a, PL.crow + o.a, D.ccount + o.seg, F.kb_prio_refine);
            else
                hipLaunchKernelGGL(k_refine_centres<false>, dim3(std::min(rounds, F.tail_allowed ? K.rf_grid / 8u * 7u : K.rf_grid)),
                                   dim3(64), 0, s, D.d_codes, D.total, tab, ctx->d_divtab, rows0, D.huge, PL.clist + o.a,
                                   PL.crow + o.a, D.ccount + o.seg, F.kb_prio_refine);
            break;
        }
        case K_COPY_WINDOW_FIELDS:
            hipLaunchKernelGGL(k_copy_window_fields, dim3((nprot + 255u) / 256u), dim3(256), 0, s,
                               (const plaac_row *)D.d_rows[C.groups[(size_t)G->kb_base].first_point], rows0, nprot);
            break;
        case K_LLR_AT_CENTRE_LISTED:
            hipLaunchKernelGGL(k_llr_at_centre, dim3(std::min((nprot + 3u) / 4u, 16384u)), dim3(256), 0, s, D.d_codes, D.d_offsets,
                               PL.neff, nprot, tab, (const plaac_row *)D.d_rows[C.groups[o.a].first_point], rows0, PL.fblist,
                               D.fbcount, D.huge);
            break;
        case K_LLR_AT_CENTRE_ALL:
            hipLaunchKernelGGL(k_llr_at_centre, dim3(std::min((nprot + 3u) / 4u, 1u << 20)), dim3(256), 0, s, D.d_codes, D.d_offsets,
                               PL.neff, nprot, tab, (const plaac_row *)D.d_rows[C.groups[o.a].first_point], rows0);
            break;
        default: return fail(ctx, PLAAC_ERR_DEVICE, "unknown operation in the schedule");
        }
    }
    return PLAAC_OK;
}

// CallKind of a call: the sweep groups of the points and what K-B needs to know about them (host only)
void describe_groups(const plaac_ctx *ctx, const plaac_params *points, uint32_t npoints, bool tracks,
                     const std::vector<DevTables> *host_tabs, sched::CallKind &C) {
    using namespace sched;
    const Knobs &K = ctx->knobs;
    C.groups.clear();
    for (uint32_t i = 0; i < npoints; ++i) { // same tables up to the core length
        bool placed = false;
        for (GroupKind &g : C.groups) {
            plaac_params a = points[g.first_point], b = points[i];
            a.corelength = b.corelength = 0;
            // (members of a group are consecutive points in the callers this library has; a group is first_point .. + members)
            if (std::memcmp(&a, &b, sizeof a) == 0 && g.first_point + g.members == i) {
                ++g.members;
                placed = true;
                break;
            }
        }
        if (!placed) {
            GroupKind g;
            g.first_point = i;
            C.groups.push_back(g);
        }
    }
    for (size_t g = 0; g < C.groups.size(); ++g) {
        GroupKind &G = C.groups[g];
        const plaac_params &P = points[G.first_point];
        G.fi_int = host_tabs ? (*host_tabs)[g].fi_int != 0 : ctx->fi_int;
        G.wmax = std::max(P.ww1 / 2, std::max(P.ww2 / 2, P.ww3 / 2));
        G.fast20 = P.ww1 / 2 == TW && P.ww2 / 2 == TW && P.ww3 / 2 == TW && !K.generic_tracks;
        // K-B base: an earlier group whose window tracks differ only through the llr table (another alpha of a sweep); such a
        // group needs PAPAllr / PAPAllr2 at the known PAPA centre only (k_llr_at_centre)
        G.kb_base = -1;
        if (P.ww3 / 2 <= LLRAT_MAXW)
            for (size_t h = 0; h < g && G.kb_base < 0; ++h) {
                const plaac_params &Q = points[C.groups[h].first_point];
                if (P.ww1 == Q.ww1 && P.ww2 == Q.ww2 && P.ww3 == Q.ww3 && P.adjustprolines == Q.adjustprolines &&
                    std::memcmp(P.cc, Q.cc, sizeof P.cc) == 0 && std::memcmp(P.hydro2, Q.hydro2, sizeof P.hydro2) == 0 &&
                    std::memcmp(P.charge, Q.charge, sizeof P.charge) == 0 && std::memcmp(P.lodpapa, Q.lodpapa, sizeof P.lodpapa) == 0)
                    G.kb_base = (long)h;
            }
        // the lane-per-protein form of the filter tier (k_tracksL): summary mode, half windows of 20, FoldIndex in integers
        G.lane_possible = !tracks && K.kb_filter && K.kb_lane && !K.generic_tracks && !K.per_protein_tracks && P.ww1 / 2 == TW &&
                          P.ww2 / 2 == TW && P.ww3 / 2 == TW && G.fi_int && K.fi_int_allowed && G.kb_base < 0;
        G.core_par_tables = npoints == 1 && ctx->core_par_ok && std::memcmp(&points[G.first_point], &ctx->params, sizeof(plaac_params)) == 0;
    }
}

} // namespace

// One planned pass over a resident batch for `npoints` parameter sets (npoints == 1: the ordinary call).
// Points whose tables differ only in the core length form a group: the plan, the packed copy, the forward pass,
// the window tracks and Viterbi + traceback run once per group; only the two prefix-sum window searches run per
// core length (inside the same kernels, up to MAXC at a time). The WHAT and WHERE of the call is decided and listed by
// schedule.hip.inc; this function allocates, hands the two halves of the schedule to the launcher and keeps the books.
static plaac_status score_points(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets, uint32_t nprot,
                                 uint64_t total_residues, const plaac_params *points, uint32_t npoints,
                                 plaac_row *const *d_rows, const plaac_tracks *d_tracks, void *stream_) {
    using namespace sched;
    if (nprot == 0 || npoints == 0) return PLAAC_OK;
    if (!d_offsets || !d_rows || !points || (!d_codes && total_residues))
        return fail(ctx, PLAAC_ERR_ARG, "null device buffer");
    if ((uintptr_t)d_codes & 15u) return fail(ctx, PLAAC_ERR_ARG, "d_codes must be 16-byte aligned");
    if (d_tracks && npoints != 1) return fail(ctx, PLAAC_ERR_ARG, "tracks are not available in sweeps");
    for (uint32_t i = 0; i < npoints; ++i) {
        if (!d_rows[i]) return fail(ctx, PLAAC_ERR_ARG, "null row array");
        if (const char *why = check_params(points[i])) return fail(ctx, PLAAC_ERR_ARG, why);
    }
    CallData D{};
    if (d_tracks) {
        D.tp = TrackPtrs{d_tracks->vit,   d_tracks->map,  d_tracks->charge,     d_tracks->hydro,
                         d_tracks->fi,    d_tracks->plaacllr, d_tracks->papa,   d_tracks->fix2,
                         d_tracks->plaacllrx2, d_tracks->papax2, d_tracks->post0, d_tracks->post1};
        const void *all[] = {D.tp.vit, D.tp.map, D.tp.charge, D.tp.hydro, D.tp.fi, D.tp.plaacllr,
                             D.tp.papa, D.tp.fix2, D.tp.plaacllrx2, D.tp.papax2, D.tp.post0, D.tp.post1};
        for (const void *q : all)
            if (!q) return fail(ctx, PLAAC_ERR_ARG, "tracks struct has a null array");
    }
    PL_HIP(ctx, hipSetDevice(ctx->device));
    const Knobs &K = ctx->knobs;
    const hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    plaac_status rc;

    // ---- what is asked for: the sweep groups, their tables
    CallKind C;
    C.nprot = nprot, C.ngroups = (nprot + 63u) / 64u, C.residues = total_residues, C.npoints = npoints, C.tracks = d_tracks != nullptr;
    C.ncalls = ctx->ncalls, C.overlap = ctx->overlap, C.last_chain_bound = ctx->last_chain_bound, C.last_mixed = ctx->last_mixed;
    C.last_single_summary = ctx->last_single_summary;
    const unsigned par = C.par(); // this call's plan buffers, `huge` word and body events
    C.old_tail = ctx->tail_open2[par];
    ctx->tail_open2[par] = false;
    const bool own_tables = !(npoints == 1 && std::memcmp(&points[0], &ctx->params, sizeof(plaac_params)) == 0);
    std::vector<DevTables> host_tabs;
    describe_groups(ctx, points, npoints, C.tracks, nullptr, C);
    if (C.groups.size() > (size_t)MAXG) return fail(ctx, PLAAC_ERR_ARG, "too many distinct parameter groups in one sweep");
    const size_t ng = C.groups.size();
    D.gtab0 = ctx->d_tab; // device tables: slot 0 keeps the ctx parameters (single-point calls), sweep groups use slots of d_tabs
    if (own_tables) {
        if ((rc = grow(ctx, ctx->d_tabs, ctx->cap_tabs, ng)) != PLAAC_OK) return rc;
        host_tabs.resize(ng);
        for (size_t g = 0; g < ng; ++g) fill_tables(points[C.groups[g].first_point], host_tabs[g]);
        describe_groups(ctx, points, npoints, C.tracks, &host_tabs, C); // (FoldIndex in integers: per group's tables)
        PL_HIP(ctx, hipMemcpyAsync(ctx->d_tabs, host_tabs.data(), sizeof(DevTables) * ng, hipMemcpyHostToDevice, st));
        PL_HIP(ctx, hipStreamSynchronize(st)); // `host_tabs` is a temporary
        D.gtab0 = ctx->d_tabs;
    }

    // ---- buffers whose size does not depend on the plan
    plaac_ctx::PlanBufs &PL = ctx->pl[par];
    if ((rc = grow(ctx, PL.neff, PL.cap_prot, (size_t)nprot)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, PL.order, PL.cap_order, (size_t)nprot)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, PL.grow, PL.cap_grow, (size_t)C.ngroups + 3)) != PLAAC_OK) return rc;
    if (C.single() && (rc = grow(ctx, PL.lat, PL.cap_lat, 3 * (size_t)nprot)) != PLAAC_OK) return rc;
    if (!d_tracks && K.kb_filter) { // lists of the filter form of the window kernel
        if ((rc = grow(ctx, PL.clist, PL.cap_clist, (size_t)nprot)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, PL.crow, PL.cap_crow, (size_t)nprot)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, PL.fblist, PL.cap_fblist, (size_t)nprot)) != PLAAC_OK) return rc;
    }
    while (ctx->gev.size() < 4 * (size_t)sched::MAXG) { // per group: forward pass done / long-run Viterbi pass done / long-run core search done / the other wave-groups' Viterbi pass done
        hipEvent_t e = nullptr;
        PL_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->gev.push_back(e);
    }
    // The groups of a sweep are independent of each other: every group gets its own three side streams (and its own
    // traceback-bit buffer), so that the long serial chains of all groups advance together
    while (!K.serial && ctx->gstreams.size() < 3 * (ng - 1)) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        hipStream_t a = nullptr;
        PL_HIP(ctx, hipStreamCreateWithPriority(&a, hipStreamNonBlocking, greatest));
        ctx->gstreams.push_back(a);
        a = nullptr;
        PL_HIP(ctx, hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0));
        ctx->gstreams_n.push_back(a);
        hipEvent_t e = nullptr;
        PL_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->gjev.push_back(e);
    }
    if (!K.serial && (!ctx->roles_assigned || ctx->roles_for != st)) { // normal-class roles: apart from `st`, too
        if (ctx->ncalls > 0) // (the probe kernels must not run beside a batch that is still being scored)
            PL_HIP(ctx, hipEventSynchronize(ctx->ev[(ctx->ncalls - 1) % plaac_ctx::EV_SETS][E_JOIN_IDX]));
        if (const char *why = assign_role_streams(ctx, 0, st)) return fail(ctx, PLAAC_ERR_DEVICE, why);
        ctx->roles_for = st;
        ctx->roles_assigned = true;
    }

    D.d_codes = d_codes, D.d_offsets = d_offsets, D.nprot = nprot, D.ngroups = C.ngroups, D.total = total_residues;
    D.points = points, D.d_rows = d_rows, D.tracks = C.tracks, D.PL = &PL, D.st = st;
    D.evs = ctx->ev[ctx->ncalls % plaac_ctx::EV_SETS];
    D.seq = (uint32_t)(ctx->ncalls + 1) | 0x80000000u;
    D.huge = ctx->d_huge + (ctx->ncalls & 3u);
    D.counters = ctx->d_kbcnt + (size_t)KB_COUNTER_WORDS * (ctx->ncalls & 3u);
    D.fbcount = D.counters, D.ccount = D.counters + 1, D.corecount = D.counters + 16;

    // ---- the head: planning kernels, group 0's window tracks where they need no packed copy, the row offsets
    Forms F;
    PlanWords W;
    KbState kb;
    Sched S;
    decide_entry(K, C, F);
    emit_head(K, C, F, S);
    if (!K.serial && !F.kb_after_pack) emit_tracks(K, C, F, W, 0, kb, S, false);
    emit_head_rows(K, C, F, S);
    if ((rc = run_ops(ctx, D, C, F, W, S.ops, 0)) != PLAAC_OK) return rc;
    PL_HIP(ctx, hipGetLastError());
    // ---- the one host round trip: the plan words (polled from pinned memory; if the word does not arrive - memory not
    // coherent on this system, a failed launch - the stream is synchronised as before and polling is switched off)
    {
        volatile uint32_t *hp = ctx->h_pin;
        bool arrived = false;
        if (ctx->poll_ok) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0;; ++spins) {
                if (__atomic_load_n(&ctx->h_pin[5], __ATOMIC_ACQUIRE) == D.seq) {
                    arrived = true;
                    ++ctx->polled;
                    break;
                }
                if ((spins & 1023u) == 1023u &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50 + (long)(nprot >> 16)))
                    break;
                cpu_relax();
            }
        }
        if (!arrived) {
            PL_HIP(ctx, hipStreamSynchronize(slot_stream(ctx, D, F.s_pack)));
            ++ctx->synced;
            if (ctx->poll_ok && hp[5] != D.seq) return fail(ctx, PLAAC_ERR_DEVICE, "plan words did not reach the host");
            ctx->poll_ok = false;
        }
    }
    W.total_rows = ctx->h_pin[0], W.rows_first = ctx->h_pin[2], W.long_groups = ctx->h_pin[3], W.long_rows = ctx->h_pin[4];
    for (int k = 0; k < SCAN_SEGS - 1; ++k) W.run_mark[k] = ctx->h_pin[6 + k];

    // ---- the forms of this call, the buffers they need
    decide_forms(K, C, W, F);
    D.total_rows = W.total_rows;
    D.core_lrows = F.core_lrows;
    if (F.core_long) { // scratch of k_core_*: the rows of the first CORE_MAX_GROUPS wave-groups
        if ((rc = grow(ctx, PL.corep, PL.cap_corep, F.core_lrows * 1024u * F.core_copies)) != PLAAC_OK) return rc;
        char *&cp = reinterpret_cast<char *&>(PL.corepart);
        if ((rc = grow(ctx, cp, PL.cap_corepart, F.core_lrows * 64u * sizeof(CorePart) * F.core_copies)) != PLAAC_OK) return rc;
    }
    if (F.use_core_list) {
        if ((rc = grow(ctx, PL.corelist, PL.cap_corelist, (size_t)nprot * ng)) != PLAAC_OK) return rc;
    }
    if ((rc = grow(ctx, PL.packed, PL.cap_packed, W.total_rows * 64u + 64u)) != PLAAC_OK) return rc;
    D.bits_stride = W.total_rows * 64u + 64u; // one traceback-bit buffer per group
    if ((rc = grow(ctx, PL.bits, PL.cap_bits, D.bits_stride * ng)) != PLAAC_OK) return rc;
    if (d_tracks) {
        if ((rc = grow(ctx, ctx->d_fwd, ctx->cap_fwd, W.total_rows * 16u * 64u + 64u)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, ctx->d_bwd, ctx->cap_bwd, W.total_rows * 16u * 64u + 64u)) != PLAAC_OK) return rc;
    }

    // ---- the body
    const size_t body = S.ops.size();
    emit_body(K, C, F, W, kb, S);
    if (K.stream_debug && ctx->ncalls < 4) std::fprintf(stderr, "plaac: schedule of call %lu\n%s", (unsigned long)ctx->ncalls, dump(S).c_str());
    if ((rc = run_ops(ctx, D, C, F, W, S.ops, body)) != PLAAC_OK) return rc;
    PL_HIP(ctx, hipGetLastError());
    ctx->last_chain_bound = F.chain_bound;
    ctx->last_mixed = F.mixed;
    ctx->last_single_summary = C.single() && !d_tracks;
    ctx->ncalls++;
    return PLAAC_OK;
}

// Test hook (see schedule.hip.inc, read_knobs): key without the PLAAC_ prefix; value == NULL removes it. Read by the contexts
// created afterwards. The release library knows the form-forcing keys only; the diagnostic build every key.
plaac_status plaac_debug_set_knob(const char *key, const char *value) {
    if (!key) return PLAAC_ERR_ARG;
    bool diag_only = false;
    if (!sched::knob_key_known(key, &diag_only)) return PLAAC_ERR_ARG;
#ifndef PLAAC_DIAG
    if (diag_only) return PLAAC_ERR_UNSUPPORTED; // (a form that only `make DIAG=1` compiles)
#endif
    try {
        sched::KnobHook &h = sched::knob_hook();
        std::lock_guard<std::mutex> lock(h.mu);
        if (value) h.v[key] = value;
        else h.v.erase(key);
    } catch (...) {
        return PLAAC_ERR_NOMEM;
    }
    return PLAAC_OK;
}
int plaac_diag_build(void) {
#ifdef PLAAC_DIAG
    return 1;
#else
    return 0;
#endif
}

long plaac_debug_schedule(const plaac_sched_query *q, char *buf, size_t cap) {
    using namespace sched;
    if (!q || !buf || q->nprot == 0 || q->npoints == 0 || q->ngroups_sweep == 0 || q->ngroups_sweep > (uint32_t)MAXG) return -1;
    const Knobs K = read_knobs();
    CallKind C;
    C.nprot = q->nprot, C.ngroups = (q->nprot + 63u) / 64u, C.residues = q->residues, C.npoints = q->npoints, C.tracks = q->tracks != 0;
    C.ncalls = q->ncalls, C.overlap = q->overlap != 0, C.last_chain_bound = q->last_chain_bound != 0;
    C.last_mixed = q->last_mixed != 0, C.last_single_summary = q->last_single_summary != 0, C.old_tail = q->old_tail != 0;
    uint32_t pt = 0;
    for (uint32_t g = 0; g < q->ngroups_sweep; ++g) {
        GroupKind G;
        G.first_point = pt, G.members = std::max(1u, q->group_members[g]);
        pt += G.members;
        G.kb_base = q->kb_base[g] >= 0 && q->kb_base[g] < (int32_t)g ? q->kb_base[g] : -1;
        G.fast20 = q->fast20 != 0 && !K.generic_tracks, G.wmax = q->wmax, G.fi_int = q->lane_possible != 0;
        G.lane_possible = q->lane_possible != 0 && !C.tracks && K.kb_filter && K.kb_lane && !K.generic_tracks && !K.per_protein_tracks &&
                          K.fi_int_allowed && G.kb_base < 0;
        G.core_par_tables = q->core_par_tables != 0 && q->npoints == 1;
        C.groups.push_back(G);
    }
    if (pt != q->npoints) return -1;
    PlanWords W;
    W.total_rows = q->total_rows, W.rows_first = q->rows_first, W.long_groups = q->long_groups, W.long_rows = q->long_rows;
    for (int k = 0; k < SCAN_SEGS - 1; ++k) W.run_mark[k] = q->run_mark[k];
    Forms F;
    KbState kb;
    Sched S;
    decide_entry(K, C, F);
    emit_head(K, C, F, S);
    if (!K.serial && !F.kb_after_pack) emit_tracks(K, C, F, W, 0, kb, S, false);
    emit_head_rows(K, C, F, S);
    const size_t body = S.ops.size();
    decide_forms(K, C, W, F);
    emit_body(K, C, F, W, kb, S);
    std::string out;
    char line[320];
    std::snprintf(line, sizeof line,
                  "F par=%u head_aside=%d tail_allowed=%d kb_after_pack=%d chain_bound=%d latency_mode=%d mixed=%d lat_all=%d gl=%u "
                  "use_core_list=%d sweep_lat=%d core_long=%d kb_deferred=%d maybe_huge=%d ka_wait=%d runs=%zu body=%zu serial=%d\n",
                  C.par(), F.head_aside, F.tail_allowed, F.kb_after_pack, F.chain_bound, F.latency_mode, F.mixed, F.lat_all, F.gl,
                  F.use_core_list, F.sweep_lat, F.core_long, F.kb_deferred, F.maybe_huge, F.ka_wait, F.ntseg(), body, K.serial ? 1 : 0);
    out += line;
    std::snprintf(line, sizeof line, "ALIAS %u %u\nALIAS %u %u\n", hi(R_BWD), hi(R_WIN2), no(R_WIN2), no(R_WIN));
    out += line;
    out += dump(S);
    if (out.size() + 1 > cap) return -1;
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (long)out.size();
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

plaac_status plaac_ctx_set_value_tolerance(plaac_ctx *ctx, int on) {
    if (!ctx) return PLAAC_ERR_ARG;
    ctx->value_tolerance = on != 0;
    return PLAAC_OK;
}

plaac_status plaac_ctx_set_overlap(plaac_ctx *ctx, int on) {
    if (!ctx) return PLAAC_ERR_ARG;
    ctx->overlap = on != 0;
    return PLAAC_OK;
}

plaac_status plaac_clock_probe(plaac_ctx *ctx, uint32_t micros, double *mhz) {
    if (!ctx || !mhz || micros == 0 || micros > 2000000u) return PLAAC_ERR_ARG;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    // a stream and a result buffer of the probe's own: the scoring streams are not touched (the probe runs beside them)
    hipStream_t s = nullptr;
    unsigned long long *d = nullptr, h[2] = {0, 0};
    PL_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipError_t e = hipMalloc((void **)&d, sizeof h);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, (unsigned long long)micros * 100ull, d);
        e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (d) (void)hipFree(d);
    (void)hipStreamDestroy(s);
    if (e != hipSuccess) return fail(ctx, PLAAC_ERR_DEVICE, hipGetErrorString(e));
    if (h[0] == 0) return fail(ctx, PLAAC_ERR_DEVICE, "clock probe returned nothing");
    // s_sleep n waits 64 n cycles (plus up to 64): 127 -> 8,128 .. 8,192; the midpoint prices a step
    *mhz = (double)h[1] * 8160.0 / ((double)h[0] / 100.0);
    return PLAAC_OK;
}

plaac_status plaac_calibration_reads(plaac_ctx *ctx, const uint8_t *d_codes, uint64_t total_residues, void *stream_) {
    if (!ctx || !d_codes) return PLAAC_ERR_ARG;
    if ((uintptr_t)d_codes & 15u) return fail(ctx, PLAAC_ERR_ARG, "d_codes must be 16-byte aligned");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : ctx->stream;
    if (!ctx->d_flag) PL_HIP(ctx, hipMalloc((void **)&ctx->d_flag, sizeof(uint32_t)));
    const unsigned grid = (unsigned)ctx->num_cus * 16u;
    hipLaunchKernelGGL((k_calib_read<16, 0>), dim3(grid), dim3(256), 0, st, d_codes, total_residues, ctx->d_flag);
    hipLaunchKernelGGL((k_calib_read<4, 0>), dim3(grid), dim3(256), 0, st, d_codes, total_residues, ctx->d_flag);
    hipLaunchKernelGGL((k_calib_read<4, 1>), dim3(grid), dim3(256), 0, st, d_codes, total_residues, ctx->d_flag);
    PL_HIP(ctx, hipGetLastError());
    return PLAAC_OK;
}

int plaac_fi_integer_form(const plaac_params *params, int32_t info[6]) {
    if (!params) return 0;
    DevTables *h = new (std::nothrow) DevTables();
    if (!h) return 0;
    int32_t tmp[6] = {0, 0, 0, 0, 0, 0};
    derive_fi_int(*params, *h, tmp);
    const int ok = h->fi_int;
    tmp[0] = h->fi_A2;
    tmp[1] = h->fi_B2;
    tmp[2] = h->fi_C2;
    delete h;
    if (info)
        for (int i = 0; i < 6; ++i) info[i] = ok ? tmp[i] : 0;
    return ok;
}

plaac_status plaac_last_exact_fallbacks(plaac_ctx *ctx, uint32_t *count) {
    if (!ctx || !count) return PLAAC_ERR_ARG;
    *count = 0;
    if (ctx->ncalls == 0) return PLAAC_OK;
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_HIP(ctx, hipEventSynchronize(ctx->ev[(ctx->ncalls - 1) % plaac_ctx::EV_SETS][10 /* E_JOIN */]));
    size_t word = 0;
#ifdef PLAAC_DIAG // (diagnostic build only: PLAAC_DEBUG_COUNTER=16 reads another word of the call's counter set - 16: the core list's length)
    {
        std::string dbg;
        if (sched::knob_value("DEBUG_COUNTER", dbg)) word = (size_t)std::min(31L, std::max(0L, std::atol(dbg.c_str())));
    }
#endif
    PL_HIP(ctx, hipMemcpy(count, ctx->d_kbcnt + (size_t)KB_COUNTER_WORDS * ((ctx->ncalls - 1) & 3u) + word, sizeof(uint32_t),
                          hipMemcpyDeviceToHost));
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
        // one resident round of waves, each with an equal share of the residue stream (the kernel reads the total itself)
        const unsigned blocks = (unsigned)ctx->num_cus * HIST_BLOCKS_PER_CU;
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
static plaac_status copy_in_through(plaac_ctx *ctx, uint8_t *(&hs)[2], hipEvent_t (&evs)[2], void *dst, const void *src, size_t bytes,
                                    hipStream_t st) {
    if (bytes < (1u << 20)) {
        PL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        PL_HIP(ctx, hipStreamSynchronize(st)); // (pageable source: the caller may let go of it)
        return PLAAC_OK;
    }
    for (int i = 0; i < 2; ++i) {
        if (!hs[i]) PL_HIP(ctx, hipHostMalloc((void **)&hs[i], plaac_ctx::STAGE_BYTES, hipHostMallocDefault));
        if (!evs[i]) PL_HIP(ctx, hipEventCreateWithFlags(&evs[i], hipEventDisableTiming));
    }
    size_t done = 0;
    for (int k = 0; done < bytes; ++k) {
        const int b = k & 1;
        const size_t n = std::min(plaac_ctx::STAGE_BYTES, bytes - done);
        if (k >= 2) PL_HIP(ctx, hipEventSynchronize(evs[b])); // DMA out of this buffer has finished
        parallel_memcpy(hs[b], (const char *)src + done, n);
        PL_HIP(ctx, hipMemcpyAsync((char *)dst + done, hs[b], n, hipMemcpyHostToDevice, st));
        PL_HIP(ctx, hipEventRecord(evs[b], st));
        done += n;
    }
    PL_HIP(ctx, hipStreamSynchronize(st)); // the staging buffers are free again when this returns
    return PLAAC_OK;
}
static plaac_status copy_in(plaac_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes < (1u << 20)) {
        PL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return PLAAC_OK;
    }
    return copy_in_through(ctx, ctx->h_stage, ctx->stage_ev, dst, src, bytes, st);
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

// host batch -> device copies (d_codes / d_offsets, grown on demand) on stream `sx`, codes validated; returns with the
// copies complete
static plaac_status stage_in_to(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                uint64_t *total_out, uint8_t *&d_codes, size_t &cap_codes, uint64_t *&d_offsets,
                                size_t &cap_offs, hipStream_t sx) {
    if (!offsets) return fail(ctx, PLAAC_ERR_ARG, "null offsets");
    if (offsets[0] != 0) return fail(ctx, PLAAC_ERR_ARG, "offsets[0] must be 0");
    for (uint32_t p = 0; p < nprot; ++p) {
        if (offsets[p + 1] < offsets[p]) return fail(ctx, PLAAC_ERR_ARG, "offsets must be non-decreasing");
        if (offsets[p + 1] - offsets[p] >= 0x7fffffffull) return fail(ctx, PLAAC_ERR_ARG, "a record of 2^31 residues or more");
    }
    const uint64_t total = offsets[nprot];
    if (total && !codes) return fail(ctx, PLAAC_ERR_ARG, "null codes");
    plaac_status rc;
    if ((rc = grow(ctx, d_codes, cap_codes, (size_t)total + 64)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, d_offsets, cap_offs, (size_t)nprot + 1)) != PLAAC_OK) return rc;
    if (total && (rc = copy_in(ctx, d_codes, codes, total, sx)) != PLAAC_OK) return rc;
    if ((rc = copy_in(ctx, d_offsets, offsets, sizeof(uint64_t) * ((size_t)nprot + 1), sx)) != PLAAC_OK) return rc;
    // codes must be 0..21 (they index the kernels' tables): checked on the device, the upload is there anyway
    if (total) {
        if (!ctx->d_flag) PL_HIP(ctx, hipMalloc((void **)&ctx->d_flag, sizeof(uint32_t)));
        PL_HIP(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(uint32_t), sx));
        const unsigned vb = (unsigned)std::min<uint64_t>(((total >> 4) + 255u) / 256u + 1u, 2048u);
        hipLaunchKernelGGL(k_validate, dim3(vb), dim3(256), 0, sx, d_codes, total, ctx->d_flag);
        PL_HIP(ctx, hipMemcpyAsync(ctx->h_pin + 1, ctx->d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, sx));
        PL_HIP(ctx, hipStreamSynchronize(sx));
        if (ctx->h_pin[1]) return fail(ctx, PLAAC_ERR_ARG, "residue code > 21");
    } else {
        PL_HIP(ctx, hipStreamSynchronize(sx));
    }
    *total_out = total;
    return PLAAC_OK;
}

static plaac_status stage_in(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                             uint64_t *total_out) {
    return stage_in_to(ctx, codes, offsets, nprot, total_out, ctx->d_codes, ctx->cap_codes, ctx->d_offsets, ctx->cap_offs,
                       ctx->stream);
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

// the context's device track arrays, sized for `total` residues; trimmed stops' entries made deterministic (0 / NaN)
static plaac_status device_tracks(plaac_ctx *ctx, uint64_t total, plaac_tracks &dt) {
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
        if ((rc = device_tracks(ctx, total, dt)) != PLAAC_OK) return rc;
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

// plotsomefastas' per-residue table for a batch of selected records, made on the device (round 5, late; format_device.hip.inc):
// scored in track mode like plaac_score with tracks, but the twelve arrays stay on the device and the table's TEXT comes back.
// *table: malloc'ed (plaac_table_free), or NULL with *needs_host != 0 when a value is one the host's formatter has to take
// (>= 1e9, an infinity) - then score the batch with plaac_score and format on the host as before.
plaac_status plaac_score_tracks_table(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot, const char *labels,
                                      const uint64_t *label_off, plaac_row *rows, char **table, uint64_t *table_len, int *needs_host) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (!table || !table_len || !needs_host) return fail(ctx, PLAAC_ERR_ARG, "null argument");
    *table = nullptr, *table_len = 0, *needs_host = 0;
    if (nprot == 0) return PLAAC_OK;
    if (!labels || !label_off) return fail(ctx, PLAAC_ERR_ARG, "null labels");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->slots_busy) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_tracks_table: batches are pending on this context");
    uint64_t total = 0;
    plaac_status rc = stage_in(ctx, codes, offsets, nprot, &total);
    if (rc != PLAAC_OK) return rc;
    if (total >= 0xffffffffull) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_tracks_table: a batch of at most 2^32 - 2 residues");
    if ((rc = grow(ctx, ctx->d_rows, ctx->cap_rows, (size_t)nprot)) != PLAAC_OK) return rc;
    plaac_tracks dt{};
    if ((rc = device_tracks(ctx, total, dt)) != PLAAC_OK) return rc;
    if ((rc = plaac_score_device(ctx, ctx->d_codes, ctx->d_offsets, nprot, total, ctx->d_rows, &dt, ctx->stream)) != PLAAC_OK) return rc;
    if (rows && (rc = copy_out(ctx, rows, ctx->d_rows, sizeof(plaac_row) * (size_t)nprot, ctx->stream)) != PLAAC_OK) return rc;
    if (total == 0) return PLAAC_OK;
    // the text: the slot buffers of the text pipeline serve (no batch is pending): labels -> d_text, label_off -> d_starts,
    // line lengths -> d_len, line offsets -> d_toffs
    plaac_ctx::Slot &S = ctx->slot[0];
    const uint64_t lab_bytes = label_off[nprot];
    const uint32_t nres = (uint32_t)total;
    const unsigned ns = (nres + FA_SCAN - 1u) / FA_SCAN;
    size_t cap_tot = S.d_total ? 1 : 0;
    if ((rc = grow(ctx, S.d_text, S.cap_text, (size_t)lab_bytes + 16)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_starts, S.cap_starts, (size_t)nprot + 1)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_len, S.cap_len, (size_t)nres)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_toffs, S.cap_toffs, (size_t)nres + 1)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_bsum, S.cap_bsum, (size_t)ns)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_total, cap_tot, (size_t)1)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_fmtflags, S.cap_fmtflags, (size_t)4)) != PLAAC_OK) return rc;
    if ((rc = copy_in(ctx, S.d_text, labels, (size_t)lab_bytes, ctx->stream)) != PLAAC_OK) return rc;
    if ((rc = copy_in(ctx, S.d_starts, label_off, sizeof(uint64_t) * ((size_t)nprot + 1), ctx->stream)) != PLAAC_OK) return rc;
    PL_HIP(ctx, hipMemsetAsync(S.d_fmtflags, 0, 4 * sizeof(uint32_t), ctx->stream));
    const FmtTrackPtrs T{dt.vit, dt.map, {dt.charge, dt.hydro, dt.fi, dt.plaacllr, dt.papa, dt.fix2, dt.plaacllrx2, dt.papax2}, dt.post0, dt.post1};
    const unsigned nb = (unsigned)((total + 255u) / 256u);
    hipLaunchKernelGGL(k_format_track_lines<false>, dim3(nb), dim3(256), 0, ctx->stream, ctx->d_codes, ctx->d_offsets, nprot, total, ctx->d_rows, T,
                       S.d_text, S.d_starts, S.d_len, (const uint64_t *)nullptr, (char *)nullptr, S.d_fmtflags);
    hipLaunchKernelGGL(k_fasta_block_sums, dim3(ns), dim3(FA_SCAN), 0, ctx->stream, S.d_len, nres, S.d_bsum);
    hipLaunchKernelGGL(k_fasta_offsets, dim3(ns), dim3(FA_SCAN), 0, ctx->stream, S.d_len, nres, S.d_bsum, S.d_toffs, S.d_total);
    struct {
        unsigned long long total;
        uint32_t flags[4];
    } h{};
    PL_HIP(ctx, hipMemcpyAsync(&h.total, S.d_total, sizeof h.total, hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipMemcpyAsync(h.flags, S.d_fmtflags, sizeof h.flags, hipMemcpyDeviceToHost, ctx->stream));
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h.flags[0]) {
        *needs_host = 1;
        return PLAAC_OK;
    }
    if ((rc = grow(ctx, S.d_table, S.cap_table, (size_t)h.total + 16)) != PLAAC_OK) return rc;
    hipLaunchKernelGGL(k_format_track_lines<true>, dim3(nb), dim3(256), 0, ctx->stream, ctx->d_codes, ctx->d_offsets, nprot, total, ctx->d_rows, T,
                       S.d_text, S.d_starts, S.d_len, S.d_toffs, S.d_table, S.d_fmtflags);
    // (hundreds of MB per batch, touched once: huge pages where the host grants them on request)
    const size_t huge = 2u << 20, out_bytes = ((size_t)h.total + 1 + huge - 1) / huge * huge;
    char *out = (char *)std::aligned_alloc(huge, out_bytes);
    if (!out) return fail(ctx, PLAAC_ERR_NOMEM, "out of host memory");
    (void)madvise(out, out_bytes, MADV_HUGEPAGE);
    if ((rc = copy_out(ctx, out, S.d_table, (size_t)h.total, ctx->stream)) != PLAAC_OK) {
        std::free(out);
        return rc;
    }
    PL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *table = out;
    *table_len = h.total;
    return PLAAC_OK;
}
void plaac_table_free(char *table) { std::free(table); }

static plaac_status score_begin(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot, bool counting) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (ctx->slots_busy >= 2) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_begin: two batches are pending (call plaac_score_end)");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->xfer) PL_HIP(ctx, hipStreamCreateWithFlags(&ctx->xfer, hipStreamNonBlocking));
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_next];
    S.nprot = nprot;
    S.call_no = ~0ull;
    S.counted = counting;
    S.from_text = false;
    S.table_sized = false;
    if (nprot) {
        uint64_t total = 0;
        plaac_status rc = stage_in_to(ctx, codes, offsets, nprot, &total, S.d_codes, S.cap_codes, S.d_offsets, S.cap_offs, ctx->xfer);
        if (rc != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_rows, S.cap_rows, (size_t)nprot)) != PLAAC_OK) return rc;
        if (counting) { // the background pass (countaas / isvalidprotein, plaac.java:1698-1739) over the copy the scoring reads
            if (!S.d_counts) PL_HIP(ctx, hipMalloc(&S.d_counts, sizeof(unsigned long long) * NAA));
            if ((rc = plaac_histogram_device(ctx, S.d_codes, S.d_offsets, nprot, (int64_t *)S.d_counts, ctx->stream)) != PLAAC_OK) return rc;
        }
        const uint64_t call_no = ctx->ncalls;
        rc = plaac_score_device(ctx, S.d_codes, S.d_offsets, nprot, total, S.d_rows, nullptr, ctx->stream);
        if (rc != PLAAC_OK) return rc;
        S.call_no = call_no;
    }
    S.busy = true;
    ctx->slot_next ^= 1u;
    ++ctx->slots_busy;
    return PLAAC_OK;
}

plaac_status plaac_score_begin(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot) {
    return score_begin(ctx, codes, offsets, nprot, false);
}

// K1 (round 5): the same pipeline fed with FASTA TEXT - the records' bytes and where each begins - parsed and encoded on the
// device (fasta_device.hip.inc) into the slot the scoring kernels read. Upload and parse run on the copy stream, beside the
// kernels of the batch before; the codes are complete when the scoring call is made (the overlap contract).
static plaac_status begin_text(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec, int counting,
                               bool score) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (ctx->slots_busy >= 2) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_begin_text: two batches are pending (call plaac_score_end_text)");
    if (nrec && (!text || !starts)) return fail(ctx, PLAAC_ERR_ARG, "null text / starts");
    for (uint32_t i = 0; i < nrec; ++i)
        if (starts[i + 1] < starts[i] || starts[i + 1] > text_len || starts[i + 1] - starts[i] >= 0x7fffffffull)
            return fail(ctx, PLAAC_ERR_ARG, "record starts must be non-decreasing and inside the text, records below 2^31 bytes");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->xfer) PL_HIP(ctx, hipStreamCreateWithFlags(&ctx->xfer, hipStreamNonBlocking));
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_next];
    S.nprot = nrec;
    S.call_no = ~0ull;
    S.counted = counting != 0;
    S.from_text = true;
    S.table_sized = false;
    S.hist_only = !score;
    S.nres = 0;
    if (nrec) {
        plaac_status rc;
        size_t cap_tot = S.d_total ? 1 : 0;
        if ((rc = grow(ctx, S.d_text, S.cap_text, (size_t)text_len + 16)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_starts, S.cap_starts, (size_t)nrec + 1)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_len, S.cap_len, (size_t)nrec)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_blank, S.cap_blank, (size_t)nrec)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_total, cap_tot, (size_t)1)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_codes, S.cap_codes, (size_t)text_len + 64)) != PLAAC_OK) return rc; // (a record's codes are fewer than its bytes)
        if ((rc = grow(ctx, S.d_offsets, S.cap_offs, (size_t)nrec + 1)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_rows, S.cap_rows, (size_t)nrec)) != PLAAC_OK) return rc;
        if ((rc = copy_in(ctx, S.d_text, text, (size_t)text_len, ctx->xfer)) != PLAAC_OK) return rc;
        if ((rc = copy_in(ctx, S.d_starts, starts, sizeof(uint64_t) * ((size_t)nrec + 1), ctx->xfer)) != PLAAC_OK) return rc;
        static const FastaLut lut = [] {
            FastaLut l;
            char all[256];
            for (int i = 0; i < 256; ++i) all[i] = (char)i;
            plaac_encode(all, 256, l.t);
            return l;
        }();
        const unsigned nb = (nrec + FA_RECS - 1u) / FA_RECS, ns = (nrec + FA_SCAN - 1u) / FA_SCAN;
        if ((rc = grow(ctx, S.d_ext, S.cap_ext, (size_t)nrec)) != PLAAC_OK) return rc;
        if ((rc = grow(ctx, S.d_bsum, S.cap_bsum, (size_t)ns)) != PLAAC_OK) return rc;
        hipLaunchKernelGGL(k_fasta_lengths, dim3(nb), dim3(FA_BLOCK), 0, ctx->xfer, S.d_text, S.d_starts, nrec, S.d_len, S.d_blank, S.d_ext);
        hipLaunchKernelGGL(k_fasta_block_sums, dim3(ns), dim3(FA_SCAN), 0, ctx->xfer, S.d_len, nrec, S.d_bsum);
        hipLaunchKernelGGL(k_fasta_offsets, dim3(ns), dim3(FA_SCAN), 0, ctx->xfer, S.d_len, nrec, S.d_bsum, S.d_offsets, S.d_total);
        hipLaunchKernelGGL(k_fasta_encode, dim3(nb), dim3(FA_BLOCK), 0, ctx->xfer, S.d_text, S.d_starts, nrec, S.d_ext, S.d_offsets, lut,
                           S.d_codes);
        unsigned long long total = 0;
        PL_HIP(ctx, hipMemcpyAsync(&total, S.d_total, sizeof total, hipMemcpyDeviceToHost, ctx->xfer));
        PL_HIP(ctx, hipStreamSynchronize(ctx->xfer));
        S.nres = total;
        if (counting) {
            if (!S.d_counts) PL_HIP(ctx, hipMalloc(&S.d_counts, sizeof(unsigned long long) * NAA));
            if ((rc = plaac_histogram_device(ctx, S.d_codes, S.d_offsets, nrec, (int64_t *)S.d_counts, ctx->stream)) != PLAAC_OK) return rc;
        }
        if (score) {
            const uint64_t call_no = ctx->ncalls;
            rc = plaac_score_device(ctx, S.d_codes, S.d_offsets, nrec, total, S.d_rows, nullptr, ctx->stream);
            if (rc != PLAAC_OK) return rc;
            S.call_no = call_no;
        } else {
            if (!S.hist_ev) PL_HIP(ctx, hipEventCreateWithFlags(&S.hist_ev, hipEventDisableTiming));
            PL_HIP(ctx, hipEventRecord(S.hist_ev, ctx->stream));
        }
    }
    S.busy = true;
    ctx->slot_next ^= 1u;
    ++ctx->slots_busy;
    return PLAAC_OK;
}
plaac_status plaac_score_begin_text(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                                    int counting) {
    return begin_text(ctx, text, text_len, starts, nrec, counting, true);
}
// ---- a text batch uploaded and parsed AHEAD of its scoring call, by another host thread (round 5, late) ----
// The thread that scores (plaac_score_begin_* / _end_*) spends 4.5 ms of a 262,144-record batch's 7.5 ms inside the upload;
// with plaac_text_upload an uploader thread of the host does that part for batch k + 1 while the scoring thread collects
// batch k (H2D beside D2H: the link is full duplex). One upload at a time per context; it touches nothing the scoring calls
// use (its own stream, pinned buffers and device buffers). plaac_score_begin_uploaded takes the batch over: the pending slot
// and the batch swap their device buffers.
static const FastaLut &fasta_lut() {
    static const FastaLut lut = [] {
        FastaLut l;
        char all[256];
        for (int i = 0; i < 256; ++i) all[i] = (char)i;
        plaac_encode(all, 256, l.t);
        return l;
    }();
    return lut;
}
plaac_status plaac_text_upload(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                               plaac_text_batch **out) {
    if (!ctx || !out) return PLAAC_ERR_ARG;
    ErrRedirect own_slot(&ctx->up_err); // (every message below goes to plaac_text_upload_error, not to the scoring thread's slot)
    *out = nullptr;
    if (nrec && (!text || !starts)) return fail(ctx, PLAAC_ERR_ARG, "plaac_text_upload: null text or starts");
    for (uint32_t i = 0; i < nrec; ++i)
        if (starts[i + 1] < starts[i] || starts[i + 1] > text_len || starts[i + 1] - starts[i] >= 0x7fffffffull)
            return fail(ctx, PLAAC_ERR_ARG, "plaac_text_upload: record starts must ascend inside the text, a record below 2^31 bytes");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    plaac_text_batch *tb = nullptr;
    {
        std::lock_guard<std::mutex> l(ctx->up_mu);
        if (!ctx->up) PL_HIP(ctx, hipStreamCreateWithFlags(&ctx->up, hipStreamNonBlocking));
        if (!ctx->up_pool.empty()) {
            tb = ctx->up_pool.back();
            ctx->up_pool.pop_back();
        }
    }
    if (!tb) tb = new (std::nothrow) plaac_text_batch();
    if (!tb) return fail(ctx, PLAAC_ERR_NOMEM, "plaac_text_upload: out of host memory");
    tb->ctx = ctx;
    tb->nrec = nrec;
    tb->total = 0;
    auto give_back = [&](plaac_status rc) {
        std::lock_guard<std::mutex> l(ctx->up_mu);
        ctx->up_pool.push_back(tb);
        return rc;
    };
    if (nrec) {
        plaac_status rc;
        size_t cap_tot = tb->d_total ? 1 : 0;
        const unsigned nb = (nrec + FA_RECS - 1u) / FA_RECS, ns = (nrec + FA_SCAN - 1u) / FA_SCAN;
        if ((rc = grow(ctx, tb->d_text, tb->cap_text, (size_t)text_len + 16)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_starts, tb->cap_starts, (size_t)nrec + 1)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_len, tb->cap_len, (size_t)nrec)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_blank, tb->cap_blank, (size_t)nrec)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_total, cap_tot, (size_t)1)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_codes, tb->cap_codes, (size_t)text_len + 64)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_offsets, tb->cap_offs, (size_t)nrec + 1)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_ext, tb->cap_ext, (size_t)nrec)) != PLAAC_OK) return give_back(rc);
        if ((rc = grow(ctx, tb->d_bsum, tb->cap_bsum, (size_t)ns)) != PLAAC_OK) return give_back(rc);
        if ((rc = copy_in_through(ctx, ctx->h_stage_up, ctx->stage_ev_up, tb->d_text, text, (size_t)text_len, ctx->up)) != PLAAC_OK) return give_back(rc);
        if ((rc = copy_in_through(ctx, ctx->h_stage_up, ctx->stage_ev_up, tb->d_starts, starts, sizeof(uint64_t) * ((size_t)nrec + 1), ctx->up)) != PLAAC_OK)
            return give_back(rc);
        hipLaunchKernelGGL(k_fasta_lengths, dim3(nb), dim3(FA_BLOCK), 0, ctx->up, tb->d_text, tb->d_starts, nrec, tb->d_len, tb->d_blank, tb->d_ext);
        hipLaunchKernelGGL(k_fasta_block_sums, dim3(ns), dim3(FA_SCAN), 0, ctx->up, tb->d_len, nrec, tb->d_bsum);
        hipLaunchKernelGGL(k_fasta_offsets, dim3(ns), dim3(FA_SCAN), 0, ctx->up, tb->d_len, nrec, tb->d_bsum, tb->d_offsets, tb->d_total);
        hipLaunchKernelGGL(k_fasta_encode, dim3(nb), dim3(FA_BLOCK), 0, ctx->up, tb->d_text, tb->d_starts, nrec, tb->d_ext, tb->d_offsets, fasta_lut(),
                           tb->d_codes);
        unsigned long long total = 0;
        hipError_t e = hipMemcpyAsync(&total, tb->d_total, sizeof total, hipMemcpyDeviceToHost, ctx->up);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->up);
        if (e != hipSuccess) return give_back(fail(ctx, PLAAC_ERR_DEVICE, hipGetErrorString(e)));
        tb->total = total;
    }
    *out = tb;
    return PLAAC_OK;
}

void plaac_text_batch_free(plaac_text_batch *tb) { // (an uploaded batch that will not be scored: its buffers go back to the context)
    if (!tb || !tb->ctx) return;
    std::lock_guard<std::mutex> l(tb->ctx->up_mu);
    tb->ctx->up_pool.push_back(tb);
}

plaac_status plaac_score_begin_uploaded(plaac_ctx *ctx, plaac_text_batch *tb, int counting) {
    if (!ctx || !tb || tb->ctx != ctx) return ctx ? fail(ctx, PLAAC_ERR_ARG, "plaac_score_begin_uploaded: not a batch of this context") : PLAAC_ERR_ARG;
    if (ctx->slots_busy >= 2) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_begin_uploaded: two batches are pending (call plaac_score_end_text)");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->xfer) PL_HIP(ctx, hipStreamCreateWithFlags(&ctx->xfer, hipStreamNonBlocking));
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_next];
    // the slot is free (its last batch has been collected): its buffers and the uploaded batch's change places
    std::swap(S.d_text, tb->d_text), std::swap(S.cap_text, tb->cap_text);
    std::swap(S.d_starts, tb->d_starts), std::swap(S.cap_starts, tb->cap_starts);
    std::swap(S.d_len, tb->d_len), std::swap(S.cap_len, tb->cap_len);
    std::swap(S.d_blank, tb->d_blank), std::swap(S.cap_blank, tb->cap_blank);
    std::swap(S.d_ext, tb->d_ext), std::swap(S.cap_ext, tb->cap_ext);
    std::swap(S.d_bsum, tb->d_bsum), std::swap(S.cap_bsum, tb->cap_bsum);
    std::swap(S.d_total, tb->d_total);
    std::swap(S.d_codes, tb->d_codes), std::swap(S.cap_codes, tb->cap_codes);
    std::swap(S.d_offsets, tb->d_offsets), std::swap(S.cap_offs, tb->cap_offs);
    const uint32_t nrec = tb->nrec;
    const uint64_t total = tb->total;
    plaac_text_batch_free(tb); // (with the slot's old buffers, for the next upload)
    S.nprot = nrec;
    S.call_no = ~0ull;
    S.counted = counting != 0;
    S.from_text = true;
    S.table_sized = false;
    S.hist_only = false;
    S.nres = total;
    if (nrec) {
        plaac_status rc;
        if ((rc = grow(ctx, S.d_rows, S.cap_rows, (size_t)nrec)) != PLAAC_OK) return rc;
        if (counting) {
            if (!S.d_counts) PL_HIP(ctx, hipMalloc(&S.d_counts, sizeof(unsigned long long) * NAA));
            if ((rc = plaac_histogram_device(ctx, S.d_codes, S.d_offsets, nrec, (int64_t *)S.d_counts, ctx->stream)) != PLAAC_OK) return rc;
        }
        const uint64_t call_no = ctx->ncalls;
        rc = plaac_score_device(ctx, S.d_codes, S.d_offsets, nrec, total, S.d_rows, nullptr, ctx->stream);
        if (rc != PLAAC_OK) return rc;
        S.call_no = call_no;
    }
    S.busy = true;
    ctx->slot_next ^= 1u;
    ++ctx->slots_busy;
    return PLAAC_OK;
}

// The counting pass of a two-pass run fed with text (round 5, late): parsed on the device like a scored batch, counted, not
// scored. Shares the two pending slots with the scoring calls; collected by plaac_histogram_end_text only.
plaac_status plaac_histogram_begin_text(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec) {
    return begin_text(ctx, text, text_len, starts, nrec, 1, false);
}
plaac_status plaac_histogram_end_text(plaac_ctx *ctx, int64_t counts[PLAAC_NAA], uint64_t *residues) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (!counts) return fail(ctx, PLAAC_ERR_ARG, "null counts");
    if (ctx->slots_busy == 0) return fail(ctx, PLAAC_ERR_ARG, "plaac_histogram_end_text: no batch is pending");
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_oldest];
    if (!S.hist_only) return fail(ctx, PLAAC_ERR_ARG, "plaac_histogram_end_text: the oldest batch was begun for scoring");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    S.busy = false;
    S.hist_only = false;
    ctx->slot_oldest ^= 1u;
    --ctx->slots_busy;
    for (int i = 0; i < NAA; ++i) counts[i] = 0;
    if (residues) *residues = S.nres;
    if (S.nprot == 0) return PLAAC_OK;
    PL_HIP(ctx, hipEventSynchronize(S.hist_ev));
    PL_HIP(ctx, hipMemcpyAsync(counts, S.d_counts, sizeof(int64_t) * NAA, hipMemcpyDeviceToHost, ctx->xfer));
    PL_HIP(ctx, hipStreamSynchronize(ctx->xfer));
    return PLAAC_OK;
}
plaac_status plaac_score_begin_counting(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot) {
    return score_begin(ctx, codes, offsets, nprot, true);
}

static plaac_status score_end(plaac_ctx *ctx, plaac_row *rows, int64_t *counts) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (ctx->slots_busy == 0) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end: no batch is pending");
    PL_HIP(ctx, hipSetDevice(ctx->device));
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_oldest];
    if (S.hist_only) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end: the oldest batch was begun for counting only (plaac_histogram_end_text)");
    S.busy = false; // (whatever happens below, the slot is given up)
    ctx->slot_oldest ^= 1u;
    --ctx->slots_busy;
    if (counts)
        for (int i = 0; i < NAA; ++i) counts[i] = 0;
    if (S.nprot == 0 || S.call_no == ~0ull) return PLAAC_OK;
    if (!rows) return fail(ctx, PLAAC_ERR_ARG, "null rows");
    if (counts && !S.counted) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_counts: the oldest batch was begun without counting");
    // (the histogram kernel was enqueued on the scoring stream ahead of the call whose join event this is)
    PL_HIP(ctx, hipEventSynchronize(ctx->ev[S.call_no % plaac_ctx::EV_SETS][10 /* E_JOIN */]));
    if (counts) {
        PL_HIP(ctx, hipMemcpyAsync(counts, S.d_counts, sizeof(int64_t) * NAA, hipMemcpyDeviceToHost, ctx->xfer));
        PL_HIP(ctx, hipStreamSynchronize(ctx->xfer));
    }
    return copy_out(ctx, rows, S.d_rows, sizeof(plaac_row) * (size_t)S.nprot, ctx->xfer);
}
plaac_status plaac_score_end(plaac_ctx *ctx, plaac_row *rows) { return score_end(ctx, rows, nullptr); }

plaac_status plaac_score_end_text(plaac_ctx *ctx, plaac_row *rows, uint8_t *codes, uint64_t codes_cap, uint64_t *offsets,
                                  uint8_t *blank_end, uint32_t *extents, int64_t *counts) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (ctx->slots_busy == 0) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text: no batch is pending");
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_oldest];
    if (!S.from_text) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text: the oldest batch was not begun from text");
    const uint32_t nrec = S.nprot;
    const uint64_t nres = S.nres;
    if (nrec && (!offsets || !blank_end)) {
        (void)score_end(ctx, rows, counts); // (the slot is given up whatever happens)
        return fail(ctx, PLAAC_ERR_ARG, "null output buffer");
    }
    if (codes && nres > codes_cap) {
        (void)score_end(ctx, rows, counts);
        return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text: the codes buffer is too small (the text's length always suffices)");
    }
    if (offsets && nrec == 0) offsets[0] = 0;
    plaac_status rc = score_end(ctx, rows, counts); // (waits for the batch; the slot's buffers stay until the next begin)
    if (rc != PLAAC_OK || nrec == 0) return rc;
    if (codes && nres && (rc = copy_out(ctx, codes, S.d_codes, (size_t)nres, ctx->xfer)) != PLAAC_OK) return rc;
    static_assert(sizeof(FaExtent) == 2 * sizeof(uint32_t), "extents are (h, b) pairs of uint32");
    if (extents && (rc = copy_out(ctx, extents, S.d_ext, sizeof(FaExtent) * (size_t)nrec, ctx->xfer)) != PLAAC_OK) return rc;
    if ((rc = copy_out(ctx, offsets, S.d_offsets, sizeof(uint64_t) * ((size_t)nrec + 1), ctx->xfer)) != PLAAC_OK) return rc;
    return copy_out(ctx, blank_end, S.d_blank, (size_t)nrec, ctx->xfer);
}
plaac_status plaac_score_end_counts(plaac_ctx *ctx, plaac_row *rows, int64_t counts[PLAAC_NAA]) {
    if (ctx && !counts) return fail(ctx, PLAAC_ERR_ARG, "null counts");
    return score_end(ctx, rows, counts);
}

// The oldest pending text batch's summary rows AS TEXT (round 5, format_device.hip.inc): _size waits for the batch, runs the
// length pass and says how many bytes the table takes and whether the host has to format this batch itself (|v| >= 1e9, an
// infinity the reference prints as such, a record without a sequence: then
// *needs_host != 0 and the caller collects the batch with plaac_score_end_text as before); _table writes the bytes, copies
// them to `table` (capacity >= the size reported) and gives the slot up like every other end call.
plaac_status plaac_score_end_text_table_size(plaac_ctx *ctx, int corelength, int ww2, int prev_blank, uint64_t *table_bytes,
                                             int *needs_host, int *last_blank, uint64_t *residues) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (!table_bytes || !needs_host || !last_blank) return fail(ctx, PLAAC_ERR_ARG, "null argument");
    if (ctx->slots_busy == 0) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text_table_size: no batch is pending");
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_oldest];
    if (!S.from_text || S.hist_only) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text_table_size: the oldest batch was not begun from text for scoring");
    *table_bytes = 0, *needs_host = 0, *last_blank = prev_blank;
    if (residues) *residues = S.nres;
    S.table_sized = false;
    const uint32_t nrec = S.nprot;
    if (nrec == 0 || S.call_no == ~0ull) {
        S.table_bytes = 0, S.table_sized = true;
        return PLAAC_OK;
    }
    PL_HIP(ctx, hipSetDevice(ctx->device));
    PL_HIP(ctx, hipEventSynchronize(ctx->ev[S.call_no % plaac_ctx::EV_SETS][10 /* E_JOIN */]));
    plaac_status rc;
    if ((rc = grow(ctx, S.d_toffs, S.cap_toffs, (size_t)nrec + 1)) != PLAAC_OK) return rc;
    if ((rc = grow(ctx, S.d_fmtflags, S.cap_fmtflags, (size_t)4)) != PLAAC_OK) return rc;
    PL_HIP(ctx, hipMemsetAsync(S.d_fmtflags, 0, 4 * sizeof(uint32_t), ctx->xfer));
    const unsigned nb = (nrec + 255u) / 256u, ns = (nrec + FA_SCAN - 1u) / FA_SCAN;
    hipLaunchKernelGGL(k_format_rows<false>, dim3(nb), dim3(256), 0, ctx->xfer, S.d_rows, S.d_codes, S.d_offsets, S.d_text, S.d_starts,
                       S.d_ext, S.d_blank, nrec, prev_blank, corelength, ww2, S.d_len, (const uint64_t *)nullptr, (char *)nullptr,
                       S.d_fmtflags);
    hipLaunchKernelGGL(k_fasta_block_sums, dim3(ns), dim3(FA_SCAN), 0, ctx->xfer, S.d_len, nrec, S.d_bsum);
    hipLaunchKernelGGL(k_fasta_offsets, dim3(ns), dim3(FA_SCAN), 0, ctx->xfer, S.d_len, nrec, S.d_bsum, S.d_toffs, S.d_total);
    struct {
        unsigned long long total;
        uint32_t flags[4];
        uint8_t lastb;
    } h{};
    PL_HIP(ctx, hipMemcpyAsync(&h.total, S.d_total, sizeof h.total, hipMemcpyDeviceToHost, ctx->xfer));
    PL_HIP(ctx, hipMemcpyAsync(h.flags, S.d_fmtflags, sizeof h.flags, hipMemcpyDeviceToHost, ctx->xfer));
    PL_HIP(ctx, hipMemcpyAsync(&h.lastb, S.d_blank + (nrec - 1), 1, hipMemcpyDeviceToHost, ctx->xfer));
    PL_HIP(ctx, hipStreamSynchronize(ctx->xfer));
    *table_bytes = h.total;
    *needs_host = (h.flags[0] || h.flags[1]) ? 1 : 0;
    *last_blank = h.lastb ? 1 : 0;
    S.table_bytes = h.total;
    S.table_sized = !*needs_host;
    ctx->table_corelength = corelength, ctx->table_ww2 = ww2, ctx->table_prev_blank = prev_blank;
    return PLAAC_OK;
}

plaac_status plaac_score_end_text_table(plaac_ctx *ctx, char *table, uint64_t table_cap, int64_t *counts) {
    if (!ctx) return PLAAC_ERR_ARG;
    if (ctx->slots_busy == 0) return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text_table: no batch is pending");
    plaac_ctx::Slot &S = ctx->slot[ctx->slot_oldest];
    if (!S.from_text || !S.table_sized)
        return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text_table: call plaac_score_end_text_table_size first (and plaac_score_end_text when it asks for the host)");
    if (S.table_bytes > table_cap || (S.table_bytes && !table))
        return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text_table: the buffer is smaller than the size reported");
    if (counts && S.nprot && S.call_no != ~0ull && !S.counted)
        return fail(ctx, PLAAC_ERR_ARG, "plaac_score_end_text_table: the batch was begun without counting");
    const uint32_t nrec = S.nprot;
    S.table_sized = false;
    if (nrec && S.call_no != ~0ull && S.table_bytes) {
        PL_HIP(ctx, hipSetDevice(ctx->device));
        plaac_status rc;
        if ((rc = grow(ctx, S.d_table, S.cap_table, (size_t)S.table_bytes + 16)) != PLAAC_OK) return rc;
        hipLaunchKernelGGL(k_format_rows<true>, dim3((nrec + 255u) / 256u), dim3(256), 0, ctx->xfer, S.d_rows, S.d_codes, S.d_offsets,
                           S.d_text, S.d_starts, S.d_ext, S.d_blank, nrec, ctx->table_prev_blank, ctx->table_corelength, ctx->table_ww2,
                           S.d_len, S.d_toffs, S.d_table, S.d_fmtflags);
        if ((rc = copy_out(ctx, table, S.d_table, (size_t)S.table_bytes, ctx->xfer)) != PLAAC_OK) return rc;
    }
    // the slot is given up; the counts come back as with plaac_score_end_counts (no rows are copied)
    S.busy = false;
    ctx->slot_oldest ^= 1u;
    --ctx->slots_busy;
    if (counts) {
        for (int i = 0; i < NAA; ++i) counts[i] = 0;
        if (nrec && S.call_no != ~0ull) {
            PL_HIP(ctx, hipMemcpyAsync(counts, S.d_counts, sizeof(int64_t) * NAA, hipMemcpyDeviceToHost, ctx->xfer));
            PL_HIP(ctx, hipStreamSynchronize(ctx->xfer));
        }
    }
    return PLAAC_OK;
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
