// plaac_node.cpp — all GPUs of one node behind one handle (include/plaac_native.h, "node" section).
//
// The reference scores a proteome in one serial loop (cli/src/plaac.java:755 summary, :610 tracks). Proteins are
// independent given the parameter tables, so a node shards a batch BY SEQUENCE: one scoring context (plaac_ctx: its own
// device, streams and work buffers) per listed device, one host thread per context, no data-path collective. The only
// reduction is the 22 x int64 background histogram, summed on the host.
// ONE partitioner for every multi-GPU layer (plaac_shard_plan; plaac_amd/dist.py and bench.py call it through ctypes):
// records sorted by length, dealt to the shards in turn - SURVEY 8(e) G1's "sort by length, deal" - so that every shard
// gets the same residue count AND the same share of the long proteins whose serial chains bound a step. A shard's records
// are gathered into one contiguous batch (they are not contiguous in the input), its rows and tracks are scattered back
// to their places in the caller's arrays: input order is restored on the host.
// A plaac_node_batch keeps every shard resident on its device (upload once; histogram, scoring, the two-pass run and
// parameter sweeps reuse it). Pure host code on top of the single-device C ABI.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "plaac_native.h"

struct plaac_node_batch;
struct plaac_node {
    std::vector<plaac_ctx *> ctx;
    std::vector<int> device;
    std::string err;
    // the resident batches that are still alive: plaac_node_destroy releases their device memory and detaches them, so a
    // batch freed AFTER its node (Python's NodeBatch.__del__ behind Node.close(), JNI batchFree behind nodeDestroy) is no
    // use-after-free of the contexts
    std::mutex live_mu;
    std::vector<plaac_node_batch *> live;
    // text batches in flight (plaac_node_text_*): which context each went to, oldest first
    struct TextJob {
        int ctx;
        uint32_t nrec;
        bool hist;
        bool sized;
    };
    std::vector<TextJob> text_fifo;
    std::vector<int> text_busy;  // pending batches per context (at most two: the contexts' rule)
    int text_next = 0;           // round robin
    int text_prev_blank = 1;     // how the record before the next collected batch ended (1: a file's first batch)
    int text_last_blank = 1;     // ... as reported by the size call of the oldest batch
    std::vector<std::unique_ptr<std::mutex>> upload_mu; // one upload at a time per context
};

struct plaac_node_text_batch {
    plaac_node *node;
    int ctx;
    uint32_t nrec;
    plaac_text_batch *tb;
};

// a batch whose shards are resident on the devices of a node
struct plaac_node_batch {
    plaac_node *node = nullptr;
    uint32_t nprot = 0;
    uint64_t total = 0;
    struct Part {
        plaac_batch *b = nullptr;
        std::vector<uint32_t> idx;      // records of this shard, ascending input order
        std::vector<uint64_t> src_off;  // idx.size() + 1: offsets of those records in the CALLER'S residue numbering ...
        std::vector<uint64_t> offs;     // ... and in the shard's own (starting at 0)
    };
    std::vector<Part> part;
    bool identity = false; // one shard holding every record in input order: the caller's arrays are passed straight through
};

namespace {

thread_local std::string g_node_create_err;

// Runs body(k) for k = 0 .. parts-1, part 0 on the calling thread, the others on threads of their own. Nothing escapes:
// an exception inside a part (std::bad_alloc of its offsets copy ...) or from std::thread's constructor becomes a status
// in st[k], and every thread that was started is joined before this returns - the C ABI never unwinds into its caller
// (under the JNI shim that would take the JVM down instead of raising IllegalStateException).
template <class Body>
void run_parts(size_t parts, std::vector<plaac_status> &st, std::vector<std::string> &why, Body &&body) {
    auto guarded = [&](size_t k) {
        try {
            body(k);
        } catch (const std::bad_alloc &) {
            st[k] = PLAAC_ERR_NOMEM;
            why[k] = "out of host memory";
        } catch (const std::exception &e) {
            st[k] = PLAAC_ERR_DEVICE;
            why[k] = e.what();
        } catch (...) {
            st[k] = PLAAC_ERR_DEVICE;
            why[k] = "unknown exception";
        }
    };
    std::vector<std::thread> pool;
    size_t started = 1;
    try {
        pool.reserve(parts);
        for (; started < parts; ++started) pool.emplace_back(guarded, started);
    } catch (...) { // thread creation failed: the parts that have no thread run here, one after the other
        for (size_t k = started; k < parts; ++k) guarded(k);
    }
    guarded(0);
    for (auto &t : pool) t.join();
}

// first failing part -> node->err and its status
plaac_status parts_status(plaac_node *node, const std::vector<plaac_status> &st, const std::vector<std::string> &why) {
    for (size_t k = 0; k < st.size(); ++k)
        if (st[k] != PLAAC_OK) {
            node->err = std::string("device ") + std::to_string(node->device[k]) + ": " +
                        (why[k].empty() ? plaac_last_error(node->ctx[k]) : why[k].c_str());
            return st[k];
        }
    return PLAAC_OK;
}

plaac_status node_fail(plaac_node *node, plaac_status st, const char *msg) {
    if (node) node->err = msg;
    return st;
}

} // namespace

extern "C" {

plaac_status plaac_node_create(const plaac_params *params, const int *device_ids, int ndev, plaac_node **out) {
    if (!params || !out) {
        g_node_create_err = "plaac_node_create: null argument";
        return PLAAC_ERR_ARG;
    }
    *out = nullptr;
    std::vector<int> devs;
    try {
        if (device_ids && ndev > 0) {
            devs.assign(device_ids, device_ids + ndev);
        } else {
            const int n = plaac_device_count();
            if (n <= 0) {
                g_node_create_err = "no HIP device available";
                return PLAAC_ERR_DEVICE;
            }
            for (int d = 0; d < n; ++d) devs.push_back(d);
        }
    } catch (...) {
        g_node_create_err = "out of host memory";
        return PLAAC_ERR_NOMEM;
    }
    plaac_node *node = new (std::nothrow) plaac_node();
    if (!node) {
        g_node_create_err = "out of host memory";
        return PLAAC_ERR_NOMEM;
    }
    // contexts are created side by side: HIP start-up and the per-context allocations dominate, not the tables
    std::vector<plaac_status> st;
    std::vector<std::string> msg;
    try {
        node->device = devs;
        node->ctx.assign(devs.size(), nullptr);
        st.assign(devs.size(), PLAAC_OK);
        msg.assign(devs.size(), std::string());
    } catch (...) {
        g_node_create_err = "out of host memory";
        delete node;
        return PLAAC_ERR_NOMEM;
    }
    run_parts(devs.size(), st, msg, [&](size_t k) {
        st[k] = plaac_ctx_create(params, devs[k], &node->ctx[k]);
        if (st[k] != PLAAC_OK) msg[k] = plaac_last_error(nullptr); // thread-local message of this thread
    });
    for (size_t k = 0; k < devs.size(); ++k)
        if (st[k] != PLAAC_OK) {
            g_node_create_err = "device " + std::to_string(devs[k]) + ": " + msg[k];
            const plaac_status bad = st[k];
            plaac_node_destroy(node);
            return bad;
        }
    try {
        node->text_busy.assign(devs.size(), 0);
        for (size_t k = 0; k < devs.size(); ++k) node->upload_mu.emplace_back(new std::mutex());
    } catch (...) {
        g_node_create_err = "out of host memory";
        plaac_node_destroy(node);
        return PLAAC_ERR_NOMEM;
    }
    *out = node;
    return PLAAC_OK;
}

static void release_parts(plaac_node_batch *nb);

void plaac_node_destroy(plaac_node *node) {
    if (!node) return;
    {
        std::lock_guard<std::mutex> lock(node->live_mu);
        for (plaac_node_batch *nb : node->live) release_parts(nb); // (the shells stay valid until plaac_node_batch_free)
        node->live.clear();
    }
    // (side by side, as they were created: a context's streams and buffers take tens of ms to hand back)
    std::vector<plaac_status> st(node->ctx.size(), PLAAC_OK);
    std::vector<std::string> why(node->ctx.size());
    if (!node->ctx.empty())
        run_parts(node->ctx.size(), st, why, [&](size_t k) {
            if (node->ctx[k]) plaac_ctx_destroy(node->ctx[k]);
        });
    delete node;
}

int plaac_node_size(const plaac_node *node) { return node ? (int)node->ctx.size() : 0; }

plaac_ctx *plaac_node_ctx(plaac_node *node, int k) {
    return (node && k >= 0 && (size_t)k < node->ctx.size()) ? node->ctx[(size_t)k] : nullptr;
}

const char *plaac_node_last_error(const plaac_node *node) {
    return node ? node->err.c_str() : g_node_create_err.c_str();
}

plaac_status plaac_node_set_params(plaac_node *node, const plaac_params *params) {
    if (!node || !params) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_set_params: null argument");
    for (size_t k = 0; k < node->ctx.size(); ++k) {
        const plaac_status st = plaac_ctx_set_params(node->ctx[k], params);
        if (st != PLAAC_OK) {
            node->err = std::string("device ") + std::to_string(node->device[k]) + ": " + plaac_last_error(node->ctx[k]);
            return st;
        }
    }
    return PLAAC_OK;
}

plaac_status plaac_node_set_overlap(plaac_node *node, int on) {
    if (!node) return PLAAC_ERR_ARG;
    for (size_t k = 0; k < node->ctx.size(); ++k) {
        const plaac_status st = plaac_ctx_set_overlap(node->ctx[k], on);
        if (st != PLAAC_OK) return node_fail(node, st, "plaac_ctx_set_overlap failed");
    }
    return PLAAC_OK;
}

// ---- the one partitioner -------------------------------------------------------------------------------------------
plaac_status plaac_shard_plan(const uint64_t *offsets, uint32_t nprot, uint32_t parts, uint32_t *index_out,
                              uint32_t *part_start) {
    if (!part_start || parts == 0 || (nprot && (!offsets || !index_out))) return PLAAC_ERR_ARG;
    try {
        // stable sort by descending length: LSD radix sort on ~length (four 8-bit passes, stable by construction)
        std::vector<uint32_t> key(nprot), idx(nprot), key2(nprot), idx2(nprot);
        for (uint32_t i = 0; i < nprot; ++i) {
            if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xfffffffeull) return PLAAC_ERR_ARG;
            key[i] = ~(uint32_t)(offsets[i + 1] - offsets[i]);
            idx[i] = i;
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 8 * pass;
            size_t cnt[257] = {0};
            for (uint32_t i = 0; i < nprot; ++i) ++cnt[((key[i] >> sh) & 255u) + 1];
            bool trivial = false;
            for (int b = 0; b < 256; ++b) {
                if (cnt[b + 1] == nprot) trivial = true; // every key has the same digit: nothing moves
                cnt[b + 1] += cnt[b];
            }
            if (trivial) continue;
            for (uint32_t i = 0; i < nprot; ++i) {
                const size_t d = cnt[(key[i] >> sh) & 255u]++;
                key2[d] = key[i];
                idx2[d] = idx[i];
            }
            key.swap(key2);
            idx.swap(idx2);
        }
        // deal: position pos of the sorted order goes to shard (lap even ? col : parts - 1 - col), lap = pos / parts
        std::vector<uint32_t> &owner = key; // (reused)
        std::vector<size_t> count(parts, 0);
        for (uint32_t pos = 0; pos < nprot; ++pos) {
            const uint32_t lap = pos / parts, col = pos % parts;
            const uint32_t o = (lap & 1u) ? parts - 1u - col : col;
            owner[idx[pos]] = o;
            ++count[o];
        }
        part_start[0] = 0;
        for (uint32_t k = 0; k < parts; ++k) part_start[k + 1] = part_start[k] + (uint32_t)count[k];
        std::vector<uint32_t> fill(part_start, part_start + parts);
        for (uint32_t i = 0; i < nprot; ++i) index_out[fill[owner[i]]++] = i; // ascending input order inside a shard
    } catch (...) {
        return PLAAC_ERR_NOMEM;
    }
    return PLAAC_OK;
}

// ---- resident batches ----------------------------------------------------------------------------------------------
plaac_status plaac_node_batch_upload(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                     plaac_node_batch **out) {
    if (!node || !out || (nprot && !offsets)) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_batch_upload: null argument");
    *out = nullptr;
    // (as the single-device entry points: rows and tracks are indexed from the first residue of the batch)
    if (nprot && offsets[0] != 0) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_batch_upload: offsets[0] must be 0");
    if (nprot && offsets[nprot] && !codes) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_batch_upload: null codes");
    const size_t parts = node->ctx.size();
    plaac_node_batch *nb = nullptr;
    std::vector<uint32_t> index, start;
    std::vector<plaac_status> st;
    std::vector<std::string> why;
    try {
        nb = new plaac_node_batch();
        nb->node = node;
        nb->nprot = nprot;
        nb->total = nprot ? offsets[nprot] : 0;
        nb->part.resize(parts);
        nb->identity = parts == 1;
        index.resize(nprot);
        start.resize(parts + 1);
        st.assign(parts, PLAAC_OK);
        why.assign(parts, std::string());
    } catch (...) {
        delete nb;
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    const plaac_status ps = plaac_shard_plan(offsets, nprot, (uint32_t)parts, index.data(), start.data());
    if (ps != PLAAC_OK) {
        delete nb;
        return node_fail(node, ps, ps == PLAAC_ERR_NOMEM ? "out of host memory" : "offsets must be non-decreasing, records below 2^32 residues");
    }
    run_parts(parts, st, why, [&](size_t k) {
        plaac_node_batch::Part &P = nb->part[k];
        const uint32_t n = start[k + 1] - start[k];
        P.idx.assign(index.begin() + start[k], index.begin() + start[k + 1]);
        P.src_off.resize((size_t)n + 1);
        P.offs.resize((size_t)n + 1);
        P.offs[0] = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t r = P.idx[i];
            P.src_off[i] = offsets[r];
            P.offs[i + 1] = P.offs[i] + (offsets[r + 1] - offsets[r]);
        }
        P.src_off[n] = 0;
        if (n == 0) return;
        if (nb->identity) { // one device: the shard IS the batch, no host copy of it
            st[k] = plaac_batch_upload(node->ctx[k], codes, offsets, n, &P.b);
            return;
        }
        // the shard's records as one contiguous batch (runs of neighbouring records are copied in one piece)
        std::vector<uint8_t> shard((size_t)P.offs[n]);
        for (uint32_t i = 0; i < n;) {
            uint32_t j = i + 1;
            while (j < n && P.idx[j] == P.idx[j - 1] + 1u) ++j;
            std::memcpy(shard.data() + P.offs[i], codes + offsets[P.idx[i]], (size_t)(P.offs[j] - P.offs[i]));
            i = j;
        }
        st[k] = plaac_batch_upload(node->ctx[k], shard.data(), P.offs.data(), n, &P.b);
    });
    const plaac_status bad = parts_status(node, st, why);
    if (bad != PLAAC_OK) {
        plaac_node_batch_free(nb);
        return bad;
    }
    try {
        std::lock_guard<std::mutex> lock(node->live_mu);
        node->live.push_back(nb);
    } catch (...) {
        plaac_node_batch_free(nb);
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    *out = nb;
    return PLAAC_OK;
}

const char *plaac_node_batch_last_error(const plaac_node_batch *nb) {
    if (!nb) return "null batch handle";
    return nb->node ? nb->node->err.c_str() : "the batch's node has been destroyed (the batch is detached: free it)";
}

uint32_t plaac_node_batch_records(const plaac_node_batch *nb) { return nb ? nb->nprot : 0u; }
uint64_t plaac_node_batch_residues(const plaac_node_batch *nb) { return nb ? nb->total : 0ull; }

// the device side of a batch (its contexts must still exist)
static void release_parts(plaac_node_batch *nb) {
    for (auto &P : nb->part) {
        if (P.b) plaac_batch_free(P.b);
        P.b = nullptr;
    }
    nb->node = nullptr;
}

void plaac_node_batch_free(plaac_node_batch *nb) {
    if (!nb) return;
    if (plaac_node *node = nb->node) { // (null: the node went first and has released the device side already)
        std::lock_guard<std::mutex> lock(node->live_mu);
        node->live.erase(std::remove(node->live.begin(), node->live.end(), nb), node->live.end());
        release_parts(nb);
    }
    delete nb;
}

plaac_status plaac_node_batch_histogram(plaac_node_batch *nb, int64_t counts[PLAAC_NAA]) {
    if (!nb || !nb->node) return PLAAC_ERR_ARG;
    plaac_node *node = nb->node;
    if (!counts) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_batch_histogram: null counts");
    for (int i = 0; i < PLAAC_NAA; ++i) counts[i] = 0;
    const size_t parts = nb->part.size();
    std::vector<int64_t> part;
    std::vector<plaac_status> st;
    std::vector<std::string> why;
    try {
        part.assign(parts * PLAAC_NAA, 0);
        st.assign(parts, PLAAC_OK);
        why.assign(parts, std::string());
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    run_parts(parts, st, why, [&](size_t k) {
        if (nb->part[k].b) st[k] = plaac_batch_histogram(nb->part[k].b, part.data() + k * PLAAC_NAA);
    });
    const plaac_status bad = parts_status(node, st, why);
    if (bad != PLAAC_OK) return bad;
    for (size_t k = 0; k < parts; ++k) // the one reduction of the path: 22 x int64 per device
        for (int i = 0; i < PLAAC_NAA; ++i) counts[i] += part[k * PLAAC_NAA + i];
    return PLAAC_OK;
}

// rows[point][record]: every shard scores its resident records (npoints == 0: the contexts' current parameters, with
// tracks when asked for) and puts its rows / track slices back to their places in the caller's arrays
static plaac_status node_batch_run(plaac_node_batch *nb, const plaac_params *points, uint32_t npoints, plaac_row *const *rows,
                                   const plaac_tracks *tracks) {
    plaac_node *node = nb->node;
    const size_t parts = nb->part.size();
    const uint32_t nout = npoints ? npoints : 1u;
    std::vector<plaac_status> st;
    std::vector<std::string> why;
    try {
        st.assign(parts, PLAAC_OK);
        why.assign(parts, std::string());
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    run_parts(parts, st, why, [&](size_t k) {
        const plaac_node_batch::Part &P = nb->part[k];
        const size_t n = P.idx.size();
        if (!P.b || n == 0) return;
        if (nb->identity) { // one device, records in input order: straight into the caller's arrays, no second copy
            st[k] = npoints ? plaac_batch_sweep(P.b, points, npoints, rows) : plaac_batch_score(P.b, rows[0], tracks);
            return;
        }
        // (uninitialised: every row is written by the device copy; a 9-point sweep over a 1.25 M-record shard is 1.8 GB)
        std::unique_ptr<plaac_row[]> tmp(new plaac_row[n * nout]);
        std::vector<plaac_row *> tp(nout);
        for (uint32_t i = 0; i < nout; ++i) tp[i] = tmp.get() + (size_t)i * n;
        std::unique_ptr<uint8_t[]> t8;
        std::unique_ptr<double[]> t64;
        plaac_tracks tt{};
        const uint64_t total = P.offs[n];
        if (tracks) { // the shard's tracks in its own residue numbering, scattered record by record afterwards
            t8.reset(new uint8_t[2 * (size_t)total + 2]);
            t64.reset(new double[10 * (size_t)total + 10]);
            double *b = t64.get();
            tt = plaac_tracks{t8.get(), t8.get() + total, b,           b + total,     b + 2 * total, b + 3 * total,
                              b + 4 * total, b + 5 * total, b + 6 * total, b + 7 * total, b + 8 * total, b + 9 * total};
        }
        st[k] = npoints ? plaac_batch_sweep(P.b, points, npoints, tp.data()) : plaac_batch_score(P.b, tp[0], tracks ? &tt : nullptr);
        if (st[k] != PLAAC_OK) return;
        for (uint32_t i = 0; i < nout; ++i)
            for (size_t r = 0; r < n; ++r) rows[i][P.idx[r]] = tp[i][r];
        if (tracks) {
            uint8_t *const d8[2] = {tracks->vit, tracks->map};
            const uint8_t *const s8[2] = {tt.vit, tt.map};
            double *const dd[10] = {tracks->charge, tracks->hydro,      tracks->fi,     tracks->plaacllr, tracks->papa,
                                    tracks->fix2,   tracks->plaacllrx2, tracks->papax2, tracks->post0,    tracks->post1};
            const double *const sd[10] = {tt.charge, tt.hydro, tt.fi, tt.plaacllr, tt.papa, tt.fix2, tt.plaacllrx2, tt.papax2,
                                          tt.post0,  tt.post1};
            for (size_t r = 0; r < n; ++r) {
                const uint64_t so = P.offs[r], len = P.offs[r + 1] - so, dof = P.src_off[r];
                for (int a = 0; a < 2; ++a) std::memcpy(d8[a] + dof, s8[a] + so, (size_t)len);
                for (int a = 0; a < 10; ++a) std::memcpy(dd[a] + dof, sd[a] + so, (size_t)len * sizeof(double));
            }
        }
    });
    return parts_status(node, st, why);
}

plaac_status plaac_node_batch_score(plaac_node_batch *nb, plaac_row *rows, const plaac_tracks *tracks) {
    if (!nb || !nb->node) return PLAAC_ERR_ARG;
    if (nb->nprot == 0) return PLAAC_OK;
    if (!rows) return node_fail(nb->node, PLAAC_ERR_ARG, "plaac_node_batch_score: null rows");
    if (tracks) {
        const void *all[] = {tracks->vit,  tracks->map,  tracks->charge,     tracks->hydro,  tracks->fi,    tracks->plaacllr,
                             tracks->papa, tracks->fix2, tracks->plaacllrx2, tracks->papax2, tracks->post0, tracks->post1};
        for (const void *q : all)
            if (!q) return node_fail(nb->node, PLAAC_ERR_ARG, "tracks struct has a null array");
    }
    plaac_row *one[1] = {rows};
    return node_batch_run(nb, nullptr, 0, one, tracks);
}

plaac_status plaac_node_batch_sweep(plaac_node_batch *nb, const plaac_params *points, uint32_t npoints,
                                    plaac_row *const *rows) {
    if (!nb || !nb->node) return PLAAC_ERR_ARG;
    if (nb->nprot == 0 || npoints == 0) return PLAAC_OK;
    if (!points || !rows) return node_fail(nb->node, PLAAC_ERR_ARG, "plaac_node_batch_sweep: null argument");
    for (uint32_t i = 0; i < npoints; ++i)
        if (!rows[i]) return node_fail(nb->node, PLAAC_ERR_ARG, "plaac_node_batch_sweep: null row array");
    return node_batch_run(nb, points, npoints, rows, nullptr);
}

// ---- one-shot forms: upload, use, free -----------------------------------------------------------------------------
plaac_status plaac_node_histogram(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                  int64_t counts[PLAAC_NAA]) {
    if (!node || !counts || (nprot && !offsets)) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_histogram: null argument");
    for (int i = 0; i < PLAAC_NAA; ++i) counts[i] = 0;
    if (nprot == 0) return PLAAC_OK;
    plaac_node_batch *nb = nullptr;
    plaac_status st = plaac_node_batch_upload(node, codes, offsets, nprot, &nb);
    if (st != PLAAC_OK) return st;
    st = plaac_node_batch_histogram(nb, counts);
    plaac_node_batch_free(nb);
    return st;
}

plaac_status plaac_node_score(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                              plaac_row *rows, const plaac_tracks *tracks) {
    if (!node || (nprot && (!offsets || !rows))) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_score: null argument");
    if (nprot == 0) return PLAAC_OK;
    plaac_node_batch *nb = nullptr;
    plaac_status st = plaac_node_batch_upload(node, codes, offsets, nprot, &nb);
    if (st != PLAAC_OK) return st;
    st = plaac_node_batch_score(nb, rows, tracks);
    plaac_node_batch_free(nb);
    return st;
}


// ---- FASTA text through a node (round 6): the contexts' text entry points, dealt round robin, collected in order -----------
namespace {
plaac_status ctx_fail(plaac_node *node, int k, plaac_status st) {
    node->err = std::string("device ") + std::to_string(node->device[(size_t)k]) + ": " + plaac_last_error(node->ctx[(size_t)k]);
    return st;
}
// the context the next batch goes to: round robin over those with a free slot
int text_pick(plaac_node *node) {
    const int n = (int)node->ctx.size();
    for (int i = 0; i < n; ++i) {
        const int k = (node->text_next + i) % n;
        if (node->text_busy[(size_t)k] < 2) {
            node->text_next = (k + 1) % n;
            return k;
        }
    }
    return -1;
}
plaac_status text_push(plaac_node *node, int k, uint32_t nrec, bool hist) {
    try {
        node->text_fifo.push_back(plaac_node::TextJob{k, nrec, hist, false});
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    ++node->text_busy[(size_t)k];
    return PLAAC_OK;
}
void text_pop(plaac_node *node) {
    --node->text_busy[(size_t)node->text_fifo.front().ctx];
    node->text_fifo.erase(node->text_fifo.begin());
}
} // namespace

plaac_status plaac_node_text_begin(plaac_node *node, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                                   int counting) {
    if (!node) return PLAAC_ERR_ARG;
    const int k = text_pick(node);
    if (k < 0) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_begin: every context holds two pending batches (collect the oldest first)");
    try {
        node->text_fifo.reserve(node->text_fifo.size() + 1);
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    const plaac_status st = plaac_score_begin_text(node->ctx[(size_t)k], text, text_len, starts, nrec, counting);
    if (st != PLAAC_OK) return ctx_fail(node, k, st);
    return text_push(node, k, nrec, false);
}

plaac_status plaac_node_text_upload(plaac_node *node, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                                    plaac_node_text_batch **out) {
    if (!node || !out) return PLAAC_ERR_ARG;
    *out = nullptr;
    // (may run on another host thread than the scoring calls: touches nothing of the node but the per-context upload lock and
    //  an atomic-free round robin of its own - the context is chosen by the batch's turn, which only this thread advances)
    static thread_local int turn = 0;
    const int n = (int)node->ctx.size();
    if (n == 0) return PLAAC_ERR_ARG;
    const int k = turn++ % n;
    plaac_node_text_batch *b = new (std::nothrow) plaac_node_text_batch{node, k, nrec, nullptr};
    if (!b) return PLAAC_ERR_NOMEM;
    plaac_status st;
    {
        std::lock_guard<std::mutex> lock(*node->upload_mu[(size_t)k]);
        st = plaac_text_upload(node->ctx[(size_t)k], text, text_len, starts, nrec, &b->tb);
    }
    if (st != PLAAC_OK) {
        delete b;
        return st; // (the message: plaac_text_upload_error(plaac_node_ctx(node, k)); node->err belongs to the scoring thread)
    }
    *out = b;
    return PLAAC_OK;
}

plaac_status plaac_node_text_begin_uploaded(plaac_node *node, plaac_node_text_batch *batch, int counting) {
    if (!node || !batch || batch->node != node) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_begin_uploaded: not a batch of this node");
    const int k = batch->ctx;
    if (node->text_busy[(size_t)k] >= 2)
        return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_begin_uploaded: the batch's context holds two pending batches (collect the oldest first)");
    try {
        node->text_fifo.reserve(node->text_fifo.size() + 1);
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    const plaac_status st = plaac_score_begin_uploaded(node->ctx[(size_t)k], batch->tb, counting);
    if (st != PLAAC_OK) return ctx_fail(node, k, st); // (the batch is still the caller's)
    const uint32_t nrec = batch->nrec;
    delete batch; // (the context consumed the upload)
    return text_push(node, k, nrec, false);
}

void plaac_node_text_batch_free(plaac_node_text_batch *batch) {
    if (!batch) return;
    if (batch->tb) plaac_text_batch_free(batch->tb);
    delete batch;
}

plaac_status plaac_node_text_table_size(plaac_node *node, int corelength, int ww2, uint64_t *table_bytes, int *needs_host,
                                        uint64_t *residues) {
    if (!node) return PLAAC_ERR_ARG;
    if (!table_bytes || !needs_host) return node_fail(node, PLAAC_ERR_ARG, "null argument");
    if (node->text_fifo.empty()) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_table_size: no batch is pending");
    plaac_node::TextJob &J = node->text_fifo.front();
    if (J.hist) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_table_size: the oldest batch was begun for counting only");
    int lastb = node->text_prev_blank;
    const plaac_status st = plaac_score_end_text_table_size(node->ctx[(size_t)J.ctx], corelength, ww2, node->text_prev_blank, table_bytes,
                                                            needs_host, &lastb, residues);
    if (st != PLAAC_OK) return ctx_fail(node, J.ctx, st);
    node->text_last_blank = lastb;
    J.sized = true;
    return PLAAC_OK;
}

plaac_status plaac_node_text_table(plaac_node *node, char *table, uint64_t table_cap, int64_t *counts) {
    if (!node) return PLAAC_ERR_ARG;
    if (node->text_fifo.empty()) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_table: no batch is pending");
    const plaac_node::TextJob J = node->text_fifo.front();
    if (!J.sized) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_table: call plaac_node_text_table_size first");
    const plaac_status st = plaac_score_end_text_table(node->ctx[(size_t)J.ctx], table, table_cap, counts);
    if (st != PLAAC_OK) return ctx_fail(node, J.ctx, st); // (the context keeps the batch: a larger buffer, or plaac_node_text_rows)
    node->text_prev_blank = node->text_last_blank;
    text_pop(node);
    return PLAAC_OK;
}

plaac_status plaac_node_text_rows(plaac_node *node, plaac_row *rows, uint8_t *codes, uint64_t codes_cap, uint64_t *offsets,
                                  uint8_t *blank_end, uint32_t *extents, int64_t *counts) {
    if (!node) return PLAAC_ERR_ARG;
    if (node->text_fifo.empty()) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_rows: no batch is pending");
    const plaac_node::TextJob J = node->text_fifo.front();
    if (J.hist) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_rows: the oldest batch was begun for counting only");
    const plaac_status st = plaac_score_end_text(node->ctx[(size_t)J.ctx], rows, codes, codes_cap, offsets, blank_end, extents, counts);
    text_pop(node); // (plaac_score_end_text gives the slot up whatever happens)
    if (st != PLAAC_OK) return ctx_fail(node, J.ctx, st);
    if (J.nrec && blank_end) node->text_prev_blank = blank_end[J.nrec - 1] ? 1 : 0;
    return PLAAC_OK;
}

plaac_status plaac_node_text_discard(plaac_node *node) {
    if (!node) return PLAAC_ERR_ARG;
    if (node->text_fifo.empty()) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_text_discard: no batch is pending");
    const plaac_node::TextJob J = node->text_fifo.front();
    plaac_status st;
    if (J.hist) {
        int64_t counts[PLAAC_NAA];
        st = plaac_histogram_end_text(node->ctx[(size_t)J.ctx], counts, nullptr);
    } else {
        // how the batch's last record ended is still owed to the batch behind it
        std::vector<plaac_row> rows;
        std::vector<uint64_t> offs;
        std::vector<uint8_t> blank;
        try {
            rows.resize(J.nrec);
            offs.resize((size_t)J.nrec + 1);
            blank.resize(J.nrec);
        } catch (...) {
            return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
        }
        st = plaac_score_end_text(node->ctx[(size_t)J.ctx], rows.data(), nullptr, 0, offs.data(), blank.data(), nullptr, nullptr);
        if (st == PLAAC_OK && J.nrec) node->text_prev_blank = blank[J.nrec - 1] ? 1 : 0;
    }
    text_pop(node);
    if (st != PLAAC_OK) return ctx_fail(node, J.ctx, st);
    return PLAAC_OK;
}

int plaac_node_text_pending(const plaac_node *node) { return node ? (int)node->text_fifo.size() : 0; }
uint32_t plaac_node_text_oldest_records(const plaac_node *node) {
    return (node && !node->text_fifo.empty()) ? node->text_fifo.front().nrec : 0u;
}
void plaac_node_text_reset(plaac_node *node) {
    if (node) node->text_prev_blank = node->text_last_blank = 1;
}

plaac_status plaac_node_histogram_text_begin(plaac_node *node, const char *text, uint64_t text_len, const uint64_t *starts,
                                             uint32_t nrec) {
    if (!node) return PLAAC_ERR_ARG;
    const int k = text_pick(node);
    if (k < 0) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_histogram_text_begin: every context holds two pending batches");
    try {
        node->text_fifo.reserve(node->text_fifo.size() + 1);
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    const plaac_status st = plaac_histogram_begin_text(node->ctx[(size_t)k], text, text_len, starts, nrec);
    if (st != PLAAC_OK) return ctx_fail(node, k, st);
    return text_push(node, k, nrec, true);
}

plaac_status plaac_node_histogram_text_end(plaac_node *node, int64_t counts[PLAAC_NAA], uint64_t *residues) {
    if (!node) return PLAAC_ERR_ARG;
    if (!counts) return node_fail(node, PLAAC_ERR_ARG, "null counts");
    if (node->text_fifo.empty()) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_histogram_text_end: no batch is pending");
    const plaac_node::TextJob J = node->text_fifo.front();
    if (!J.hist) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_histogram_text_end: the oldest batch was begun for scoring");
    int64_t c[PLAAC_NAA];
    uint64_t r = 0;
    const plaac_status st = plaac_histogram_end_text(node->ctx[(size_t)J.ctx], c, &r);
    text_pop(node);
    if (st != PLAAC_OK) return ctx_fail(node, J.ctx, st);
    for (int i = 0; i < PLAAC_NAA; ++i) counts[i] += c[i];
    if (residues) *residues += r;
    return PLAAC_OK;
}

plaac_status plaac_node_score_tracks_table(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                           const char *labels, const uint64_t *label_off, plaac_row *rows, char **table,
                                           uint64_t *table_len, int *needs_host) {
    if (!node) return PLAAC_ERR_ARG;
    if (!node->text_fifo.empty()) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_score_tracks_table: text batches are pending");
    const int n = (int)node->ctx.size();
    if (n == 0) return PLAAC_ERR_ARG;
    const int k = node->text_next % n;
    node->text_next = (k + 1) % n;
    const plaac_status st = plaac_score_tracks_table(node->ctx[(size_t)k], codes, offsets, nprot, labels, label_off, rows, table, table_len,
                                                     needs_host);
    if (st != PLAAC_OK) return ctx_fail(node, k, st);
    return PLAAC_OK;
}

} // extern "C"
