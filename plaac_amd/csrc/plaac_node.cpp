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

} // extern "C"
