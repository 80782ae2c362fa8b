// plaac_node.cpp — all GPUs of one node behind one handle (include/plaac_native.h, "node" section).
//
// The reference scores a proteome in one serial loop (cli/src/plaac.java:755 summary, :610 tracks). Proteins are
// independent given the parameter tables, so a node shards a batch BY SEQUENCE: one scoring context (plaac_ctx: its own
// device, streams and work buffers) per listed device, one host thread per context, contiguous ranges of records with
// about equal residue counts, rows written straight to their place in the caller's array (input order is restored
// for free), no data-path collective. The only reduction is the 22 x int64 background histogram, summed on the host.
// Pure host code on top of the single-device C ABI.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "plaac_native.h"

struct plaac_node {
    std::vector<plaac_ctx *> ctx;
    std::vector<int> device;
    std::string err;
};

namespace {

thread_local std::string g_node_create_err;

// record ranges [cut[k], cut[k+1]) of about equal residue counts
std::vector<uint32_t> balanced_cuts(const uint64_t *offsets, uint32_t nprot, size_t parts) {
    std::vector<uint32_t> cut(parts + 1, nprot);
    cut[0] = 0;
    const uint64_t total = offsets[nprot] - offsets[0];
    for (size_t k = 1; k < parts; ++k) {
        const uint64_t target = offsets[0] + total / parts * k;
        const uint64_t *it = std::lower_bound(offsets, offsets + nprot + 1, target);
        uint32_t r = (uint32_t)(it - offsets);
        if (r > nprot) r = nprot;
        cut[k] = std::max(r, cut[k - 1]);
    }
    return cut;
}

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

template <class Fn>
plaac_status for_each_part(plaac_node *node, const uint64_t *offsets, uint32_t nprot, Fn &&fn) {
    const size_t parts = node->ctx.size();
    std::vector<uint32_t> cut;
    std::vector<plaac_status> st;
    std::vector<std::string> why;
    try {
        cut = balanced_cuts(offsets, nprot, parts);
        st.assign(parts, PLAAC_OK);
        why.assign(parts, std::string());
    } catch (...) {
        node->err = "out of host memory";
        return PLAAC_ERR_NOMEM;
    }
    run_parts(parts, st, why, [&](size_t k) {
        const uint32_t a = cut[k], b = cut[k + 1];
        if (a == b) return;
        std::vector<uint64_t> offs((size_t)(b - a) + 1); // the shard's own offsets, starting at 0
        for (uint32_t i = a; i <= b; ++i) offs[i - a] = offsets[i] - offsets[a];
        st[k] = fn(k, a, b, offs.data());
    });
    for (size_t k = 0; k < parts; ++k)
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

void plaac_node_destroy(plaac_node *node) {
    if (!node) return;
    for (plaac_ctx *c : node->ctx)
        if (c) plaac_ctx_destroy(c);
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

plaac_status plaac_node_histogram(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                  int64_t counts[PLAAC_NAA]) {
    if (!node || !counts || (nprot && !offsets)) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_histogram: null argument");
    for (int i = 0; i < PLAAC_NAA; ++i) counts[i] = 0;
    if (nprot == 0) return PLAAC_OK;
    std::vector<int64_t> part;
    try {
        part.assign(node->ctx.size() * PLAAC_NAA, 0);
    } catch (...) {
        return node_fail(node, PLAAC_ERR_NOMEM, "out of host memory");
    }
    const plaac_status st = for_each_part(node, offsets, nprot, [&](size_t k, uint32_t a, uint32_t b, const uint64_t *offs) {
        return plaac_histogram(node->ctx[k], codes + offsets[a], offs, b - a, part.data() + k * PLAAC_NAA);
    });
    if (st != PLAAC_OK) return st;
    for (size_t k = 0; k < node->ctx.size(); ++k) // the one reduction of the path: 22 x int64 per device
        for (int i = 0; i < PLAAC_NAA; ++i) counts[i] += part[k * PLAAC_NAA + i];
    return PLAAC_OK;
}

plaac_status plaac_node_score(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                              plaac_row *rows, const plaac_tracks *tracks) {
    if (!node || (nprot && (!offsets || !rows))) return node_fail(node, PLAAC_ERR_ARG, "plaac_node_score: null argument");
    if (nprot == 0) return PLAAC_OK;
    return for_each_part(node, offsets, nprot, [&](size_t k, uint32_t a, uint32_t b, const uint64_t *offs) {
        plaac_tracks t, *tp = nullptr;
        if (tracks) { // per-residue arrays are indexed like `codes`: the shard's slice starts at its first residue
            const uint64_t o = offsets[a];
            t = plaac_tracks{tracks->vit + o,   tracks->map + o,  tracks->charge + o,     tracks->hydro + o,
                             tracks->fi + o,    tracks->plaacllr + o, tracks->papa + o,   tracks->fix2 + o,
                             tracks->plaacllrx2 + o, tracks->papax2 + o, tracks->post0 + o, tracks->post1 + o};
            tp = &t;
        }
        return plaac_score(node->ctx[k], codes + offsets[a], offs, b - a, rows + a, tp);
    });
}

} // extern "C"
