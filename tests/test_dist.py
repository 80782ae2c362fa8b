"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: shard by sequence, no data-path collective,
histogram all-reduce, final gather of 160-byte rows restored to input order. The scorer is the oracle
(tests may use it) standing in for the per-rank HIP context."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import oracle_ctypes as oc
    from plaac_amd import dist as pdist
    from plaac_amd import synth
    r, _, w = pdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    P = oc.build_params()
    codes, offs = synth.make_batch(2, nprot=301, seed=21, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.2)
    plan = pdist.shard_plan(offs, world)
    c_s, o_s = pdist.extract_shard(codes, offs, plan[rank])
    # exchange (i): background histogram
    counts = pdist.allreduce_counts(oc.histogram(c_s, o_s))
    assert np.array_equal(counts, oc.histogram(codes, offs))
    # data path: no collective
    P2 = oc.build_params(alpha=0.5, bgcounts=counts.astype(np.float64))
    rows_s = oc.score_batch(P2, c_s, o_s)
    # exchange (ii): gather rows to rank 0, input order
    out = pdist.gather_rows(rows_s, plan[rank], len(offs) - 1)
    # ... and as 136-byte wire rows in blocks of their exact sizes, with and without the indices known to rank 0
    out_w = pdist.gather_rows(rows_s, plan[rank], len(offs) - 1, offsets=offs, corelength=P2.corelength)
    out_p = pdist.gather_rows(rows_s, plan[rank], len(offs) - 1, offsets=offs, corelength=P2.corelength, plans=plan)
    # the adversarial set (NaN / -inf fields, no core, n < c, empty and stop-only records, trimmed stops) with a short
    # core length, three ranks' worth of records dealt to two so that the blocks differ in size
    from plaac_amd import native
    ac, ao = _adversarial(native)
    P3 = oc.build_params(corelength=25)
    aplan = pdist.shard_plan(ao, world)
    a_cs, a_os = pdist.extract_shard(ac, ao, aplan[rank])
    a_rows = oc.score_batch(P3, a_cs, a_os)
    out_a = pdist.gather_rows(a_rows, aplan[rank], len(ao) - 1, offsets=ao, corelength=25)
    # exchange (ii) without the funnel (RangeExchange): an all-to-all after which rank d holds records [b[d], b[d+1]) of the
    # table in input order; brought to rank 0 (gather_ranges) it is the same table
    import torch
    ranges = []
    for rows_l, pl, of, c in ((rows_s, plan, offs, P2.corelength), (a_rows, aplan, ao, 25)):
        rx = pdist.RangeExchange(pl, of, rank, world)
        mine = torch.full((rx.count, pdist.ROW_BYTES), 0xFF, dtype=torch.uint8)
        rx.exchange(torch.from_numpy(rows_l.view(np.uint8).reshape(-1).copy()), c, mine)
        b = pdist.range_bounds(len(of) - 1, world)
        assert (rx.first, rx.count) == (b[rank], b[rank + 1] - b[rank]) and sum(rx.send_counts) == len(pl[rank])
        ranges.append(pdist.gather_ranges(mine, len(of) - 1))
    if rank == 0:
        assert ranges[0].numpy().tobytes() == oc.score_batch(P2, codes, offs).tobytes(), "range exchange differs"
        assert ranges[1].numpy().tobytes() == oc.score_batch(P3, ac, ao).tobytes(), "range exchange of the adversarial set differs"
    else:
        assert ranges == [None, None]
    if rank == 0:
        want = oc.score_batch(P2, codes, offs)
        assert out.tobytes() == want.tobytes()
        assert out_w.tobytes() == want.tobytes(), "wire-row gather differs"
        assert out_p.tobytes() == want.tobytes(), "wire-row gather with known plans differs"
        want_a = oc.score_batch(P3, ac, ao)
        assert np.isnan(want_a["core_score"]).any() and np.isinf(want_a["llr_score"]).any() and (want_a["prot_len"] == 0).any()
        assert np.isnan(want_a["papa_prop"]).any() and (want_a["papa_cen"] >= 0).any()
        assert out_a.tobytes() == want_a.tobytes(), "wire-row gather of the adversarial set differs"
        open(os.path.join(tmp, "ok"), "w").write("ok")
    else:
        assert out is None and out_w is None and out_p is None and out_a is None
    dist.barrier()
    dist.destroy_process_group()


def _adversarial(native):
    """records whose rows hold every sentinel: empty, stop-only, shorter than the core length / the PAPA window / the MW
    window, homopolymers (exact ties), prion-like repeats (cores), with and without a trailing stop"""
    rng = np.random.default_rng(8)
    aas = "ACDEFGHIKLMNPQRSTVWY"
    seqs = ["", "*", "A", "A*", "QN", "Q" * 24, "Q" * 25, "Q" * 25 + "*", "N" * 40, "N" * 41 + "*", "QNQNQNQNQNYYGGSSQQNN" * 6,
            "K" * 79, "K" * 80, "K" * 81, "DE" * 60, "P" * 90 + "*", "X" * 50, "QNX*" * 20]
    seqs += ["".join(rng.choice(list(aas), int(n))) + ("*" if k % 3 == 0 else "") for k, n in enumerate(rng.integers(1, 400, 40))]
    return native.pack(seqs)


def test_wire_rows_round_trip_on_host_and_in_torch():
    """136-byte wire rows (include/plaac_native.h): from_wire(to_wire(rows)) == rows byte for byte, in C and in the torch
    form the RCCL gather uses, and the two wire encodings are the same bytes; malformed input is refused"""
    import torch
    from oracle import oracle_ctypes as oc
    from plaac_amd import dist as pdist
    from plaac_amd import native, synth
    c1, o1 = _adversarial(native)
    c2, o2 = synth.make_batch(2, nprot=1500, seed=4, stop_fraction=0.3)
    codes = np.concatenate([c1, c2])
    offs = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
    lens = torch.from_numpy(np.diff(offs.astype(np.int64)))
    for c in (60, 25, 7, 500):
        want = oc.score_batch(oc.build_params(corelength=c), codes, offs, nthreads=4)
        wire = native.rows_to_wire(want, offs)
        assert len(wire) == 136 * len(want)
        assert native.rows_from_wire(wire, offs, c).tobytes() == want.tobytes()
        wt = pdist.rows_to_wire_torch(torch.from_numpy(want.view(np.uint8).reshape(-1).copy()), lens)
        assert wt.numpy().tobytes() == wire.tobytes()
        back = pdist.rows_from_wire_torch(wt, lens, c)
        assert back.numpy().tobytes() == want.tobytes()
    bad = want.copy()
    bad["prot_len"][5] += 7  # not the record's length, trimmed or not
    with pytest.raises(native.PlaacError):
        native.rows_to_wire(bad, offs)


def test_two_rank_shard_histogram_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


def test_four_rank_range_exchange(tmp_path):
    """the same worker over four ranks: the all-to-all of the range exchange with four ranges (a rank's rows for another
    rank's range are a slice of its shard; shards of different sizes; ranges that differ by a row)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


def test_shard_plan_balances_residues_and_keeps_every_protein():
    from plaac_amd import dist as pdist
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(11, 2000, 5000), [36000, 30000, 1]])
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    for world in (1, 2, 4, 8):
        plan = pdist.shard_plan(offs, world)
        allidx = np.sort(np.concatenate(plan))
        assert np.array_equal(allidx, np.arange(len(lens)))
        res = np.array([lens[p].sum() for p in plan])
        assert res.max() - res.min() <= 36000
        for p in plan:
            assert np.all(np.diff(p) > 0)


def _numpy_shard_plan(offsets, world):
    """the scheme restated in numpy (sort by length, deal): stable descending-length sort, boustrophedon"""
    lens = np.diff(np.asarray(offsets).astype(np.int64))
    order = np.argsort(-lens, kind="stable")
    pos = np.arange(len(order))
    lap, col = pos // world, pos % world
    owner_sorted = np.where(lap % 2 == 0, col, world - 1 - col)
    owner = np.empty(len(order), dtype=np.int64)
    owner[order] = owner_sorted
    return [np.nonzero(owner == r)[0] for r in range(world)]


def test_the_c_partitioner_is_the_length_dealt_scheme():
    """plaac_shard_plan (C: the node layer, dist.py and bench.py all call it) against the numpy restatement, incl. ties in
    length (stability), empty records, more shards than records, one record, no record"""
    from plaac_amd import dist as pdist
    rng = np.random.default_rng(11)
    cases = [np.concatenate([rng.integers(0, 300, 4000), [70000, 36000, 36000, 0, 0, 11]]), rng.integers(5, 8, 1000),
             np.array([5, 5, 5]), np.array([7]), np.array([], dtype=np.int64), rng.integers(0, 1 << 20, 513)]
    for lens in cases:
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        for world in (1, 2, 3, 5, 8, 16):
            got, want = pdist.shard_plan(offs, world), _numpy_shard_plan(offs, world)
            assert len(got) == world
            for g, w in zip(got, want):
                assert np.array_equal(g, w)


def test_extract_shard_roundtrip():
    from plaac_amd import dist as pdist
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 50, 200)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    codes = rng.integers(0, 22, int(offs[-1])).astype(np.uint8)
    idx = np.sort(rng.choice(200, 77, replace=False))
    c, o = pdist.extract_shard(codes, offs, idx)
    for k, i in enumerate(idx):
        assert np.array_equal(c[int(o[k]):int(o[k + 1])], codes[int(offs[i]):int(offs[i + 1])])


def test_single_process_gather_is_a_permutation():
    from plaac_amd import dist as pdist
    rows = np.arange(5 * 160, dtype=np.uint8).reshape(5, 160)
    idx = np.array([3, 0, 4, 1, 2])
    out = pdist.gather_rows(rows, idx, 5).reshape(5, 160)
    assert np.array_equal(out[idx], rows)


def test_torch_shard_plan_and_extract_equal_the_numpy_ones():
    """bench.py cuts its HBM-resident proteome with the torch versions (strong scaling: ONE proteome over the ranks)"""
    import torch
    from plaac_amd import dist as pdist
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(0, 300, 3000), [36000, 36000, 11, 11, 11]])
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    codes = rng.integers(0, 22, int(offs[-1])).astype(np.uint8)
    t_off, t_codes = torch.from_numpy(offs.astype(np.int64)), torch.from_numpy(codes)
    for world in (1, 2, 3, 8):
        plan = pdist.shard_plan(offs, world)
        for rank in range(world):
            idx = pdist.shard_plan_torch(t_off, world, rank)
            assert np.array_equal(idx.numpy(), plan[rank])
            c, o = pdist.extract_shard_torch(t_codes, t_off, idx)
            c_np, o_np = pdist.extract_shard(codes, offs, plan[rank])
            assert np.array_equal(c.numpy(), c_np) and np.array_equal(o.numpy().astype(np.uint64), o_np)


def test_range_exchange_splits_are_consistent_for_any_world_and_tiny_batches():
    """RangeExchange without a process group: for every rank the rows it sends to d are what d expects from it, every record
    arrives exactly once at the place its index names - also when there are fewer records than ranks, or none"""
    import torch
    from plaac_amd import dist as pdist
    rng = np.random.default_rng(12)
    for n, world in ((0, 2), (1, 4), (3, 4), (5, 8), (64, 3), (1000, 8), (301, 6)):
        lens = rng.integers(0, 50, n)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        plans = pdist.shard_plan(offs, world) if n else [np.zeros(0, dtype=np.int64) for _ in range(world)]
        xs = [pdist.RangeExchange(plans, offs, r, world) for r in range(world)]
        b = pdist.range_bounds(n, world)
        assert b[0] == 0 and b[-1] == n and all(b[d] <= b[d + 1] for d in range(world))
        seen = np.zeros(n, dtype=np.int64)
        for d, x in enumerate(xs):
            assert (x.first, x.count) == (b[d], b[d + 1] - b[d])
            assert x.recv_counts == [xs[s].send_counts[d] for s in range(world)]
            assert sum(x.send_counts) == len(plans[d]) and sum(x.recv_counts) == x.count
            pos = x.recv_pos.numpy()
            assert sorted(pos.tolist()) == list(range(x.count))  # every place of the range exactly once
            seen[pos + x.first] += 1
            # the arriving records' lengths are those of the records at these places
            assert np.array_equal(x.recv_lens.numpy(), lens[pos + x.first])
        assert np.all(seen == 1)

