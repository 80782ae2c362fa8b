"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Bar (BASELINE.json north_star): HMM parse boundaries and core start/end bit-exact; here EVERY integer
field must be equal and every float field must have identical bits (NaN == NaN), because both sides
evaluate the same fp64 operation order without FMA. The only tolerance is on the posteriors, which go
through exp(): the device libm and glibc may differ in the last ulp (POST_RTOL below).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, runs_of_ones

pytestmark = pytest.mark.gpu

POST_RTOL = 1e-12  # exp() of device libm vs glibc; everything else is compared bit for bit

INT_FIELDS = ("mw_score", "mw_start", "mw_end", "llr_start", "llr_end", "vit_maxrun", "core_start", "core_end",
              "prd_start", "prd_end", "prot_len", "fi_numaa", "fi_maxrun", "papa_cen")
F64_FIELDS = ("llr_score", "core_score", "prd_score", "hmm_all", "hmm_vit", "fi_meanhydro", "fi_meancharge",
              "fi_meancombo", "papa_combo", "papa_prop", "papa_fi", "papa_llr", "papa_llr2")
EXACT_TRACKS = ("charge", "hydro", "fi", "plaacllr", "papa", "fix2", "plaacllrx2", "papax2")


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def assert_rows_equal(got, want, what=""):
    assert len(got) == len(want)
    for f in INT_FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, "%s int field %s differs at %s: got %s want %s" % (
            what, f, bad[:5], got[f][bad[:5]], want[f][bad[:5]])
    for f in F64_FIELDS:
        ok = same_bits(got[f], want[f])
        bad = np.nonzero(~ok)[0]
        assert bad.size == 0, "%s f64 field %s differs at %s: got %r want %r" % (
            what, f, bad[:5], got[f][bad[:5]], want[f][bad[:5]])


def valid_mask(codes, offsets):
    """positions that belong to a scored residue (not a trimmed stop)"""
    m = np.ones(len(codes), dtype=bool)
    ends = offsets[1:].astype(np.int64)
    starts = offsets[:-1].astype(np.int64)
    nz = ends > starts
    last = ends[nz] - 1
    m[last[codes[last] == 21]] = False
    return m


def assert_tracks_equal(got, want, codes, offsets, what=""):
    m = valid_mask(codes, offsets)
    for k in ("vit", "map"):
        bad = np.nonzero((got[k] != want[k]) & m)[0]
        assert bad.size == 0, "%s track %s differs at %s" % (what, k, bad[:5])
    for k in EXACT_TRACKS:
        bad = np.nonzero(~same_bits(got[k], want[k]) & m)[0]
        assert bad.size == 0, "%s track %s differs at %s: %r vs %r" % (what, k, bad[:5], got[k][bad[:5]],
                                                                         want[k][bad[:5]])
    for k in ("post0", "post1"):
        np.testing.assert_allclose(got[k][m], want[k][m], rtol=POST_RTOL, atol=1e-300, err_msg=what + k)


def both_params(native, oracle, **kw):
    return native.make_params(**kw), oracle.build_params(**kw)


@pytest.fixture(scope="module")
def ctx(native):
    c = native.Context(native.make_params())
    yield c
    c.close()


def check_batch(native, oracle, ctx, codes, offsets, tracks=True, what="", **kw):
    Pn, Po = both_params(native, oracle, **kw)
    ctx.set_params(Pn)
    want = oracle.score_batch(Po, codes, offsets, tracks=tracks, nthreads=8)
    got = ctx.score(codes, offsets, tracks=tracks)
    if tracks:
        assert_rows_equal(got[0], want[0], what)
        assert_tracks_equal(got[1], want[1], codes, offsets, what)
        # summary mode takes another window kernel (filter tier + exact tier): same rows
        assert_rows_equal(ctx.score(codes, offsets), want[0], what + " (summary mode)")
        return got[0]
    assert_rows_equal(got, want, what)
    return got


def test_kat28_viterbi_boundaries(native, oracle, ctx, kat28):
    """The reference's own known answers: 28 PrD [start-end] annotations, fg = prd_freq_scer_04."""
    recs, rows = kat28
    fg04 = np.loadtxt(os.path.join(GOLDEN, "prd_freq_scer_04.txt"), usecols=0)
    ctx.set_params(native.make_params(fg=fg04))
    codes, offs = native.pack([s for _, s in recs])
    got, tr = ctx.score(codes, offs, tracks=True)
    for i, (gene, orf, s, e) in enumerate(rows):
        v = tr["vit"][int(offs[i]):int(offs[i]) + int(got["prot_len"][i])]
        assert (s, e) in runs_of_ones(v), (gene, orf, s, e, runs_of_ones(v))


def test_classic_prions_config1(native, oracle, ctx, classic4):
    """BASELINE config 1 (Sup35p, -c 60 -a 1.0) plus the three other classic prions."""
    codes, offs = native.pack([s for _, s in classic4])
    rows = check_batch(native, oracle, ctx, codes, offs, what="classic4")
    # plausibility anchors: Sup35p core [5-64], PrD [1-133]
    assert (rows["core_start"][0] + 1, rows["core_end"][0] + 1) == (5, 64)
    assert (rows["prd_start"][0] + 1, rows["prd_end"][0] + 1) == (1, 133)
    assert abs(rows["core_score"][0] - 51.215) < 1e-3


def test_kat28_full_rows(native, oracle, ctx, kat28):
    recs, _ = kat28
    codes, offs = native.pack([s for _, s in recs])
    check_batch(native, oracle, ctx, codes, offs, what="kat28-default")
    fg04 = np.loadtxt(os.path.join(GOLDEN, "prd_freq_scer_04.txt"), usecols=0)
    check_batch(native, oracle, ctx, codes, offs, what="kat28-fg04", fg=fg04)


def test_yeast_shaped_batch(native, oracle, ctx):
    from plaac_amd import synth
    codes, offs = synth.make_batch(2, nprot=1500, stop_fraction=0.2)
    rows = check_batch(native, oracle, ctx, codes, offs, what="cfg2")
    assert (rows["core_start"] >= 0).sum() > 10  # the generator does produce cores


@pytest.mark.parametrize("kw", [
    dict(alpha=0.5), dict(alpha=0.0), dict(corelength=30), dict(corelength=90),
    dict(ww1=21, ww2=31), dict(ww1=61, ww2=41), dict(ww1=40, ww2=40), dict(ww1=101, ww2=81),
    dict(ww1=3, ww2=1), dict(ww1=301, ww2=201), dict(adjustprolines=False),
])
def test_parameter_variants(native, oracle, ctx, kw):
    from plaac_amd import synth
    codes, offs = synth.make_batch(2, nprot=300, seed=77, stop_fraction=0.1)
    if "alpha" in kw:
        kw = dict(kw, bgcounts=oracle.histogram(codes, offs).astype(np.float64))
    check_batch(native, oracle, ctx, codes, offs, what=str(kw), **kw)


def test_edge_lengths(native, oracle, ctx):
    """n = 1, 2, ... around every window/threshold; trailing stops; X runs; empty and stop-only records."""
    rng = np.random.default_rng(5)
    aas = "ACDEFGHIKLMNPQRSTVWY"
    seqs = []
    for n in list(range(1, 90)) + [119, 120, 121, 127, 128, 129, 160, 161, 255, 256, 257, 1000]:
        seqs.append("".join(rng.choice(list(aas), n)))
    seqs += ["", "*", "A*", "Q" * 70, "N" * 200 + "*", "X" * 50, "QNQNQ" * 30 + "XXXX" + "QN" * 40, "m" * 61,
             "PPPPPPPPPP" * 8, "P.P-P P" * 12, "QQQQ**", "KDEKR" * 20, "qnqnqnqnyy" * 9]
    codes, offs = native.pack(seqs)
    rows = check_batch(native, oracle, ctx, codes, offs, what="edges")
    assert rows["prot_len"][seqs.index("")] == 0 and rows["prot_len"][seqs.index("*")] == 0
    check_batch(native, oracle, ctx, codes, offs, what="edges-c1", corelength=1, ww1=5, ww2=7)


def test_long_protein_and_mixed_lengths(native, oracle, ctx):
    from plaac_amd import synth
    P = native.make_params()
    rng = np.random.default_rng(9)
    lens = np.array([36000, 11, 8192, 64, 63, 65, 20000, 1, 300, 300, 300, 5000], dtype=np.int64)
    codes, offs = synth.residues(lens, np.array(P.fg), np.array(P.bg), rng)
    check_batch(native, oracle, ctx, codes, offs, what="long")


def test_human_shaped_two_pass(native, oracle, ctx):
    """config 3 shape: background from the input itself (histogram pass), alpha = 0.5"""
    from plaac_amd import synth
    codes, offs = synth.make_batch(3, nprot=400, stop_fraction=0.3)
    counts = ctx.histogram(codes, offs)
    assert np.array_equal(counts, oracle.histogram(codes, offs))
    check_batch(native, oracle, ctx, codes, offs, tracks=False, what="cfg3", alpha=0.5,
                bgcounts=counts.astype(np.float64))


def test_histogram_validity_rules(native, oracle, ctx):
    seqs = ["MKV*", "MKVX", "XMKV", "MK*V", "MKV**", "M", "X", "*", "", "MXKV", "mkvl", "AAAA*", "A" * 1000 + "*",
            "A" * 999 + "X", "*AAAA"]
    codes, offs = native.pack(seqs)
    got = ctx.histogram(codes, offs)
    assert np.array_equal(got, oracle.histogram(codes, offs))
    assert got.sum() > 0


def _random_records(rng, nrec, bad_fraction, maxlen=600, x_rate=0.0, stop_fraction=0.3, empty_fraction=0.01):
    """codes/offsets of random records; bad_fraction of them get an X or a stop at a random position (which may be a
    position the rule exempts: 0, or the last one for a stop)"""
    lens = rng.integers(1, maxlen, nrec)
    lens[rng.random(nrec) < empty_fraction] = 0
    offs = np.zeros(nrec + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    codes = rng.integers(1, 21, int(offs[-1])).astype(np.uint8)
    if x_rate:
        codes[rng.random(len(codes)) < x_rate] = 0
    nz = np.nonzero(lens > 0)[0]
    stops = nz[rng.random(len(nz)) < stop_fraction]
    codes[(offs[1:][stops] - 1).astype(np.int64)] = 21
    bad = nz[rng.random(len(nz)) < bad_fraction]
    pos = (offs[:-1][bad] + rng.integers(0, 1 << 30, len(bad)) % lens[bad]).astype(np.int64)
    codes[pos] = np.where(rng.random(len(bad)) < 0.5, 0, 21).astype(np.uint8)
    return codes, offs


def test_histogram_streaming_kernel_random_sets(native, oracle, ctx):
    """k_hist streams the flat residue buffer (whole records per wave, flagged bytes decide validity, invalid records
    subtracted): against countaas / isvalidprotein (plaac.java:1698-1739) on 1 M random records with 5 % invalid ones,
    on X-rich text, on tiny / huge records and on every alignment of the buffer start."""
    rng = np.random.default_rng(20261003)
    codes, offs = _random_records(rng, 1_000_000, 0.05, maxlen=500)
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))
    codes, offs = _random_records(rng, 20_000, 0.3, maxlen=300, x_rate=0.02)   # most records invalid, X everywhere
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))
    codes, offs = _random_records(rng, 300_000, 0.2, maxlen=4, empty_fraction=0.3)  # 0..3-residue records
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))
    codes, offs = _random_records(rng, 40, 0.5, maxlen=400_000)                 # records longer than a wave's share
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))
    codes, offs = _random_records(rng, 30_000, 0.5, maxlen=200, x_rate=0.5)     # every other byte flagged: the ring of
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))  # flagged positions fills within a row
    codes, offs = _random_records(rng, 2_000, 0.0, maxlen=40_000, x_rate=0.3, stop_fraction=0.0)  # long records full of X
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))
    codes, offs = _random_records(rng, 400_000, 1.0, maxlen=40, stop_fraction=1.0)  # a flagged byte or two in EVERY record,
    assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs))  # about half of them invalid
    for seqs in (["X" * 5000], ["A" * 70000 + "X" + "A" * 3], ["*" * 40, "X", "AX", "XA", "A*", "*A", "AXA", "A*A", "A**"],
                 ["A" * 17] * 1000 + ["AXA"] + ["C" * 15] * 1000, [""] * 100 + ["MKV"] + [""] * 100, [""], []):
        codes, offs = native.pack(seqs)
        assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs)), seqs[:3]


def _sparse_records(rng, nrec, maxlen, x_rate, stop_fraction, last_x_fraction=0.0):
    """like _random_records for tens of MB: X at x_rate of the residues (positions drawn, not a mask over every residue), a
    stop as the last residue of stop_fraction of the records, an X there for last_x_fraction"""
    lens = rng.integers(1, maxlen, nrec)
    offs = np.zeros(nrec + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    n = int(offs[-1])
    codes = rng.integers(1, 21, n, dtype=np.uint8)
    codes[rng.integers(0, n, int(n * x_rate))] = 0
    last = (offs[1:] - 1).astype(np.int64)
    codes[last[rng.random(nrec) < stop_fraction]] = 21
    codes[last[rng.random(nrec) < last_x_fraction]] = 0
    return codes, offs


def test_histogram_window_path_on_large_sets(native, oracle, ctx):
    """k_hist's deciding-before-counting path needs whole 4 KiB groups per wave (tens of MB per batch): records much shorter
    than a group (the window of 64 record starts falls short inside a group: both paths in one group), X-rich text (several
    invalid records per group, records straddling groups), records of megabytes (blanked over hundreds of groups), and an X or
    a stop as the LAST residue of every other record (the one place where the two flagged codes differ)."""
    rng = np.random.default_rng(20261005)
    for nrec, maxlen, x_rate, stops, last_x in ((2_000_000, 60, 0.01, 0.5, 0.1), (200_000, 600, 0.02, 0.3, 0.0),
                                                (60, 2_000_000, 1e-5, 0.5, 0.3), (600_000, 200, 0.0, 0.3, 0.3)):
        codes, offs = _sparse_records(rng, nrec, maxlen, x_rate, stops, last_x)
        assert np.array_equal(ctx.histogram(codes, offs), oracle.histogram(codes, offs)), (nrec, maxlen)


def test_histogram_device_entry_any_alignment(native, oracle, ctx):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(7)
    codes, offs = _random_records(rng, 5000, 0.1, maxlen=200)
    dev = torch.device("cuda:0")
    buf = torch.zeros(len(codes) + 64, dtype=torch.uint8, device=dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    cnt = torch.zeros(22, dtype=torch.int64, device=dev)
    want = oracle.histogram(codes, offs)
    for shift in (0, 1, 7, 15, 16, 33):
        buf[shift:shift + len(codes)] = torch.from_numpy(codes).to(dev)
        torch.cuda.synchronize()
        ctx.histogram_device(buf.data_ptr() + shift, d_offs.data_ptr(), len(offs) - 1, cnt.data_ptr())
        ctx.sync()
        assert np.array_equal(cnt.cpu().numpy(), want), shift


def test_reciprocal_selftest_passes_at_context_creation(native):
    """plaac_ctx_create checks every tabulated reciprocal against the host's IEEE quotient and the in-kernel reciprocal
    of every denominator up to 2^19 against the device's (SharedDiv is the correctly rounded quotient only then);
    a context exists only if both passed"""
    with native.Context(native.make_params()) as c:
        assert c.score(*native.pack(["MKVLAAGQQNNQQ" * 10]))["prot_len"][0] == 130


def test_results_independent_of_batch_split(native, ctx):
    from plaac_amd import synth
    ctx.set_params(native.make_params())
    codes, offs = synth.make_batch(2, nprot=257, seed=3)
    full = ctx.score(codes, offs)
    cut = 100
    a = ctx.score(codes[:int(offs[cut])], offs[:cut + 1])
    b = ctx.score(codes[int(offs[cut]):], offs[cut:] - offs[cut])
    assert full.tobytes() == np.concatenate([a, b]).tobytes()


def test_device_resident_entry_matches_host_entry(native, ctx):
    torch = pytest.importorskip("torch")
    from plaac_amd import synth
    ctx.set_params(native.make_params())
    codes, offs = synth.make_batch(2, nprot=500, seed=11)
    want = ctx.score(codes, offs)
    dev = torch.device("cuda:0")
    d_codes = torch.from_numpy(codes).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_rows = torch.zeros(len(want) * native.ROW_BYTES, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.score_device(d_codes.data_ptr(), d_offs.data_ptr(), len(want), int(offs[-1]), d_rows.data_ptr())
    ctx.sync()
    got = d_rows.cpu().numpy().view(native.ROW_DTYPE)
    assert got.tobytes() == want.tobytes()
    t = ctx.last_timings()
    assert t["total"] > 0


def test_full_size_config2_properties(native, ctx):
    """BASELINE config 2 at full size (5,880 proteins): size-independent invariants of the outputs."""
    from plaac_amd import synth
    P = native.make_params()
    ctx.set_params(P)
    codes, offs = synth.make_batch(2)
    rows, tr = ctx.score(codes, offs, tracks=True)
    n = rows["prot_len"]
    c = P.corelength
    assert np.array_equal(n, np.diff(offs).astype(np.int64))
    has = rows["core_start"] >= 0
    # debug checks of the reference (plaac.java:841-848): longest PrD run >= c  <=>  a core exists
    assert np.array_equal(has, rows["vit_maxrun"] >= c)
    assert np.all(rows["core_end"][has] - rows["core_start"][has] + 1 == c)
    assert np.all(rows["prd_start"][has] <= rows["core_start"][has])
    assert np.all(rows["prd_end"][has] >= rows["core_end"][has])
    assert np.all(np.isnan(rows["core_score"][~has])) and np.all(rows["prd_score"][~has] == 0.0)
    # the unmasked LLR window can only be at least as good as the masked core window, up to the rounding
    # noise the -1e6 mask puts into the prefix sums (ulp(3.6e10) = 7.6e-6 for a 36k-residue protein)
    assert np.all(rows["llr_score"][has] >= rows["core_score"][has] - 1e-4)
    # posteriors are a (LUT-noisy) probability pair
    m = valid_mask(codes, offs)
    s = tr["post0"][m] + tr["post1"][m]
    assert np.all(np.abs(s - 1.0) < 1e-3)
    assert np.all(rows["mw_score"] <= np.minimum(n, 80))
    assert np.all((rows["fi_maxrun"] <= n) & (rows["fi_numaa"] <= n))


def test_context_reuse_with_growing_and_shrinking_batches(native, oracle, ctx):
    """work buffers are grown on demand and reused: small -> large -> small -> larger must all stay exact"""
    from plaac_amd import synth
    Pn, Po = both_params(native, oracle)
    ctx.set_params(Pn)
    for nprot, seed in ((7, 1), (3000, 2), (65, 3), (9000, 4), (1, 5), (640, 6)):
        codes, offs = synth.make_batch(4, nprot=nprot, seed=seed)
        assert_rows_equal(ctx.score(codes, offs), oracle.score_batch(Po, codes, offs, nthreads=8), "n=%d" % nprot)


def test_resident_batch_histogram_score_and_sweep(native, oracle, ctx):
    """upload once; background pass, scoring pass and a BASELINE config-5 style alpha x core-length sweep reuse
    the resident residues. Every sweep point must equal an independent oracle run."""
    from plaac_amd import synth
    codes, offs = synth.make_batch(4, nprot=700, seed=31, stop_fraction=0.1)
    ctx.set_params(native.make_params())
    with ctx.upload(codes, offs) as batch:
        counts = batch.histogram()
        assert np.array_equal(counts, oracle.histogram(codes, offs))
        bg = counts.astype(np.float64)
        points = [(a, c) for a in (0.0, 0.5, 1.0) for c in (30, 60, 90)]
        psets = [native.make_params(alpha=a, corelength=c, bgcounts=bg) for a, c in points]
        got = batch.sweep(psets)                # grouped: shared passes per alpha
        naive = batch.sweep(psets, naive=True)  # one full pass per point
        for (a, c), rows, rows_naive in zip(points, got, naive):
            want = oracle.score_batch(oracle.build_params(alpha=a, corelength=c, bgcounts=bg), codes, offs, nthreads=8)
            assert_rows_equal(rows, want, "alpha=%s c=%d" % (a, c))
            assert rows.tobytes() == rows_naive.tobytes()
        # more core lengths than one launch carries (MAXC = 4), unsorted, with a duplicate and one > every protein
        cs = [60, 15, 90, 33, 60, 7, 100000, 45]
        got = batch.sweep([native.make_params(alpha=0.25, corelength=c, bgcounts=bg) for c in cs])
        for c, rows in zip(cs, got):
            want = oracle.score_batch(oracle.build_params(alpha=0.25, corelength=c, bgcounts=bg), codes, offs, nthreads=8)
            assert_rows_equal(rows, want, "c=%d" % c)
        # the context still works for ordinary calls after handing its staging buffers to the batch
        ctx.set_params(native.make_params())
        assert_rows_equal(ctx.score(codes, offs), oracle.score_batch(oracle.build_params(), codes, offs, nthreads=8))
        rows, tr = batch.score(tracks=True)
        want, wtr = oracle.score_batch(oracle.build_params(), codes, offs, tracks=True, nthreads=8)
        assert_rows_equal(rows, want)
        assert_tracks_equal(tr, wtr, codes, offs)


@pytest.mark.parametrize("spread", ["1", "0"])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_sweep_on_adversarial_sequences_in_both_scheduling_modes(native, oracle, monkeypatch, mode, spread):
    """a sweep whose dependent alpha groups take plaacllr / plaacllrx2 from the llr-only refine kernel (listed centres)
    and from the one-wave-per-protein kernel (what the exact tier scored: the adversarial set is full of those), with
    the core lists of the throughput-bound schedule (mode 0) and the in-kernel core sweep of the chain-bound one (1);
    the second and third group on the idle role streams of both priority classes (spread 1) or on streams of their own"""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_LATENCY_MODE", mode)
    monkeypatch.setenv("PLAAC_SWEEP_SPREAD", spread)
    P = native.make_params()
    c1, o1 = _adversarial_batch(native)
    c2, o2 = synth.make_batch(4, nprot=3000, seed=77, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.05)
    codes = np.concatenate([c1, c2])
    offs = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
    points = [(a, c) for a in (1.0, 0.0, 0.5) for c in (30, 60, 90, 20, 45)]  # five core lengths: two launches per group
    with native.Context(P) as c:
        with c.upload(codes, offs) as batch:
            bg = batch.histogram().astype(np.float64)
            got = batch.sweep([native.make_params(alpha=a, corelength=cl, bgcounts=bg) for a, cl in points])
            nfb = c.last_exact_fallbacks()
    assert nfb > 0
    for (a, cl), rows in zip(points, got):
        want = oracle.score_batch(oracle.build_params(alpha=a, corelength=cl, bgcounts=bg), codes, offs, nthreads=8)
        assert_rows_equal(rows, want, "alpha=%s c=%d mode=%s" % (a, cl, mode))


@pytest.mark.parametrize("par", ["1", "0"])
def test_masked_core_of_long_proteins_parallel_and_serial(native, oracle, monkeypatch, par):
    """latency form, long wave-groups: the masked prefix sums come from k_core_par (exact sums on the rounding grid of the
    chain, position-parallel) or, with PLAAC_CORE_PAR=0, from the serial chain. Proteins that begin inside a PrD run
    (no masked residue yet: the serial part of k_core_par), that consist of one run, with many short runs, with runs
    after thousands of masked residues, and ordinary ones; several core lengths."""
    monkeypatch.setenv("PLAAC_LATENCY_MODE", "1")
    monkeypatch.setenv("PLAAC_CORE_PAR", par)
    rng = np.random.default_rng(4242)
    aas = "ACDEFGHIKLMNPQRSTVWY"
    bgp = np.array(native.make_params().bg)[1:21]
    bgp = bgp / bgp.sum()

    def bgseq(n):
        return "".join(rng.choice(list(aas), n, p=bgp))

    def prd(n):
        return "".join(rng.choice(list("QNQNSGYQ"), n))

    seqs = [prd(400) + bgseq(9000) + prd(200) + bgseq(300),              # begins in a run
            prd(5000),                                                    # one run, never masked
            bgseq(30000) + prd(150) + bgseq(3000) + prd(90) + bgseq(500),  # runs after tens of thousands of masked residues
            "".join(bgseq(150) + prd(int(k)) for k in rng.integers(20, 200, 40)),  # many runs of all sizes
            prd(59) + bgseq(4000) + prd(60) + bgseq(10),                  # runs of c-1 and c residues
            bgseq(2500), bgseq(36000) + prd(300), prd(70) + "X" * 3 + prd(70) + bgseq(3000)]
    seqs += [bgseq(int(n)) + prd(int(m)) + bgseq(200) for n, m in zip(rng.integers(10, 3000, 70), rng.integers(0, 150, 70))]
    codes, offs = native.pack(seqs)
    for kw in ({}, {"corelength": 30}, {"corelength": 140, "alpha": 0.4, "bgcounts": np.arange(22.0) + 3}):
        Pn, Po = both_params(native, oracle, **kw)
        with native.Context(Pn) as c:
            assert_rows_equal(c.score(codes, offs), oracle.score_batch(Po, codes, offs, nthreads=8),
                              what="core par=%s %s" % (par, kw))


def test_proteins_beyond_the_last_length_bin(native, oracle, ctx):
    """lengths >= 65535 share one (unsorted) length bin of the planner: group row counts must use the true max"""
    from plaac_amd import synth
    P = native.make_params()
    rng = np.random.default_rng(77)
    lens = np.array([100, 65535, 70001, 65534, 66000, 12, 65600, 300], dtype=np.int64)
    codes, offs = synth.residues(lens, np.array(P.fg), np.array(P.bg), rng)
    check_batch(native, oracle, ctx, codes, offs, what="beyond-bins")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,lo,hi,nprot", [(11, 0, 30, 6000), (12, 1, 70, 4000), (13, 35, 300, 3000)])
def test_many_tiny_proteins_share_one_position_stream(native, oracle, ctx, seed, lo, hi, nprot):
    """The window kernel lays 32 proteins end to end on one position axis (k_tracks20s): many very short records, empty
    records and stop-only records in a row, a few long ones between them, several segments per 256-position iteration."""
    rng = np.random.default_rng(seed)
    aas = list("ACDEFGHIKLMNPQRSTVWY")
    lens = rng.integers(lo, hi + 1, nprot)
    lens[rng.integers(0, nprot, 12)] = rng.integers(900, 2600, 12)
    # prion-like stretches make cores / PAPA centres exist
    seqs = []
    for n in lens:
        s = rng.choice(aas, int(n))
        if n >= 40 and rng.random() < 0.5:
            a = int(rng.integers(0, n - 30))
            s[a:a + 30] = rng.choice(list("QNSYG"), 30)
        seqs.append("".join(s) + ("*" if rng.random() < 0.2 else ""))
    codes, offs = native.pack(seqs)
    check_batch(native, oracle, ctx, codes, offs, tracks=bool(seed & 1), what="tiny%d" % seed)


@pytest.mark.gpu
def test_stream_and_per_protein_window_kernels_agree(native, oracle):
    """PLAAC_KB_PER_PROTEIN=1 selects the older one-protein-at-a-time form of the ww=41 window kernel: both forms must
    give the same bytes (and therefore both equal the oracle, which other tests check for the default form)."""
    from plaac_amd import synth
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=20000, fg=np.array(P.fg), bg=np.array(P.bg))
    rows = {}
    for flag in ("0", "1"):
        os.environ["PLAAC_KB_PER_PROTEIN"] = flag
        try:
            with native.Context(P) as c:
                rows[flag] = c.score(codes, offs, tracks=False)
        finally:
            os.environ.pop("PLAAC_KB_PER_PROTEIN", None)
    assert rows["0"].tobytes() == rows["1"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("clamp", ["2", "1", "0"])
@pytest.mark.parametrize("mode", ["0", "1"])
def test_log_sum_exp_with_and_without_its_in_range_select(native, oracle, monkeypatch, clamp, mode):
    """the forward / backward kernels leave out the in-range select of the LUT log-sum-exp when the tables allow it
    (lse_clamp_ok: every emission log-probability <= -0.125, the table's last entry below 2^-57; PLAAC_LSE_CLAMP=1 stops
    there) and the clamp of the table index as well when the two states provably never drift 39 apart (lse_range_ok, the
    default for the reference's tables); PLAAC_LSE_CLAMP=0 keeps the select. All must give the oracle's rows and posteriors,
    in the throughput and the latency forms - also for tables that do NOT qualify: a background of nearly one residue (an
    emission log-probability close to zero: no clamped form) and one with a residue of frequency 1e-12 (emission log-odds
    of 25: the drift bound fails, the unclamped form is refused)."""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_LSE_CLAMP", clamp)
    monkeypatch.setenv("PLAAC_LATENCY_MODE", mode)
    P = native.make_params()
    codes, offs = synth.make_batch(2, nprot=700, seed=5, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.05)
    lopsided = np.full(22, 1e-4)
    lopsided[13] = 1.0  # background almost all proline: log(freq) ~ -0.002 > -0.125
    rare = np.ones(22)
    rare[5] = 1e-12
    for kw in ({}, {"alpha": 0.0, "bgcounts": lopsided}, {"alpha": 0.0, "bgcounts": rare}):
        with native.Context(native.make_params(**kw)) as c:
            check_batch(native, oracle, c, codes, offs, tracks=True, what="lse clamp %s mode %s %s" % (clamp, mode, kw), **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["1", "0", "1, core search behind the whole pass", "1, chains first", "1, own streams", "1, own streams, the other wave-groups aside"])
def test_chain_bound_sweep_with_long_proteins_in_both_viterbi_forms(native, oracle, monkeypatch, form):
    """a chain-bound sweep over a batch whose first wave-groups hold proteins of >= 2048 residues: those groups are a run of
    their own in the latency form of k_vit, their core windows come from k_core_chain / _eval / _reduce per core length
    with scratch per sweep group (form 1); form 0 keeps the throughput form for every wave-group. Five core lengths per
    alpha: two launches per group. Prion-like long proteins, so that the long groups do have cores."""
    from plaac_amd import synth
    # round 5: the core search of the long wave-groups runs on the group's forward stream as soon as the latency-form pass over
    # them is through (default); the older place and the other launch order are forms of their own
    if "behind the whole pass" in form:
        monkeypatch.setenv("PLAAC_SWEEP_CORE_ASIDE", "0")
    if "chains first" in form:
        monkeypatch.setenv("PLAAC_SWEEP_CHAINS_FIRST", "1")
    if "own streams" in form:  # (as with GPU_MAX_HW_QUEUES >= 12: the other wave-groups' Viterbi pass then runs beside the long chains)
        monkeypatch.setenv("PLAAC_SWEEP_SPREAD", "0")
    if "wave-groups aside" in form:
        monkeypatch.setenv("PLAAC_SWEEP_REST_ASIDE", "1")
    form = form[0]
    monkeypatch.setenv("PLAAC_LATENCY_MODE", "1")
    monkeypatch.setenv("PLAAC_SWEEP_LATENCY", form)
    P = native.make_params()
    rng = np.random.default_rng(31)
    aas = "ACDEFGHIKLMNPQRSTVWY"
    longs = ["".join(rng.choice(list("QNGSYQNQ"), 700)) + "".join(rng.choice(list(aas), n)) + "QN" * 150
             for n in (1200, 2100, 4000, 9000)] + ["".join(rng.choice(list(aas), 2048)), "".join(rng.choice(list(aas), 2047))]
    c1, o1 = native.pack(longs)
    c2, o2 = synth.make_batch(4, nprot=9000, seed=78, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.05)
    codes = np.concatenate([c1, c2])
    offs = np.concatenate([o1, o2[1:] + o1[-1]]).astype(np.uint64)
    points = [(a, c) for a in (1.0, 0.0, 0.5) for c in (30, 60, 90, 20, 45)]
    with native.Context(P) as c:
        with c.upload(codes, offs) as batch:
            bg = batch.histogram().astype(np.float64)
            got = batch.sweep([native.make_params(alpha=a, corelength=cl, bgcounts=bg) for a, cl in points])
    ncores = 0
    for (a, cl), rows in zip(points, got):
        want = oracle.score_batch(oracle.build_params(alpha=a, corelength=cl, bgcounts=bg), codes, offs, nthreads=8)
        assert_rows_equal(rows, want, "alpha=%s c=%d viterbi form %s" % (a, cl, form))
        ncores += int((want["core_start"][:4] >= 0).sum())
    assert ncores > 0  # the long proteins do have core windows


@pytest.mark.parametrize("ww", [dict(ww1=31, ww2=21), dict(ww1=41, ww2=41, ww3=21), dict(ww1=41, ww2=21, ww3=61)])
def test_sweep_over_alpha_with_other_windows(native, oracle, ctx, ww):
    """The further alphas of a sweep take the PAPA centre from the first alpha's window kernel and recompute only the two
    llr-derived values there (k_llr_at_centre). With ww3 != ww2 the centre (PAPA window) can lie in the NaN margin of
    the llr window; with windows other than 41 the generic window kernel supplies the centre."""
    from plaac_amd import synth
    codes, offs = synth.make_batch(2, nprot=300, seed=5, stop_fraction=0.1)
    extra = native.pack(["QNQNQNQNQNYYGGSSQQNN" * k for k in (1, 2, 3, 5)] + ["A", "", "QN"])
    codes = np.concatenate([codes, extra[0]])
    offs = np.concatenate([offs, offs[-1] + extra[1][1:]])
    bg = oracle.histogram(codes, offs).astype(np.float64)
    with ctx.upload(codes, offs) as batch:
        points = [(a, c) for a in (1.0, 0.3, 0.0) for c in (20, 60)]
        got = batch.sweep([native.make_params(alpha=a, corelength=c, bgcounts=bg, **ww) for a, c in points])
        for (a, c), rows in zip(points, got):
            want = oracle.score_batch(oracle.build_params(alpha=a, corelength=c, bgcounts=bg, **ww), codes, offs,
                                      nthreads=8)
            assert_rows_equal(rows, want, "%s alpha=%s c=%d" % (ww, a, c))
    ctx.set_params(native.make_params())


@pytest.mark.gpu
@pytest.mark.parametrize("key,fasta,kw", [
    ("classic4_default", "four_classic_prions.fasta", {}),
    ("classic4_c40_alpha0", "four_classic_prions.fasta",
     dict(corelength=40, alpha=0.0, bgcounts=np.arange(22, dtype=np.float64) + 5.0)),
    ("kat28_fg04", "kat28.fasta", "fg04"),
])
def test_hip_path_reproduces_the_committed_rows(native, ctx, key, fasta, kw):
    """HIP path through the C ABI against tests/golden/oracle_rows.json (committed oracle output, floats as bit
    patterns): the golden-fixture leg of the parity tests, independent of an oracle build on the GPU box."""
    import json
    from conftest import read_fasta_simple
    if kw == "fg04":
        kw = dict(fg=np.loadtxt(os.path.join(GOLDEN, "prd_freq_scer_04.txt"), usecols=0))
    recs = read_fasta_simple(os.path.join(GOLDEN, fasta))
    codes, offs = native.pack([s for _, s in recs])
    ctx.set_params(native.make_params(**kw))
    rows = ctx.score(codes, offs)
    with open(os.path.join(GOLDEN, "oracle_rows.json")) as f:
        want = json.load(f)[key]
    for r, w in zip(rows, want):
        for fld in rows.dtype.names:
            if rows.dtype[fld].kind == "f":
                g, x = float(r[fld]), (float("nan") if w[fld] == "nan" else float.fromhex(w[fld]))
                assert (g != g and x != x) or g.hex() == x.hex(), "%s %s %s: %r vs %r" % (key, w["name"], fld, g, x)
            else:
                assert int(r[fld]) == w[fld], "%s %s %s" % (key, w["name"], fld)
    ctx.set_params(native.make_params())


def test_set_params_waits_for_the_batch_in_flight_on_a_caller_stream(native, oracle):
    """plaac_ctx_set_params between two plaac_score_device calls on an external (non-blocking) stream: the table upload
    must wait for the kernels of the first call (they read the tables on the caller's stream and on the side streams),
    and the second call must see the new tables. Both results are checked against the oracle."""
    torch = pytest.importorskip("torch")
    from plaac_amd import synth
    P1, O1 = both_params(native, oracle)
    codes, offs = synth.make_batch(4, nprot=60000, seed=5, fg=np.array(P1.fg), bg=np.array(P1.bg))
    counts = oracle.histogram(codes, offs).astype(np.float64)
    P2, O2 = both_params(native, oracle, alpha=0.0, corelength=30, bgcounts=counts)
    dev = torch.device("cuda:0")
    d_codes = torch.from_numpy(codes).to(dev)
    d_offs = torch.from_numpy(offs.astype(np.int64)).to(dev)
    n, total = len(offs) - 1, int(offs[-1])
    r1 = torch.zeros(n * native.ROW_BYTES, dtype=torch.uint8, device=dev)
    r2 = torch.zeros_like(r1)
    st = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    want1 = oracle.score_batch(O1, codes, offs, nthreads=8)
    want2 = oracle.score_batch(O2, codes, offs, nthreads=8)
    with native.Context(P1) as c:
        for rep in range(3):  # several rounds: the race (if any) depends on timing
            c.set_params(P1)
            c.score_device(d_codes.data_ptr(), d_offs.data_ptr(), n, total, r1.data_ptr(), stream=st.cuda_stream)
            c.set_params(P2)  # no host sync in between
            c.score_device(d_codes.data_ptr(), d_offs.data_ptr(), n, total, r2.data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            assert_rows_equal(r1.cpu().numpy().view(native.ROW_DTYPE), want1, what="first call, round %d" % rep)
            assert_rows_equal(r2.cpu().numpy().view(native.ROW_DTYPE), want2, what="second call, round %d" % rep)


def test_consecutive_calls_on_different_caller_streams_are_ordered(native, oracle):
    """the ctx's plan / scratch buffers are shared by consecutive calls: two calls on two different streams must not race"""
    torch = pytest.importorskip("torch")
    from plaac_amd import synth
    P, O = both_params(native, oracle)
    dev = torch.device("cuda:0")
    batches = []
    for seed, nprot in ((31, 30000), (32, 9000)):
        codes, offs = synth.make_batch(4, nprot=nprot, seed=seed, fg=np.array(P.fg), bg=np.array(P.bg))
        batches.append((codes, offs, torch.from_numpy(codes).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev),
                        torch.zeros(nprot * native.ROW_BYTES, dtype=torch.uint8, device=dev)))
    s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    with native.Context(P) as c:
        for k, (codes, offs, dc, do, dr) in enumerate(batches):
            c.score_device(dc.data_ptr(), do.data_ptr(), len(offs) - 1, int(offs[-1]), dr.data_ptr(),
                           stream=s[k].cuda_stream)
        torch.cuda.synchronize()
        for codes, offs, _, _, dr in batches:
            assert_rows_equal(dr.cpu().numpy().view(native.ROW_DTYPE), oracle.score_batch(O, codes, offs, nthreads=8))


# ---- summary-mode window kernel in filter form (k_tracks20f + k_refine_centres + exact fallback) ----
def _adversarial_batch(native):
    """sequences built to defeat the filter's certainty: exact ties (perfect repeats, homopolymers), FoldIndex sitting on
    zero, PAPA plateaus, proline runs (the PP / PxP rule), X / stop codes, every short length"""
    rng = np.random.default_rng(2024)
    seqs = ["Q" * 200, "N" * 41, "QN" * 100, "A" * 500, "P" * 300, "PAP" * 90, "KE" * 150, "G" * 40, "S" * 42,
            "QQQQQQQQQQNNNNNNNNNN" * 12, ("MKVLAAGIVG" * 9 + "QNQNQNYYGS" * 9) * 3, "DEDEDEKRKR" * 30,
            "X" * 100, "QX" * 60, "Q" * 100 + "*", "W" * 45 + "Q" * 45 + "W" * 45]
    aas = "ACDEFGHIKLMNPQRSTVWY"
    for n in list(range(1, 131)) + [163, 164, 165, 447, 448, 449, 895, 896, 897, 1500]:
        seqs.append("".join(rng.choice(list(aas), n)))
    unit = "".join(rng.choice(list(aas), 97))
    seqs += [unit * 8, (unit + "Q") * 5, unit[:50] * 11]  # perfect repeats longer than both window levels
    for n in (60, 200, 600):  # low-complexity, FoldIndex near its threshold
        seqs.append("".join(rng.choice(list("GSQNYAP"), n)))
        seqs.append("".join(rng.choice(list("LIVFKE"), n)))
    rng.shuffle(seqs)
    return native.pack(seqs)


def test_filter_window_kernel_on_adversarial_sequences(native, oracle, ctx):
    codes, offs = _adversarial_batch(native)
    for kw in ({}, {"ww1": 40, "ww2": 40}, {"alpha": 0.3, "corelength": 25, "bgcounts": np.arange(22.0) + 5},
               {"adjustprolines": False}):
        check_batch(native, oracle, ctx, codes, offs, tracks=False, what="adversarial %s" % kw, **kw)
    ctx.set_params(native.make_params())
    ctx.score(codes, offs)
    assert ctx.last_exact_fallbacks() > 0  # homopolymers / repeats tie exactly: the exact kernel must have taken them


def test_filter_window_kernel_decides_random_proteins_itself(native, oracle):
    """on HMM-sampled proteomes the filter tier must decide (nearly) everything: the exact tier is a safety net, not
    the usual path"""
    from plaac_amd import synth
    P, O = both_params(native, oracle)
    codes, offs = synth.make_batch(4, nprot=40000, seed=91, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.05)
    with native.Context(P) as c:
        got = c.score(codes, offs)
        nfb = c.last_exact_fallbacks()
    assert_rows_equal(got, oracle.score_batch(O, codes, offs, nthreads=8), what="filter tier")
    assert nfb <= 40, "%d of 40000 random proteins fell back to the exact kernel" % nfb


def test_filter_and_exact_window_kernels_agree(native, monkeypatch):
    from plaac_amd import synth
    P = native.make_params()
    codes, offs = synth.make_batch(3, nprot=3000, seed=8, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
    a_codes, a_offs = _adversarial_batch(native)
    with native.Context(P) as c:
        r1, r2 = c.score(codes, offs), c.score(a_codes, a_offs)
    monkeypatch.setenv("PLAAC_KB_FILTER", "0")
    with native.Context(P) as c:
        e1, e2 = c.score(codes, offs), c.score(a_codes, a_offs)
        assert c.last_exact_fallbacks() == 0
    assert r1.tobytes() == e1.tobytes() and r2.tobytes() == e2.tobytes()


# ---- the filter tier in LANE form (k_tracksL: one lane per protein, sliding windows). The library takes it for batches of
#      >= 4096 wave-groups only; PLAAC_KB_LANE_MIN_GROUPS=1 makes every batch take it. ----
@pytest.fixture()
def lane_ctx(native, monkeypatch):
    monkeypatch.setenv("PLAAC_KB_LANE_MIN_GROUPS", "1")
    c = native.Context(native.make_params())
    yield c
    c.close()


def _uniform_length_batch(native, rng, lengths, per=64):
    """waves whose 64 proteins have the same length (the lane kernel's scalar-predicate and predicate-free blocks), with
    X, stops, proline runs and prion-like stretches in them"""
    aas = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    seqs = []
    for n in lengths:
        for k in range(per):
            a = rng.choice(aas, n)
            if n > 60 and k % 3 == 0:
                s0 = int(rng.integers(0, n - 50))
                a[s0:s0 + 50] = rng.choice(np.array(list("QNQNGSY")), 50)
            if k % 5 == 0:
                a[rng.integers(0, n, max(1, n // 40))] = "P"
            if k % 7 == 0:
                a[int(rng.integers(0, n))] = "X"
            seqs.append("".join(a))
    return native.pack(seqs)


def test_lane_form_filter_on_every_kind_of_wave(native, oracle, lane_ctx):
    rng = np.random.default_rng(31)
    # same-length waves at the lengths where the block kinds change (41: first candidate; 79/80/81: both-ends reciprocal
    # table; 85/86/101/102: first predicate-free block), and long enough for many of them
    codes, offs = _uniform_length_batch(native, rng, [1, 2, 20, 40, 41, 42, 64, 79, 80, 81, 85, 86, 101, 102, 117, 118,
                                                        285, 300, 1000, 2047])
    for kw in ({}, {"ww1": 40, "ww2": 40}, {"adjustprolines": False}, {"alpha": 0.3, "corelength": 25,
                                                                     "bgcounts": np.arange(22.0) + 5}):
        check_batch(native, oracle, lane_ctx, codes, offs, tracks=False, what="uniform waves %s" % kw, **kw)
    # mixed lengths within a wave (per-lane predicates), adversarial sequences (ties -> exact tier), skipped records
    codes, offs = _adversarial_batch(native)
    for kw in ({}, {"ww1": 40, "ww2": 40}, {"adjustprolines": False}):
        check_batch(native, oracle, lane_ctx, codes, offs, tracks=False, what="adversarial, lane form %s" % kw, **kw)
    lane_ctx.set_params(native.make_params())
    lane_ctx.score(codes, offs)
    assert lane_ctx.last_exact_fallbacks() > 0
    codes, offs = native.pack(["", "*", "Q" * 100, "", "MKV", "QN" * 60 + "*", ""])
    check_batch(native, oracle, lane_ctx, codes, offs, tracks=False, what="skipped records, lane form")


def test_lane_form_with_long_proteins_in_the_stream_form_prefix(native, oracle, lane_ctx):
    """proteins of >= 2048 residues (the long wave-groups, a prefix of the descending-length plan) stay with the stream
    form; the rest of the batch takes the lane form; both feed one refine list (in segments) and one fallback list"""
    from plaac_amd import synth
    P = native.make_params()
    codes, offs = synth.make_batch(3, nprot=2500, seed=17, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
    lens = np.diff(offs.astype(np.int64))
    assert (lens >= 2048).sum() >= 2 and (lens < 2048).sum() > 2000
    check_batch(native, oracle, lane_ctx, codes, offs, tracks=False, what="long prefix + lane form")
    check_batch(native, oracle, lane_ctx, codes, offs, tracks=False, what="long prefix + lane form, two-pass",
                alpha=0.5, bgcounts=oracle.histogram(codes, offs).astype(np.float64))


def test_lane_form_equals_stream_form_on_a_large_batch(native, oracle, monkeypatch):
    """300 k UniRef50-shaped proteins (above the library's own 4096-group threshold: the lane form without the test knob)
    against the stream form and, on a slice, the oracle; sweeps take the lane form for their first group too"""
    from plaac_amd import synth
    P, O = both_params(native, oracle)
    codes, offs = synth.make_batch(4, nprot=300_000, seed=5, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.02)
    with native.Context(P) as c:
        lane = c.score(codes, offs)
        nfb = c.last_exact_fallbacks()
        with c.upload(codes, offs) as batch:
            sets = [native.make_params(alpha=a, corelength=cl) for a in (1.0, 0.5) for cl in (60, 30)]
            swept = batch.sweep(sets)
    monkeypatch.setenv("PLAAC_KB_LANE", "0")
    with native.Context(P) as c:
        stream = c.score(codes, offs)
    assert lane.tobytes() == stream.tobytes()
    assert nfb <= 300
    n_s = 20000
    want = oracle.score_batch(O, codes[:int(offs[n_s])], offs[:n_s + 1], nthreads=8)
    assert_rows_equal(lane[:n_s], want, "lane form, 300 k batch")
    assert swept[0].tobytes() == lane.tobytes()
    for k, (a, cl) in enumerate((a, cl) for a in (1.0, 0.5) for cl in (60, 30)):
        wk = oracle.score_batch(oracle.build_params(alpha=a, corelength=cl), codes[:int(offs[n_s])], offs[:n_s + 1], nthreads=8)
        assert_rows_equal(swept[k][:n_s], wk, "sweep point %d over the lane-form batch" % k)


def _with_tables(native, oracle, hydro2=None, cc=None, **kw):
    Pn, Po = both_params(native, oracle, **kw)
    for P in (Pn, Po):
        if hydro2 is not None:
            for k in range(22):
                P.hydro2[k] = float(hydro2[k])
        if cc is not None:
            for i in range(3):
                P.cc[i] = float(cc[i])
    return Pn, Po


@pytest.mark.parametrize("tables", ["reference", "other_rationals", "irrational", "forced_fp64"])
def test_filter_tier_signs_foldindex_in_integers_or_in_bounded_fp64(native, oracle, monkeypatch, tables):
    """the filter tier takes FoldIndex signs from exact integers when the hydropathy table and cc are small rationals
    (the reference's are) and from error-bounded fp64 prefix sums otherwise: both forms against the oracle, on random and
    on adversarial sequences, with tables of either kind"""
    from plaac_amd import synth
    rng = np.random.default_rng(77)
    if tables == "other_rationals":
        Pn, Po = _with_tables(native, oracle, hydro2=(rng.integers(-30, 60, 22)) / 7.0, cc=(1.5, -2.0, -0.25))
    elif tables == "irrational":
        Pn, Po = _with_tables(native, oracle, hydro2=rng.random(22) * 1.3 - 0.2, cc=(np.e, -1.0, -1.0 / 3.0))
    else:
        Pn, Po = both_params(native, oracle)
    assert native.fi_integer_form(Pn)[0] == (tables != "irrational")
    if tables == "forced_fp64":
        monkeypatch.setenv("PLAAC_FI_INT", "0")
    codes, offs = synth.make_batch(4, nprot=6000, seed=17, fg=np.array(Pn.fg), bg=np.array(Pn.bg), stop_fraction=0.05)
    a_codes, a_offs = _adversarial_batch(native)
    with native.Context(Pn) as c:
        assert_rows_equal(c.score(codes, offs), oracle.score_batch(Po, codes, offs, nthreads=8), what=tables)
        nfb = c.last_exact_fallbacks()
        assert_rows_equal(c.score(a_codes, a_offs), oracle.score_batch(Po, a_codes, a_offs, nthreads=8),
                          what=tables + " adversarial")
    assert nfb <= 60, "%d of 6000 random proteins fell back to the exact kernel (%s)" % (nfb, tables)


@pytest.mark.parametrize("mode", ["0", "1"])
def test_throughput_and_latency_forms_of_the_chain_kernels(native, oracle, monkeypatch, mode):
    """The lane-per-protein kernels exist in a throughput form (large batches) and a latency form (batches bound by the
    chain of their longest protein: paired-lane forward, split window kernel, core window of the long wave-groups as
    prefix chain + evaluation + reduction, k_finish); the library picks per batch, PLAAC_LATENCY_MODE forces one.
    Both must reproduce the oracle bit for bit, on proteins around the 2048-residue threshold of the long-group path,
    with core lengths on either side of the 16-residue block size and windows that straddle the first block."""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_LATENCY_MODE", mode)
    P0 = native.make_params()
    rng = np.random.default_rng(606)
    lens = np.concatenate([[2047, 2048, 2049, 5000, 3001, 36000, 1, 2, 15, 16, 17, 59, 60, 61, 79, 80, 81],
                           rng.integers(1, 700, 300)])
    rng.shuffle(lens)
    codes, offs = synth.residues(lens, np.array(P0.fg), np.array(P0.bg), rng, stop_fraction=0.1)
    for kw in ({}, {"corelength": 7}, {"corelength": 16}, {"corelength": 100, "alpha": 0.4, "bgcounts": np.arange(22.0) + 1},
               {"corelength": 3000}):
        Pn, Po = both_params(native, oracle, **kw)
        with native.Context(Pn) as c:
            got = c.score(codes, offs)
            trows, tr = c.score(codes, offs, tracks=True) if kw.get("corelength", 60) in (60, 100) else (None, None)
        want, wtr = oracle.score_batch(Po, codes, offs, tracks=True, nthreads=8)
        assert_rows_equal(got, want, what="mode %s %s" % (mode, kw))
        if tr is not None:  # track mode in both forms: paired-lane forward AND backward chains feed k_post (mode 1)
            assert_rows_equal(trows, want, what="track mode, mode %s %s" % (mode, kw))
            assert_tracks_equal(tr, wtr, codes, offs, "mode %s %s" % (mode, kw))


@pytest.mark.parametrize("knobs", [{"PLAAC_FINISH_KERNEL": "1"}, {"PLAAC_FINISH_KERNEL": "0", "PLAAC_KB_PRIO": "33"},
                                   {"PLAAC_MIXED_GROUPS": "2", "PLAAC_MIXED_MIN_REST": "1", "PLAAC_FINISH_KERNEL": "1"},
                                   {"PLAAC_MIXED_GROUPS": "2", "PLAAC_MIXED_MIN_REST": "1"}],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_hmm_scores_from_the_chain_kernels_or_from_the_finishing_kernel(native, oracle, monkeypatch, knobs):
    """HMMall / HMMvit (plaac.java:797-798) are hmm1's forward / Viterbi score minus hmm0's total. Throughput-bound calls and
    the throughput-form runs of a call in mixed forms let k_fwd / k_vit carry hmm0's running sum and write the two fields
    themselves (round 4: the finishing kernel's pass over the rows cost the step its whole stand-alone time);
    PLAAC_FINISH_KERNEL=1 keeps the three terms apart and combines them in k_finish, as the latency forms always do. Same
    bits either way, with the lane-form filter at either wave priority, for consecutive overlapping calls too."""
    from plaac_amd import synth
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PLAAC_KB_LANE_MIN_GROUPS", "1")
    if "PLAAC_MIXED_GROUPS" not in knobs:
        monkeypatch.setenv("PLAAC_LATENCY_MODE", "0")
    P0 = native.make_params()
    rng = np.random.default_rng(4404)
    lens = np.concatenate([[2500, 2047, 1, 0, 16, 17, 79, 80, 81, 600], rng.integers(1, 500, 900)])
    rng.shuffle(lens)
    codes, offs = synth.residues(lens, np.array(P0.fg), np.array(P0.bg), rng, stop_fraction=0.1)
    want = oracle.score_batch(oracle.build_params(), codes, offs, nthreads=8)
    with native.Context(P0) as c:
        assert_rows_equal(c.score(codes, offs), want, what="%s" % knobs)
        c.set_overlap(True)
        for rep, got in enumerate(c.score_stream([(codes, offs)] * 4)):
            assert_rows_equal(got, want, what="%s, overlapping call %d" % (knobs, rep))


@pytest.mark.parametrize("fused", ["0", "1", "every pair kept", "two kernels, r04 launch order", "one kernel", "window kernel behind the packed copy"])
@pytest.mark.parametrize("vit_mixed", ["0", "1", "throughput-bound"])
def test_track_mode_posteriors_from_the_forward_pass_or_from_k_post(native, oracle, monkeypatch, fused, vit_mixed):
    """Chain-bound batches in track mode (round 4): the wave-groups without a long protein run the forward pass BEHIND the
    backward pass and emit posteriors and MAP bytes on the way (k_fwd_post; the reference's lpseq comes from position
    0, plaac.java:3393-3396, so the order is exact); the long wave-groups keep forward array + k_post; the Viterbi kernel
    takes its list form outside the long wave-groups. The backward pass keeps every eighth pair only (checkpoints) and the
    forward pass recomputes the seven between two of them with the backward pass's own operations - or keeps them all
    (PLAAC_TRACK_CKPT=0). Every combination against the oracle: rows, the eight window tracks, Viterbi / MAP bytes,
    posteriors - with the long wave-groups in the first run only, in several runs, and absent; lengths around the multiples
    of 8 and 16 (where a checkpoint is the protein's last pair, or the pair behind it does not exist)."""
    from plaac_amd import synth
    if fused == "every pair kept":  # the fused pass reading a full backward array instead of recomputing between checkpoints
        fused = "1"
        monkeypatch.setenv("PLAAC_TRACK_CKPT", "0")
    # round 5's default: the forward pass stores its posteriors transposed through LDS onto 64-byte sectors (k_fwd_post_t),
    # the window kernel is enqueued behind the packed copy, the Viterbi bytes ahead of k_finish
    if fused == "two kernels, r04 launch order":  # round 4's forms: a lane stores its own 16-byte runs
        fused = "1"
        monkeypatch.setenv("PLAAC_TRACK_POST_FORM", "0")
        monkeypatch.setenv("PLAAC_TRACK_POST_OCC", "2")
        monkeypatch.setenv("PLAAC_TRACK_VIT_EARLY", "0")
    if fused == "one kernel":  # k_bwd_fwd_post: a lane reads back its own checkpoints (measured slower, kept as a form)
        fused = "1"
        monkeypatch.setenv("PLAAC_TRACK_ONE_PASS", "1")
    if fused == "window kernel behind the packed copy":
        fused = "1"
        monkeypatch.setenv("PLAAC_TRACK_KB_LATE", "1")
    latency = "1"
    if vit_mixed == "throughput-bound":  # calls not bound by a chain take the same split, with the Viterbi pass in one form
        vit_mixed, latency = "1", "0"
    monkeypatch.setenv("PLAAC_TRACK_FUSED", fused)
    monkeypatch.setenv("PLAAC_TRACK_VIT_MIXED", vit_mixed)
    monkeypatch.setenv("PLAAC_MIXED_MIN_REST", "1")
    monkeypatch.setenv("PLAAC_SEGMENT_MIN_ROWS", "1")
    monkeypatch.setenv("PLAAC_LATENCY_MODE", latency)
    P0 = native.make_params()
    rng = np.random.default_rng(8800 + int(fused) * 2 + int(vit_mixed) + 10 * int(latency))
    shapes = (np.concatenate([[9000, 2100, 2048, 1, 0, 16, 17, 2, 7, 8, 9, 15, 23, 24, 25, 31, 32, 33, 40, 41], rng.integers(1, 600, 700)]),
              np.concatenate([rng.integers(2048, 2600, 200), rng.integers(1, 300, 300)]),  # long wave-groups in more than one run
              rng.integers(1, 400, 500))                                                    # none
    for lens in shapes:
        rng.shuffle(lens)
        codes, offs = synth.residues(lens, np.array(P0.fg), np.array(P0.bg), rng, stop_fraction=0.1)
        want, wtr = oracle.score_batch(oracle.build_params(), codes, offs, tracks=True, nthreads=8)
        for runs in ("1", "2"):  # (the knobs are read when a context is created)
            monkeypatch.setenv("PLAAC_TRACK_SEGMENTS", runs)
            with native.Context(P0) as c:
                trows, tr = c.score(codes, offs, tracks=True)
            what = "fused %s, k_vit mixed %s, latency forms %s, %s run(s), %d proteins" % (fused, vit_mixed, latency, runs, len(lens))
            assert_rows_equal(trows, want, what)
            assert_tracks_equal(tr, wtr, codes, offs, what)


@pytest.mark.parametrize("consec", ["0", "1", "2", "2048"])
def test_track_mode_window_kernel_in_input_order_behind_the_long_proteins(native, oracle, monkeypatch, consec):
    """Track mode (round 4): the stream form of the window kernel takes 16 CONSECUTIVE records per block - a wave's track
    stores then meet at the proteins' ends instead of leaving half-written 64-byte sectors to other waves - and only the
    long proteins are dealt from the head of the plan, to the first blocks (PLAAC_TRACK_CONSEC = how many such blocks at
    most; 0: every protein dealt from the plan, as in summary mode). Which proteins count as long is a length: 2048
    residues, or more when the first blocks cannot hold all of those (forced here with one or two blocks). Rows and all
    eight window tracks against the oracle: long proteins more than the first blocks hold, ties at the cut, skipped and
    empty records among the consecutive ones, fewer records than one block, none at all."""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_TRACK_CONSEC", consec)
    P0 = native.make_params()
    rng = np.random.default_rng(9100)
    shapes = (np.concatenate([rng.integers(2048, 2060, 40), [2048] * 5, [2047] * 5, rng.integers(0, 500, 600)]),
              np.concatenate([[2300] * 20, [2301] * 20, rng.integers(1, 80, 100)]),  # a tie at the cut of one / two blocks
              rng.integers(1, 300, 9), np.array([0, 0, 5]), np.array([], dtype=np.int64))
    for lens in shapes:
        rng.shuffle(lens)
        codes, offs = synth.residues(lens, np.array(P0.fg), np.array(P0.bg), rng, stop_fraction=0.1)
        want, wtr = oracle.score_batch(oracle.build_params(), codes, offs, tracks=True, nthreads=8)
        with native.Context(P0) as c:
            trows, tr = c.score(codes, offs, tracks=True)
        what = "PLAAC_TRACK_CONSEC=%s, %d proteins" % (consec, len(lens))
        assert_rows_equal(trows, want, what)
        assert_tracks_equal(tr, wtr, codes, offs, what)


@pytest.mark.parametrize("mode", ["0", "1"])
@pytest.mark.parametrize("nseg", ["2", "3", "8"])
def test_calls_cut_into_runs_of_wave_groups(native, oracle, monkeypatch, mode, nseg):
    """Large single-point calls are pipelined over runs of wave-groups of about equal row counts (the planner's scan leaves
    the run boundaries with the plan words): the packed copy, every lane-per-protein kernel and, in track mode, k_post are
    launched per run with the run's slice of the plan. The threshold is lowered so that small batches take the path, in
    both forms of the chain kernels, summary and track mode, the lane form of the filter tier included; group counts that
    are not multiples of the run count, runs that hold a single long protein's group, an empty batch."""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_SEGMENT_MIN_ROWS", "1")
    monkeypatch.setenv("PLAAC_TRACK_SEGMENTS", nseg)
    monkeypatch.setenv("PLAAC_PIPE_SEGMENTS", nseg)
    monkeypatch.setenv("PLAAC_LATENCY_MODE", mode)
    monkeypatch.setenv("PLAAC_KB_LANE_MIN_GROUPS", "1")
    P0 = native.make_params()
    rng = np.random.default_rng(int(nseg) * 10 + int(mode))
    for lens in (np.concatenate([[9000, 2100, 1], rng.integers(1, 600, 700)]), rng.integers(20, 120, 130),
                 np.array([40000, 3, 70]), np.array([77])):
        rng.shuffle(lens)
        codes, offs = synth.residues(lens, np.array(P0.fg), np.array(P0.bg), rng, stop_fraction=0.1)
        want, wtr = oracle.score_batch(oracle.build_params(), codes, offs, tracks=True, nthreads=8)
        with native.Context(P0) as c:
            got = c.score(codes, offs)
            trows, tr = c.score(codes, offs, tracks=True)
        what = "runs %s mode %s, %d proteins" % (nseg, mode, len(lens))
        assert_rows_equal(got, want, what)
        assert_rows_equal(trows, want, what + " (track mode)")
        assert_tracks_equal(tr, wtr, codes, offs, what)
    with native.Context(P0) as c:
        assert len(c.score(np.zeros(0, np.uint8), np.zeros(1, np.uint64))) == 0


def test_clock_probe_reports_the_shader_clock(native):
    """plaac_clock_probe (bench.py's roofline.shader_clock): an idle MI355X holds about 2.4 GHz"""
    with native.Context(native.make_params()) as c:
        mhz = c.clock_probe(3000)
    assert 800.0 < mhz < 3000.0, mhz


@pytest.mark.parametrize("lane", ["1", "4096"])
def test_overlapped_consecutive_calls_give_the_same_rows(native, oracle, monkeypatch, lane):
    """plaac_ctx_set_overlap: the planning and packing of a call run beside the last window kernels (refine, exact tier) of
    the call before it. Batches of different sizes and contents back to back on one context WITHOUT a wait in between
    (device-resident entry point), each into its own row buffer; afterwards every buffer must hold its batch's rows - also
    for a batch with exact-tier fallbacks followed by a batch with a protein of >= 65,535 residues (the `huge` word of the
    next call must not reach the previous call's tail) and by an empty one. Lane form and stream form of the filter tier."""
    import torch
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_KB_LANE_MIN_GROUPS", lane)
    P0 = native.make_params()
    rng = np.random.default_rng(31)
    a_codes, a_offs = _adversarial_batch(native)
    batches = [synth.make_batch(4, nprot=9000, seed=5, fg=np.array(P0.fg), bg=np.array(P0.bg), stop_fraction=0.05),
               (a_codes, a_offs),
               synth.residues(np.array([70000, 300, 20]), np.array(P0.fg), np.array(P0.bg), rng),
               synth.make_batch(4, nprot=700, seed=6, fg=np.array(P0.fg), bg=np.array(P0.bg)),
               synth.make_batch(4, nprot=12000, seed=7, fg=np.array(P0.fg), bg=np.array(P0.bg), stop_fraction=0.05)]
    want = [oracle.score_batch(oracle.build_params(), c, o, nthreads=8) for c, o in batches]
    dev = torch.device("cuda", 0)
    with native.Context(P0) as ctx:
        ctx.set_overlap(True)
        st = torch.cuda.Stream(dev)
        dc = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c, _ in batches]
        do = [torch.from_numpy(np.ascontiguousarray(o).view(np.int64)).to(dev) for _, o in batches]
        rows = [torch.zeros(len(o) - 1, native.ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _, o in batches]
        torch.cuda.synchronize(dev)
        for rep in range(3):  # (the third round starts behind the tail of the second)
            for k, (c, o) in enumerate(batches):
                ctx.score_device(dc[k].data_ptr(), do[k].data_ptr(), len(o) - 1, int(o[-1]), rows[k].data_ptr(), None,
                                 stream=st.cuda_stream)
        ctx.sync()
        torch.cuda.synchronize(dev)
        for k in range(len(batches)):
            got = rows[k].cpu().numpy().view(native.ROW_DTYPE).reshape(-1)
            assert_rows_equal(got, want[k], "batch %d (filter tier from %s wave-groups in lane form)" % (k, lane))


@pytest.mark.parametrize("overlap", [False, True])
def test_pipelined_host_scoring_two_batches_in_flight(native, oracle, overlap):
    """plaac_score_begin / plaac_score_end (what bin/plaac's workers call): batches of different kinds streamed through ONE
    context with two in flight - the upload of a batch and the download of the one before it on the copy stream, beside the
    kernels -, with and without overlapping device work; an empty batch and a chain-bound one in the middle; a third
    begin without an end is refused; every batch's rows equal the oracle's."""
    from plaac_amd import synth
    P0 = native.make_params()
    rng = np.random.default_rng(77)
    batches = [synth.make_batch(4, nprot=6000, seed=21, fg=np.array(P0.fg), bg=np.array(P0.bg), stop_fraction=0.05),
               synth.residues(np.array([30000, 5000, 300, 20]), np.array(P0.fg), np.array(P0.bg), rng),
               (np.zeros(0, np.uint8), np.zeros(1, np.uint64)),
               synth.make_batch(2, nprot=900, seed=22, fg=np.array(P0.fg), bg=np.array(P0.bg)),
               _adversarial_batch(native),
               synth.make_batch(4, nprot=11000, seed=23, fg=np.array(P0.fg), bg=np.array(P0.bg), stop_fraction=0.05)]
    want = [oracle.score_batch(oracle.build_params(), c, o, nthreads=8) for c, o in batches]
    with native.Context(P0) as ctx:
        ctx.set_overlap(overlap)
        for rep in range(2):
            got = list(ctx.score_stream(batches))
            assert len(got) == len(batches)
            for k, (g, w) in enumerate(zip(got, want)):
                assert_rows_equal(g, w, "pipelined batch %d (overlap %s, round %d)" % (k, overlap, rep))
        n0 = ctx.score_begin(*batches[0])
        n1 = ctx.score_begin(*batches[3])
        with pytest.raises(native.PlaacError):
            ctx.score_begin(*batches[3])
        assert_rows_equal(ctx.score_end(n0), want[0], "after a refused third begin")
        assert_rows_equal(ctx.score_end(n1), want[3], "second pending batch")
        with pytest.raises(native.PlaacError):
            ctx.score_end(1)
        assert_rows_equal(ctx.score(*batches[3]), want[3], "the synchronous entry point afterwards")
        # round 5: the reference's first pass (computeaafreq, plaac.java:1655-1666) folded into the scoring pass - the
        # batch's background counts come back with its rows (bin/plaac's single pass when alpha = 1)
        for k, (g, cnt) in enumerate(ctx.score_stream(batches, counting=True)):
            assert_rows_equal(g, want[k], "counting batch %d (overlap %s)" % (k, overlap))
            assert np.array_equal(cnt, oracle.histogram(*batches[k])), "counts of batch %d" % k
        n0 = ctx.score_begin(*batches[0])  # a batch begun without counting cannot be collected with counts
        with pytest.raises(native.PlaacError):
            ctx.score_end_counts(n0)


def test_overlapping_calls_of_every_kind_in_any_order(native, oracle):
    """plaac_ctx_set_overlap with the kinds of call mixed on one context, back to back without a wait: summary calls that are
    throughput-bound and chain-bound (the kernel forms and the streams change between calls), a track-mode call, a sweep -
    every transition between a call whose head runs aside and one whose head does not, and the re-use of each of the two sets
    of plan buffers by a call of another kind. Every call writes its own buffers; all are checked at the end."""
    import torch
    from plaac_amd import synth
    P0 = native.make_params()
    rng = np.random.default_rng(77)
    fg, bg = np.array(P0.fg), np.array(P0.bg)
    wide = synth.make_batch(4, nprot=7000, seed=11, fg=fg, bg=bg, stop_fraction=0.05)            # throughput-bound
    deep = synth.residues(np.concatenate([[30000, 2500], rng.integers(20, 300, 400)]), fg, bg, rng)  # chain-bound
    small = synth.make_batch(4, nprot=300, seed=12, fg=fg, bg=bg)
    sweep_pts = [native.make_params(alpha=a, corelength=c) for a in (1.0, 0.5) for c in (30, 60)]
    plan = [("sum", wide), ("sum", deep), ("trk", small), ("sum", wide), ("swp", small), ("sum", deep), ("sum", small),
            ("trk", deep), ("sum", wide), ("sum", wide), ("swp", wide), ("sum", deep)]
    dev = torch.device("cuda", 0)
    up = {}
    for _, (c, o) in plan:
        if id(c) not in up:
            up[id(c)] = (torch.from_numpy(np.ascontiguousarray(c)).to(dev),
                         torch.from_numpy(np.ascontiguousarray(o).view(np.int64)).to(dev))
    # the outputs exist (and torch's fill of them has completed) before the first call: a buffer handed to an overlapping call
    # must not be in use by other pending work
    pre = []
    for kind, (c, o) in plan:
        n, tot = len(o) - 1, int(o[-1])
        if kind == "swp":
            pre.append(([torch.zeros(n, native.ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in sweep_pts], None))
            continue
        trk = None
        if kind == "trk":
            trk = {k: torch.zeros(tot, dtype=torch.uint8, device=dev) for k in native.TRACK_U8}
            trk.update({k: torch.full((tot,), float("nan"), dtype=torch.float64, device=dev) for k in native.TRACK_F64})
        pre.append((torch.zeros(n, native.ROW_DTYPE.itemsize, dtype=torch.uint8, device=dev), trk))
    outs = []
    with native.Context(P0) as ctx:
        ctx.set_overlap(True)
        st = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        for (kind, (c, o)), (rws, trk) in zip(plan, pre):
            dc, do = up[id(c)]
            n, tot = len(o) - 1, int(o[-1])
            if kind == "swp":
                ctx.score_sweep_device(dc.data_ptr(), do.data_ptr(), n, tot, sweep_pts, [r.data_ptr() for r in rws],
                                       stream=st.cuda_stream)
                outs.append((kind, c, o, rws, None))
                continue
            ctx.score_device(dc.data_ptr(), do.data_ptr(), n, tot, rws.data_ptr(),
                             None if trk is None else {k: v.data_ptr() for k, v in trk.items()}, stream=st.cuda_stream)
            outs.append((kind, c, o, rws, trk))
        ctx.sync()
        torch.cuda.synchronize(dev)
    want_cache = {}
    for i, (kind, c, o, rws, trk) in enumerate(outs):
        what = "call %d (%s, %d proteins)" % (i, kind, len(o) - 1)
        if kind == "swp":
            for P, r in zip(sweep_pts, rws):
                want = oracle.score_batch(oracle.build_params(alpha=P.alpha, corelength=P.corelength), c, o, nthreads=8)
                assert_rows_equal(r.cpu().numpy().view(native.ROW_DTYPE).reshape(-1), want, what)
            continue
        if id(c) not in want_cache:
            want_cache[id(c)] = oracle.score_batch(oracle.build_params(), c, o, tracks=True, nthreads=8)
        want, wtr = want_cache[id(c)]
        assert_rows_equal(rws.cpu().numpy().view(native.ROW_DTYPE).reshape(-1), want, what)
        if trk is not None:
            assert_tracks_equal({k: v.cpu().numpy() for k, v in trk.items()}, wtr, c, o, what)


@pytest.mark.parametrize("mode", ["0", "1"])
def test_sweep_group_of_more_than_256_core_lengths(native, oracle, monkeypatch, mode):
    """the first member of a launch used to travel as a byte (Op::m0), so core lengths 256.. of one alpha were
    never written and the caller got uninitialised rows with PLAAC_OK. 300 core lengths of one alpha + 3 of another, in the
    throughput-bound (core list) and the chain-bound schedule; every point against its own oracle run."""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_LATENCY_MODE", mode)
    P = native.make_params()
    codes, offs = synth.make_batch(2, nprot=150, seed=9, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
    extra = native.pack(["QNQNQNQNQNYYGGSSQQNN" * k for k in (2, 9, 17)] + ["A", ""])
    codes = np.concatenate([codes, extra[0]])
    offs = np.concatenate([offs, offs[-1] + extra[1][1:]]).astype(np.uint64)
    cls = list(range(5, 305))
    points = [(1.0, c) for c in cls] + [(0.5, c) for c in (30, 60, 90)]
    with native.Context(P) as c:
        with c.upload(codes, offs) as batch:
            bg = batch.histogram().astype(np.float64)
            got = batch.sweep([native.make_params(alpha=a, corelength=cl, bgcounts=bg) for a, cl in points])
    assert len(got) == len(points)
    for (a, cl), rows in zip(points, got):
        want = oracle.score_batch(oracle.build_params(alpha=a, corelength=cl, bgcounts=bg), codes, offs, nthreads=8)
        assert_rows_equal(rows, want, "alpha=%s c=%d mode=%s" % (a, cl, mode))


def test_result_breaking_switches_change_nothing_in_the_release_library(native, oracle, monkeypatch):
    """PLAAC_DEBUG_SKIP / PLAAC_VIT_STOP in a host's environment: the library that ships does not read them"""
    from plaac_amd import synth
    monkeypatch.setenv("PLAAC_DEBUG_SKIP", "k_win,k_fwd,k_vit,k_tracksL,k_refine_centres,k_tracks20f,k_core_list,k_long")
    monkeypatch.setenv("PLAAC_DEBUG_SKIP_FROM", "0")
    monkeypatch.setenv("PLAAC_VIT_STOP", "1")
    codes, offs = synth.make_batch(2, nprot=900, seed=3, stop_fraction=0.1)
    want = oracle.score_batch(oracle.build_params(), codes, offs, nthreads=8)
    with native.Context(native.make_params()) as c:
        for _ in range(4):  # (the switch used to act from the third call on)
            rows = c.score(codes, offs)
            assert_rows_equal(rows, want)


@pytest.mark.parametrize("max_records,max_bytes", [(1 << 20, 1 << 30), (257, 1 << 30), (1 << 20, 4096), (1, 1 << 30)])
def test_fasta_text_parsed_on_the_device_equals_the_host_parser(native, oracle, tmp_path, max_records, max_bytes):
    """K1 (round 5): plaac_score_begin_text / _end_text - the host only finds the records, the device splits lines, stops at
    empty lines and encodes. Codes, offsets, names (with the reference's trimming rule across batch boundaries), rows and
    background counts must equal what the host parser (plaac_fasta_read, itself held against a Python restatement of
    fastareader in tests/test_host_io.py) + plaac_score give, for any batch cut."""
    from plaac_amd import hostio
    from conftest import quirky_fasta
    data = quirky_fasta()
    p = tmp_path / "q.fa"
    p.write_bytes(data)
    names, codes, offs = hostio.read_fasta(p)
    want_rows = oracle.score_batch(oracle.build_params(), codes, offs, nthreads=8)
    want_counts = oracle.histogram(codes, offs)
    got_names, got_codes, got_rows, lens = [], [], [], []
    counts = np.zeros(22, dtype=np.int64)
    prev_blank = 1
    with native.Context(native.make_params()) as ctx:
        for text, starts, trim in hostio.stream_fasta_text(p, max_records, max_bytes):
            rows, c, o, blank, cnt = ctx.score_text(text, starts, counting=True)
            # (the reader's own look at the batch's two ends - what lets any context format any batch - is the device's)
            assert trim.flags == (prev_blank, int(blank[-1]) if len(blank) else prev_blank)
            nm, prev_blank = trim(blank, prev_blank)
            got_names += nm
            got_codes.append(c)
            lens.append(np.diff(o.astype(np.int64)))
            got_rows.append(rows)
            counts += cnt
        # the host's way to a few residues without the copy of all codes: the device's extents + the text
        rng = np.random.default_rng(1)
        text, starts, trim = next(iter(hostio.stream_fasta_text(p, 1 << 20, 1 << 30)))
        rows, ext, o, blank = ctx.score_text(text, starts, want_codes=False)
        assert_rows_equal(rows, want_rows, "rows without the codes copied back")
        assert np.array_equal(o, offs)
        for i in rng.integers(0, len(names), 400):
            n = int(offs[i + 1] - offs[i])
            first = int(rng.integers(0, n + 2))
            cnt = int(rng.integers(0, 60))
            want = codes[int(offs[i]) + min(first, n):int(offs[i]) + min(first + cnt, n)]
            assert np.array_equal(hostio.text_codes(text, starts, ext, i, first, cnt), want), (i, first, cnt)
        # the counting pass alone (plaac_histogram_begin_text / _end_text): parsed on the device, counted, not scored
        hist, nres = np.zeros(22, dtype=np.int64), 0
        for text, starts, trim in hostio.stream_fasta_text(p, max_records, max_bytes):
            c2, r2 = ctx.histogram_text(text, starts)
            hist += c2
            nres += r2
        assert np.array_equal(hist, want_counts) and nres == len(codes)
    assert got_names == names
    assert np.array_equal(np.concatenate(lens), np.diff(offs.astype(np.int64)))
    assert np.array_equal(np.concatenate(got_codes), codes)
    assert_rows_equal(np.concatenate(got_rows), want_rows, "rows of the device-parsed batches")
    assert np.array_equal(counts, want_counts)


def test_summary_rows_formatted_on_the_device_equal_the_hosts(native, oracle, tmp_path):
    """Round 5: plaac_score_end_text_table - scoreallfastas' output lines (plaac.java:899-945) written by the device from the
    rows it has just scored, the codes it has just parsed and the names in the text. Byte for byte the host formatter's
    (plaac_format_summary_row, itself tested against a restatement of java.util.Formatter in tests/test_host_io.py) over the
    same records, for any batch cut and both name-trimming states, values on a %.3f tie included (a mean charge of exactly
    1/16; ratios of small integers sit on ties all the time); a batch with a record without a sequence is handed back to the host."""
    from plaac_amd import hostio, synth
    from conftest import quirky_fasta
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=4000, seed=17, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.2)
    letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
    recs = [b">prot%05d some words  \n" % i + letters[int(offs[i]):int(offs[i + 1])].tobytes() + b"\n" + (b"\n" if i % 7 == 0 else b"")
            for i in range(len(offs) - 1) if offs[i + 1] > offs[i]]
    p = tmp_path / "t.fa"
    p.write_bytes(b"".join(recs))
    names, hc, ho = hostio.read_fasta(p)
    want_rows = oracle.score_batch(oracle.build_params(), hc, ho, nthreads=8)
    want = [hostio.format_summary_row(want_rows[i], names[i], hc[int(ho[i]):int(ho[i + 1])], corelength=60, ww2=41) for i in range(len(names))]
    want_text = b"".join((w if isinstance(w, bytes) else w.encode()) + b"\n" for w in want if w)
    for max_records in (1 << 20, 333):
        got, prev = [], 1
        with native.Context(P) as ctx:
            for text, starts, trim in hostio.stream_fasta_text(p, max_records, 1 << 30):
                table, prev, counts = ctx.score_text_table(text, starts, 60, 41, prev)
                assert table is not None, "a batch of ordinary proteins needs no host formatting"
                got.append(table)
        assert b"".join(got) == want_text, "device-formatted table, batches of %d records" % max_records
    # handed back: a record without a sequence (the host prints a note for it); a protein built to put a value on a %.3f tie
    # (one K in 16 residues: a mean charge of exactly 62.5 thousandths) comes out like the host's
    tie = tmp_path / "tie.fa"
    tie.write_bytes(b">tie\n" + b"K" + b"A" * 15 + b"\n" + recs[0])
    nos = tmp_path / "noseq.fa"
    nos.write_bytes(recs[0] + b">nothing here\n" + recs[1])
    with native.Context(P) as ctx:
        for path in (tie, nos):
            text, starts, trim = next(iter(hostio.stream_fasta_text(path, 1 << 20, 1 << 30)))
            n2, hc2, ho2 = hostio.read_fasta(path)
            rows2 = oracle.score_batch(oracle.build_params(), hc2, ho2, nthreads=8)
            table, _, _ = ctx.score_text_table(text, starts, 60, 41, 1)
            if path is nos:
                assert table is None
            if table is None:
                rows, c, o, blank, cnt = ctx.score_text_end(len(starts) - 1, len(text))  # (the batch is still there for the host)
                assert np.array_equal(c, hc2) and np.array_equal(o, ho2)
                assert_rows_equal(rows, rows2, "rows of a handed-back batch")
            else:
                w2 = [hostio.format_summary_row(rows2[i], n2[i], hc2[int(ho2[i]):int(ho2[i + 1])]) for i in range(len(n2))]
                assert table == b"".join(w.encode() + b"\n" for w in w2 if w)


def test_text_batches_and_code_batches_share_the_two_slots(native, oracle, tmp_path):
    """A context's two pending slots taken by text batches (plaac_score_begin_text) and encoded batches (plaac_score_begin /
    _begin_counting) in any order, each collected by the end call of its kind - and refused by the others without losing
    the batch: rows, tables and counts of every batch as if it had been alone."""
    from plaac_amd import hostio, synth
    P = native.make_params()
    rng = np.random.default_rng(23)
    texts, coded = [], []
    for k in range(4):
        codes, offs = synth.make_batch(4, nprot=int(rng.integers(300, 3000)), seed=40 + k, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
        keep = [i for i in range(len(offs) - 1) if offs[i + 1] > offs[i]]
        letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
        p = tmp_path / ("t%d.fa" % k)
        p.write_bytes(b"".join(b">b%d_%d\n" % (k, i) + letters[int(offs[i]):int(offs[i + 1])].tobytes() + b"\n" for i in keep))
        text, starts, trim = next(iter(hostio.stream_fasta_text(p, 1 << 20, 1 << 30)))
        names, hc, ho = hostio.read_fasta(p)
        rows = oracle.score_batch(oracle.build_params(), hc, ho, nthreads=8)
        table = b"".join(hostio.format_summary_row(rows[i], names[i], hc[int(ho[i]):int(ho[i + 1])]).encode() + b"\n" for i in range(len(names)))
        texts.append((text, np.ascontiguousarray(starts, dtype=np.uint64), table, oracle.histogram(hc, ho), rows, hc, ho))
        coded.append((hc, ho, rows))
    import ctypes as C
    with native.Context(P) as ctx:
        L, h = ctx._L, ctx._h

        def begin_text(k):
            text, starts = texts[k][0], texts[k][1]
            ctx._check(L.plaac_score_begin_text(h, text, len(text), starts.ctypes.data, len(starts) - 1, 1))

        def end_table(k):
            size, needs, lastb = C.c_uint64(), C.c_int(), C.c_int()
            ctx._check(L.plaac_score_end_text_table_size(h, 60, 41, 1, C.byref(size), C.byref(needs), C.byref(lastb), None))
            assert not needs.value
            buf = C.create_string_buffer(max(int(size.value), 1))
            counts = np.zeros(22, dtype=np.int64)
            ctx._check(L.plaac_score_end_text_table(h, buf, size.value, counts.ctypes.data))
            assert buf.raw[:size.value] == texts[k][2] and np.array_equal(counts, texts[k][3]), "table / counts of text batch %d" % k

        for order in ("TcTc", "cTTc", "TTcc", "cTcT"):
            pending = []
            it_t, it_c = iter(range(4)), iter(range(4))
            for kind in order + order:
                if len(pending) == 2:
                    what, k = pending.pop(0)
                    if what == "T":
                        with pytest.raises(native.PlaacError):  # (an encoded-batch end on a text batch's turn: refused, nothing lost)
                            ctx._check(L.plaac_score_end_text_table(h, None, 0, None))
                        end_table(k)
                    else:
                        size, needs, lastb = C.c_uint64(), C.c_int(), C.c_int()
                        assert L.plaac_score_end_text_table_size(h, 60, 41, 1, C.byref(size), C.byref(needs), C.byref(lastb), None) != 0
                        assert_rows_equal(ctx.score_end(len(coded[k][1]) - 1), coded[k][2], "encoded batch %d between text batches" % k)
                k = next(it_t, None) if kind == "T" else next(it_c, None)
                if k is None:
                    it_t, it_c = iter(range(4)), iter(range(4))
                    k = next(it_t) if kind == "T" else next(it_c)
                if kind == "T":
                    begin_text(k)
                else:
                    ctx.score_begin(coded[k][0], coded[k][1])
                pending.append((kind, k))
            for what, k in pending:
                if what == "T":
                    end_table(k)
                else:
                    assert_rows_equal(ctx.score_end(len(coded[k][1]) - 1), coded[k][2], "encoded batch %d at the end" % k)


def test_text_batches_uploaded_ahead_by_another_thread(native, oracle, tmp_path):
    """plaac_text_upload on a second host thread beside the scoring calls of the same context (bin/plaac's uploader thread):
    batch k + 1 is uploaded and parsed while batch k is scored and collected; plaac_score_begin_uploaded takes it over. The
    tables and counts of every batch as if it had gone through plaac_score_begin_text."""
    import ctypes as C
    import threading
    from plaac_amd import hostio, synth
    P = native.make_params()
    batches = []
    for k in range(6):
        codes, offs = synth.make_batch(4, nprot=1500 + 700 * k, seed=70 + k, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
        keep = [i for i in range(len(offs) - 1) if offs[i + 1] > offs[i]]
        letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
        p = tmp_path / ("u%d.fa" % k)
        p.write_bytes(b"".join(b">u%d_%d\n" % (k, i) + letters[int(offs[i]):int(offs[i + 1])].tobytes() + b"\n" for i in keep))
        text, starts, trim = next(iter(hostio.stream_fasta_text(p, 1 << 20, 1 << 30)))
        names, hc, ho = hostio.read_fasta(p)
        rows = oracle.score_batch(oracle.build_params(), hc, ho, nthreads=8)
        table = b"".join(hostio.format_summary_row(rows[i], names[i], hc[int(ho[i]):int(ho[i + 1])]).encode() + b"\n" for i in range(len(names)))
        batches.append((text, np.ascontiguousarray(starts, dtype=np.uint64), table, oracle.histogram(hc, ho)))
    with native.Context(P) as ctx:
        L, h = ctx._L, ctx._h
        handed, errors = [None] * len(batches), []
        ready = [threading.Event() for _ in batches]

        def uploader():
            try:
                for k, (text, starts, _, _) in enumerate(batches):
                    handed[k] = ctx.text_upload(text, starts)
                    ready[k].set()
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                for ev in ready:
                    ev.set()

        th = threading.Thread(target=uploader)
        th.start()
        pending = []

        def collect(k):
            size, needs, lastb = C.c_uint64(), C.c_int(), C.c_int()
            ctx._check(L.plaac_score_end_text_table_size(h, 60, 41, 1, C.byref(size), C.byref(needs), C.byref(lastb), None))
            assert not needs.value
            buf = C.create_string_buffer(max(int(size.value), 1))
            counts = np.zeros(22, dtype=np.int64)
            ctx._check(L.plaac_score_end_text_table(h, buf, size.value, counts.ctypes.data))
            assert buf.raw[:size.value] == batches[k][2] and np.array_equal(counts, batches[k][3]), "uploaded batch %d" % k

        for k in range(len(batches)):
            ready[k].wait(60)
            assert not errors, errors
            ctx.score_begin_uploaded(handed[k], counting=True)
            pending.append(k)
            if len(pending) == 2:
                collect(pending.pop(0))
        while pending:
            collect(pending.pop(0))
        th.join()
        # an uploaded batch that is not scored goes back to the context
        tb = ctx.text_upload(batches[0][0], batches[0][1])
        L.plaac_text_batch_free(tb)


def test_per_residue_table_from_the_device_equals_the_hosts(native, oracle):
    """plaac_score_tracks_table (round 5, late): plotsomefastas' lines (plaac.java:635-645) written by the device from the track
    arrays it has just filled - byte for byte the host's plaac_format_track_rows over the tracks plaac_score brings back (those
    are held against the oracle elsewhere): labels with tabs and odd bytes, proteins with a trimmed stop, short and long ones,
    the closing line after every protein."""
    from plaac_amd import hostio, synth
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=700, seed=31, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.3)
    keep = [i for i in range(len(offs) - 1) if offs[i + 1] - offs[i] > 1]
    c2 = np.concatenate([codes[int(offs[i]):int(offs[i + 1])] for i in keep])
    o2 = np.zeros(len(keep) + 1, dtype=np.uint64)
    o2[1:] = np.cumsum([int(offs[i + 1] - offs[i]) for i in keep])
    ids = [str(k + 1) if k % 5 else "7%d" % k for k in range(len(keep))]
    names = [("prot_%d" % k) if k % 3 else ("sp|Q%d|name with blanks, [brackets] & more" % k) for k in range(len(keep))]
    labels = [(a + "\t" + b).encode() for a, b in zip(ids, names)]
    with native.Context(P) as ctx:
        rows, tr = ctx.score(c2, o2, tracks=True)
        want = b"".join(hostio.format_track_rows(tr, int(o2[k]), c2[int(o2[k]):int(o2[k + 1])], int(rows[k]["prot_len"]), ids[k], names[k]).encode()
                        for k in range(len(keep)) if rows[k]["prot_len"] > 0)
        table, rows2 = ctx.score_tracks_table(c2, o2, labels)
        assert table is not None
        assert_rows_equal(rows2, rows, "rows beside the table")
        assert table == want, "per-residue table from the device (%d / %d bytes)" % (len(table), len(want))
        assert table.count(b"#" * 56 + b"\n") == int((rows["prot_len"] > 0).sum())


def test_node_text_api_deals_batches_over_contexts_and_returns_the_table_in_file_order(native, oracle, tmp_path):
    """Round 6: plaac_node_text_* - FASTA text batches in file order through a node of two contexts (devices
    {0, 0}), the table collected oldest first: byte for byte the oracle's rows through the host formatter, whatever the batch
    cut (name trimming behind an empty line crosses batches and contexts); a batch the device will not vouch for (a record
    without a sequence) is refused by text_table and comes back through text_rows - or is given up with text_discard - and the
    node goes on; the counting pass (histogram_text) and the uploader-thread entry points; the per-residue table."""
    import threading
    from plaac_amd import hostio, synth
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=3000, seed=23, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.2)
    letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
    recs = [b">prot%05d some words  \n" % i + b"".join(letters[int(offs[i]) + k:min(int(offs[i]) + k + 60, int(offs[i + 1]))].tobytes() + b"\n"
                                                       for k in range(0, int(offs[i + 1] - offs[i]), 60)) + (b"\n" if i % 7 == 0 else b"")
            for i in range(len(offs) - 1) if offs[i + 1] > offs[i]]
    p = tmp_path / "t.fa"
    p.write_bytes(b"".join(recs))
    names, hc, ho = hostio.read_fasta(p)
    want_rows = oracle.score_batch(oracle.build_params(), hc, ho, nthreads=8)
    want_text = hostio.format_summary_rows(want_rows, [n.encode() if isinstance(n, str) else n for n in names], hc, ho.astype(np.int64))
    want_counts = oracle.histogram(hc, ho)
    with native.Node(P, devices=[0, 0]) as node:
        assert len(node) == 2
        for max_records in (1 << 20, 211):
            node.text_reset()
            got, counts, nres = [], np.zeros(22, dtype=np.int64), 0
            for text, starts, trim in hostio.stream_fasta_text(p, max_records, 1 << 30):
                if node.text_pending() == 4:  # two per context
                    with pytest.raises(native.PlaacError):
                        node.text_begin(text, starts, counting=True)
                    t, r, c = node.text_table(60, 41, counting=True)
                    got.append(t); counts += c; nres += r
                node.text_begin(text, starts, counting=True)
            while node.text_pending():
                t, r, c = node.text_table(60, 41, counting=True)
                got.append(t); counts += c; nres += r
            assert b"".join(got) == want_text, "node table, batches of %d records" % max_records
            assert np.array_equal(counts, want_counts) and nres == len(hc)
        # the counting pass by itself, four batches in flight
        node.text_reset()
        counts, nres, inflight = np.zeros(22, dtype=np.int64), 0, 0
        for text, starts, trim in hostio.stream_fasta_text(p, 500, 1 << 30):
            if inflight == 4:
                nres = node.histogram_text_end(counts, nres)
                inflight -= 1
            node.histogram_text_begin(text, starts)
            inflight += 1
        while inflight:
            nres = node.histogram_text_end(counts, nres)
            inflight -= 1
        assert np.array_equal(counts, want_counts) and nres == len(hc)
        # a batch the device hands back: refused as a table, collected as rows; the next one discarded; then business as usual
        nos = tmp_path / "noseq.fa"
        nos.write_bytes(recs[0] + b">nothing here\n" + recs[1])
        text, starts, trim = next(iter(hostio.stream_fasta_text(nos, 1 << 20, 1 << 30)))
        n2, hc2, ho2 = hostio.read_fasta(nos)
        rows2 = oracle.score_batch(oracle.build_params(), hc2, ho2, nthreads=4)
        node.text_reset()
        for _ in range(3):
            node.text_begin(text, starts)
        t, r = node.text_table()
        assert t is None and node.text_pending() == 3
        with pytest.raises(native.PlaacError):
            node._check(node._L.plaac_node_text_table(node._h, None, 0, None))  # (refused: not sized as a table)
        rows, c, o, blank, ext = node.text_rows(len(text))
        assert np.array_equal(c, hc2) and np.array_equal(o, ho2)
        assert_rows_equal(rows, rows2, "rows of a handed-back batch through the node")
        node.text_discard()
        node.text_discard()
        assert node.text_pending() == 0
        with pytest.raises(native.PlaacError):
            node.text_discard()
        # the uploader-thread entry points: uploads on another thread, begun and collected here
        node.text_reset()
        batches = list(hostio.stream_fasta_text(p, 400, 1 << 30))
        handed, ready, errors = [None] * len(batches), [threading.Event() for _ in batches], []

        def uploader():
            try:
                for k, (text, starts, _) in enumerate(batches):
                    handed[k] = node.text_upload(text, starts)
                    ready[k].set()
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                for ev in ready:
                    ev.set()

        th = threading.Thread(target=uploader)
        th.start()
        got = []
        for k in range(len(batches)):
            ready[k].wait(60)
            assert not errors, errors
            if node.text_pending() == 4:
                got.append(node.text_table()[0])
            node.text_begin_uploaded(handed[k])
        while node.text_pending():
            got.append(node.text_table()[0])
        th.join()
        assert b"".join(got) == want_text, "node table from batches uploaded by another thread"
        tb = node.text_upload(batches[0][0], batches[0][1])
        node._L.plaac_node_text_batch_free(tb)
        # plotsomefastas' table through the node = a context's
        sel = [i for i in range(40) if ho[i + 1] - ho[i] > 1]
        c3 = np.concatenate([hc[int(ho[i]):int(ho[i + 1])] for i in sel])
        o3 = np.zeros(len(sel) + 1, dtype=np.uint64)
        o3[1:] = np.cumsum([int(ho[i + 1] - ho[i]) for i in sel])
        labels = [b"%d\t" % (k + 1) + (names[i].encode() if isinstance(names[i], str) else names[i]) for k, i in enumerate(sel)]
        t_node, r_node = node.score_tracks_table(c3, o3, labels)
    with native.Context(P) as ctx:
        t_ctx, r_ctx = ctx.score_tracks_table(c3, o3, labels)
    assert t_node is not None and t_node == t_ctx
    assert_rows_equal(r_node, r_ctx, "rows beside the node's per-residue table")


def test_value_tolerance_mode_moves_only_the_five_floats_at_the_papa_centre(native, oracle, monkeypatch):
    """Round 6: plaac_ctx_set_value_tolerance (off by default). The north star's bar for floats is 1e-6; with the switch on,
    the five floats reported at the PAPA centre come from first-level sums that slide over six neighbouring positions instead
    of 41 fixed-order taps each. TOLERANCE: |got - want| <= 1e-9 * max(1, |want|) for papa_combo, papa_prop, papa_fi, papa_llr,
    papa_llr2 (measured: < 1e-12), the same NaN / infinity pattern - and EVERY other field of the row, every index and
    decision included, bit-identical to the oracle. Switched off again, the rows are bit-identical as before."""
    from plaac_amd import synth
    P = native.make_params()
    monkeypatch.setenv("PLAAC_KB_LANE_MIN_GROUPS", "1")  # (the lane form, which a batch takes from 262,144 records on, at 60,000)
    codes, offs = synth.make_batch(4, nprot=60000, seed=41, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
    want = oracle.score_batch(oracle.build_params(), codes, offs, nthreads=16)
    loose = ("papa_combo", "papa_prop", "papa_fi", "papa_llr", "papa_llr2")
    with native.Context(P) as ctx:
        ctx.set_value_tolerance(True)
        got = ctx.score(codes, offs)
        for name in want.dtype.names:
            g, w = got[name], want[name]
            if name in loose:
                assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.isinf(g), np.isinf(w)), name
                fin = np.isfinite(w)
                err = np.abs(g[fin] - w[fin]) / np.maximum(1.0, np.abs(w[fin]))
                assert err.size and err.max() <= 1e-9, (name, err.max())
            elif g.dtype.kind == "f":
                assert np.array_equal(g.view(np.uint64), w.view(np.uint64)), name
            else:
                assert np.array_equal(g, w), name
        worst = max(float(np.nanmax(np.abs(got[n][np.isfinite(want[n])] - want[n][np.isfinite(want[n])]))) for n in loose)
        assert worst < 1e-11, worst  # (what the sliding sums actually cost; the contract above is 1e-9)
        assert (got["papa_combo"] != want["papa_combo"]).sum() > 1000  # (the switch did something: the sliding form is in use)
        ctx.set_value_tolerance(False)
        assert_rows_equal(ctx.score(codes, offs), want, "value tolerance switched off again")
