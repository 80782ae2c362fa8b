"""CPU tests of the oracle itself: pinned against the reference's known answers, cross-checked against
its pure-Python twin, and property-tested (the two disabled debug checks of plaac.java:772-791, 841-848)."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, runs_of_ones


def params_to_dict(P):
    def hmm(h):
        return dict(lt=[list(r) for r in h.lt], li=list(h.li), le=[list(r) for r in h.le], lf=list(h.lf))
    return dict(corelength=P.corelength, ww1=P.ww1, ww2=P.ww2, ww3=P.ww3, adjustprolines=bool(P.adjustprolines),
                cc=list(P.cc), llr=list(P.llr), lodpapa=list(P.lodpapa), hydro2=list(P.hydro2),
                charge=list(P.charge), loglut=list(P.loglut), hmm1=hmm(P.hmm1), hmm0=hmm(P.hmm0))


def test_kat28_viterbi_pins_the_oracle(oracle, kat28):
    """The reference's own 28 PrD annotations (cli/src/scer_fg_28.fasta), valid for fg = prd_freq_scer_04."""
    recs, rows = kat28
    fg04 = np.loadtxt(os.path.join(GOLDEN, "prd_freq_scer_04.txt"), usecols=0)
    assert np.array_equal(fg04, oracle.const_tables()["fg04"])
    P = oracle.build_params(fg=fg04)
    codes, offs = oracle.pack([s for _, s in recs])
    _, tr = oracle.score_batch(P, codes, offs, tracks=True)
    hits = 0
    for i, (gene, orf, s, e) in enumerate(rows):
        v = tr["vit"][int(offs[i]):int(offs[i + 1])]
        if recs[i][1].endswith("*"):
            v = v[:-1]
        hits += (s, e) in runs_of_ones(v)
    assert hits == 28


def test_kat28_is_specific_to_fg04(oracle, kat28):
    """With the default fg (prd_freq_scer_28) only 12 of the 28 coincide — guards against a
    KAT that would pass for any table."""
    recs, rows = kat28
    codes, offs = oracle.pack([s for _, s in recs])
    _, tr = oracle.score_batch(oracle.build_params(), codes, offs, tracks=True)
    hits = 0
    for i, (_, _, s, e) in enumerate(rows):
        n = int(offs[i + 1] - offs[i]) - recs[i][1].endswith("*")
        hits += (s, e) in runs_of_ones(tr["vit"][int(offs[i]):int(offs[i]) + n])
    assert hits == 12


def test_classic_prions_anchors(oracle, classic4):
    """SURVEY §8 C4 plausibility anchors (not reference output): Sup35p / Ure2p / Rnq1p / Mot3p."""
    codes, offs = oracle.pack([s for _, s in classic4])
    rows = oracle.score_batch(oracle.build_params(), codes, offs)
    one = lambda f, i: int(rows[f][i]) + 1
    assert [n for n, _ in classic4] == ["Sup35p", "Ure2p", "Rnq1p", "Mot3p"]
    assert (one("core_start", 0), one("core_end", 0), one("prd_start", 0), one("prd_end", 0)) == (5, 64, 1, 133)
    assert rows["prot_len"][0] == 685
    assert (one("core_start", 1), one("core_end", 1), one("prd_end", 1)) == (17, 76, 89)
    assert (one("core_start", 2), one("core_end", 2), one("prd_start", 2), one("prd_end", 2)) == (218, 277, 124, 405)
    assert (one("core_start", 3), one("core_end", 3)) == (98, 157)
    np.testing.assert_allclose(rows["core_score"], [51.215, 29.225, 47.459, 40.751], atol=1e-3)
    np.testing.assert_allclose(rows["hmm_all"][0], 81.82, atol=1e-2)
    np.testing.assert_allclose(rows["hmm_vit"][0], 79.60, atol=1e-2)
    np.testing.assert_allclose(rows["prd_score"][0], 89.773, atol=1e-3)


def test_table_setup_details(oracle):
    P = oracle.build_params()
    # the four normalisations really change bits: bg_freq_scer sums to 1.0001, fg28 to 0.99999
    t = oracle.const_tables()
    assert abs(t["bg_scer"].sum() - 1.0001) < 1e-12 and abs(t["fg28"].sum() - 0.99999) < 1e-12
    assert abs(sum(P.bg) - 1) < 1e-15 and abs(sum(P.fg) - 1) < 1e-15
    assert P.llr[0] == 0.0 and P.llr[21] == 0.0 and P.llr[12] > 1.0 and P.llr[14] > 1.0  # N, Q enriched
    assert P.hydro2[0] == 0.5 and P.hydro2[21] == 0.5  # X and * are 0.5, not 0
    assert P.lodpapa[0] == 0.0 and P.lodpapa[21] == 0.0
    assert P.loglut[0] == math.log(2.0) and len(P.loglut) == 4001
    assert list(P.hmm1.lf) == [0.0, 0.0] and list(P.hmm0.lf) == [0.0, 0.0]  # free end
    assert P.hmm0.lt[0][1] == -math.inf and P.hmm0.li[1] == -math.inf
    assert list(P.hmm0.le[0]) == list(P.hmm1.le[0])
    # alpha outside [0,1] falls back to 1.0 (:444-447)
    assert oracle.build_params(alpha=1.5).alpha == 1.0 and oracle.build_params(alpha=-0.1).alpha == 1.0
    # alpha = 0 with an organism background uses that background
    counts = np.arange(22, dtype=np.float64) + 5
    P0 = oracle.build_params(alpha=0.0, bgcounts=counts)
    c = counts.copy()
    c[0] = c[21] = 0
    np.testing.assert_allclose(np.array(P0.bgthis), c / c.sum(), rtol=1e-15)
    np.testing.assert_allclose(np.array(P0.bg)[1:21], (c / c.sum())[1:21], rtol=1e-4)


def test_encoding(oracle):
    s = "XACDEFGHIKLMNPQRSTVWY*acdefghiklmnpqrstvwyxBZUOJ- 1."
    got = oracle.encode(s).tolist()
    assert got[:22] == list(range(22))
    assert got[22:42] == list(range(1, 21))
    assert got[42:] == [0] * 10


def test_logeapeb(oracle):
    P = oracle.build_params()
    inf = math.inf
    assert oracle.logeapeb(P, -inf, -inf) == -inf
    assert oracle.logeapeb(P, -inf, -3.5) == -3.5 and oracle.logeapeb(P, 2.25, -inf) == 2.25
    assert oracle.logeapeb(P, 1.0, 1.0) == 1.0 + math.log(2)
    assert oracle.logeapeb(P, 0.0, -40.0) == 0.0 and 0.0 < oracle.logeapeb(P, 0.0, -30.0) < 1e-12
    # the table is log(1.0 + exp(-x)), not log1p: it is exactly 0 once 1 + e^-x rounds to 1 (x > ~36.7)
    assert P.loglut[3600] > 0.0 and all(v == 0.0 for v in P.loglut[3700:])
    rng = np.random.default_rng(0)
    for a, b in rng.normal(0, 10, (200, 2)):
        exact = np.logaddexp(a, b)
        assert abs(oracle.logeapeb(P, a, b) - exact) < 4e-6  # LUT interpolation error is part of the answer
        assert oracle.logeapeb(P, a, b) == oracle.logeapeb(P, b, a)


def test_hss2_equals_brute_force(oracle):
    """debug cross-check of plaac.java:772-791 as a property test, incl. exact ties (first window wins)"""
    rng = np.random.default_rng(1)
    for trial in range(300):
        n = int(rng.integers(1, 200))
        L = int(rng.integers(1, 90))
        if trial % 3 == 0:
            seq = rng.integers(0, 2, n).astype(np.float64)  # many exact ties
        elif trial % 3 == 1:
            seq = rng.normal(0, 2, n)
        else:
            seq = np.where(rng.random(n) < 0.3, -1000000.0, rng.normal(0, 2, n))
        a, b = oracle.hss2(seq, L, L), oracle.hss_brute(seq, L)
        assert a.tobytes() == b.tobytes(), (n, L)
        if L <= n:
            assert a[1] - a[0] + 1 == L
        else:
            assert (a[0], a[1], a[2]) == (-1, -2, -math.inf)


def test_python_twin_agrees_bit_for_bit(oracle):
    from oracle import plaac_oracle_py as twin
    from plaac_amd import synth
    rng = np.random.default_rng(2)
    for kw in (dict(), dict(corelength=20, ww1=11, ww2=21), dict(alpha=0.3, bgcounts=np.arange(1.0, 23.0)),
               dict(adjustprolines=False, ww1=40, ww2=6)):
        P = oracle.build_params(**kw)
        lens = np.array([1, 2, 3, 5, 19, 20, 21, 40, 41, 42, 59, 60, 61, 150, 400])
        codes, offs = synth.residues(lens, np.array(P.fg), np.array(P.bg), rng)
        codes[int(offs[13]) + 30:int(offs[13]) + 120] = rng.choice([12, 14, 16, 20, 13], 90)  # force a PrD + prolines
        rows, tr = oracle.score_batch(P, codes, offs, tracks=True)
        D = params_to_dict(P)
        for i in range(len(lens)):
            x = codes[int(offs[i]):int(offs[i + 1])].tolist()
            r, t = twin.score_protein(D, x)
            for k, v in r.items():
                w = rows[k][i]
                assert (v == w) or (isinstance(v, float) and math.isnan(v) and math.isnan(w)), (kw, i, k, v, w)
            for k, v in t.items():
                w = tr[k][int(offs[i]):int(offs[i + 1])]
                a = np.asarray(v, dtype=w.dtype)
                assert a.tobytes() == w.tobytes(), (kw, i, k)


def test_batch_semantics(oracle):
    """one trailing stop is trimmed for scoring (:758), empty / stop-only records are skipped (:762)"""
    P = oracle.build_params()
    codes, offs = oracle.pack(["MKVQQQNNN", "MKVQQQNNN*", "MKVQQQNNN**", "", "*"])
    rows = oracle.score_batch(P, codes, offs)
    assert rows["prot_len"].tolist() == [9, 9, 10, 0, 0]
    assert rows[0].tobytes() == rows[1].tobytes()
    assert rows["llr_start"][0] == -1 and rows["llr_end"][0] == -2 and rows["llr_score"][0] == -math.inf
    assert rows["core_start"][0] == -1 and math.isnan(rows["core_score"][0]) and rows["prd_score"][0] == 0.0
    assert rows["papa_cen"][0] == -1 and rows["papa_combo"][0] == -math.inf and math.isnan(rows["papa_prop"][0])
    # threads do not change anything
    from plaac_amd import synth
    c2, o2 = synth.make_batch(2, nprot=200, seed=4)
    assert oracle.score_batch(P, c2, o2, nthreads=1).tobytes() == oracle.score_batch(P, c2, o2, nthreads=4).tobytes()


def test_histogram_rules(oracle):
    """isvalidprotein (:1732-1739): X/* strictly inside invalidate, a trailing X invalidates, position 0 is free"""
    h = lambda s: oracle.histogram(*oracle.pack(s))
    assert h(["MKV*"]).sum() == 4 and h(["MKV*"])[21] == 1
    assert h(["MKVX"]).sum() == 0 and h(["XMKV"]).sum() == 4 and h(["MK*V"]).sum() == 0
    assert h(["MKV**"]).sum() == 0 and h(["M"]).sum() == 1 and h(["X"]).sum() == 0 and h(["*"]).sum() == 1
    assert h(["", "MKV"]).sum() == 3
    assert h(["mkv", "MKV"])[11] == 2


def test_output_invariants_on_synthetic_proteome(oracle):
    from plaac_amd import synth
    P = oracle.build_params()
    codes, offs = synth.make_batch(2, nprot=600, seed=8)
    rows, tr = oracle.score_batch(P, codes, offs, tracks=True, nthreads=4)
    has = rows["core_start"] >= 0
    assert has.sum() > 5
    assert np.array_equal(has, rows["vit_maxrun"] >= P.corelength)  # debug checks :841-848
    s = tr["post0"] + tr["post1"]
    assert np.all(np.abs(s - 1) < 1e-3) and np.any(s != 1.0)  # LUT noise: pairs need not sum to exactly 1
    # window tracks against a naive numpy evaluation (tolerance: different summation order)
    i = int(np.argmax(rows["prot_len"]))
    x = codes[int(offs[i]):int(offs[i + 1])]
    n, w = len(x), 20
    hy = np.array(P.hydro2)[x]
    naive = np.array([hy[max(0, k - w):min(n, k + w + 1)].mean() for k in range(n)])
    np.testing.assert_allclose(tr["hydro"][int(offs[i]):int(offs[i + 1])], naive, rtol=1e-12)
    fi = tr["fi"][int(offs[i]):int(offs[i + 1])]
    wt = 1.0 + np.minimum(np.arange(n), w) + np.minimum(n - 1 - np.arange(n), w)
    k = n // 2
    np.testing.assert_allclose(tr["fix2"][int(offs[i]) + k],
                               (wt[k - w:k + w + 1] * fi[k - w:k + w + 1]).sum() / wt[k - w:k + w + 1].sum(),
                               rtol=1e-12)
    assert np.all(np.isnan(tr["fix2"][int(offs[i]):int(offs[i]) + w]))


def _fixture_rows(key):
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "oracle_rows.json")) as f:
        return json.load(f)[key]


def _same_row(got, want, what):
    for f in got.dtype.names:
        if got.dtype[f].kind == "f":
            w = float.fromhex(want[f]) if want[f] != "nan" else float("nan")
            g = float(got[f])
            assert (g != g and w != w) or g.hex() == w.hex(), "%s %s: %r vs %r" % (what, f, g, w)
        else:
            assert int(got[f]) == want[f], "%s %s" % (what, f)


@pytest.mark.parametrize("key,fasta,kw", [
    ("classic4_default", "four_classic_prions.fasta", {}),
    ("classic4_c40_alpha0", "four_classic_prions.fasta",
     dict(corelength=40, alpha=0.0, bgcounts=np.arange(22, dtype=np.float64) + 5.0)),
    ("kat28_fg04", "kat28.fasta", "fg04"),
])
def test_oracle_reproduces_its_committed_rows(oracle, key, fasta, kw):
    """Regression anchor of the oracle itself (tests/golden/oracle_rows.json, generator beside it): every field of
    every row, floats bit for bit. Guards the checker against accidental edits; it is not reference output."""
    from conftest import GOLDEN, read_fasta_simple
    if kw == "fg04":
        kw = dict(fg=np.loadtxt(os.path.join(GOLDEN, "prd_freq_scer_04.txt"), usecols=0))
    recs = read_fasta_simple(os.path.join(GOLDEN, fasta))
    enc = [oracle.encode(s) for _, s in recs]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    rows = oracle.score_batch(oracle.build_params(**kw), np.concatenate(enc), offs, nthreads=2)
    want = _fixture_rows(key)
    assert [w["name"] for w in want] == [n for n, _ in recs]
    for r, w in zip(rows, want):
        _same_row(r, w, key + " " + w["name"])


def _compare_with_twin(oracle, P, codes, offs, what):
    from oracle import plaac_oracle_py as twin
    rows, tr = oracle.score_batch(P, codes, offs, tracks=True)
    D = params_to_dict(P)
    for i in range(len(offs) - 1):
        x = codes[int(offs[i]):int(offs[i + 1])].tolist()
        if x and x[-1] == 21:
            x = x[:-1]  # the batch entry point drops one trailing stop (:758); the twin scores one trimmed protein
        if not x:
            assert rows["prot_len"][i] == 0  # skipped record (:762)
            continue
        r, t = twin.score_protein(D, x)
        for k, v in r.items():
            w = rows[k][i]
            assert (v == w) or (isinstance(v, float) and math.isnan(v) and math.isnan(w)), (what, i, k, v, w)
        n = len(x)
        for k, v in t.items():
            w = tr[k][int(offs[i]):int(offs[i]) + n]
            a = np.asarray(v[:n], dtype=w.dtype)
            assert a.tobytes() == w.tobytes(), (what, i, k)


def test_differential_c_oracle_vs_python_twin_hypothesis(oracle):
    """Unpinned floats (no reference golden values exist: SURVEY 8(c) C3): two independently written restatements of
    plaac.java must agree bit for bit over randomly drawn parameters (alpha, core length, three window sizes, fg / bg
    tables, proline rule) and sequences rich in X / stop / proline runs / homopolymers / very short lengths."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st, HealthCheck

    residue = st.sampled_from(list(range(22)) + [13, 13, 13, 12, 14, 12, 14, 0, 21])
    run = st.tuples(residue, st.integers(1, 30)).map(lambda cr: [cr[0]] * cr[1])
    seq = st.lists(st.one_of(residue.map(lambda c: [c]), run), min_size=1, max_size=40).map(
        lambda parts: [c for p in parts for c in p])
    table = st.lists(st.floats(0.0, 1000.0, allow_nan=False), min_size=22, max_size=22)

    @settings(max_examples=400, deadline=None, suppress_health_check=list(HealthCheck))
    @given(seqs=st.lists(seq, min_size=1, max_size=6), alpha=st.floats(-0.5, 1.5), c=st.integers(1, 90),
           ww1=st.integers(1, 61), ww2=st.integers(1, 61), fg=st.one_of(st.none(), table), bg=st.one_of(st.none(), table),
           adjust=st.booleans(), stop=st.booleans())
    def run_case(seqs, alpha, c, ww1, ww2, fg, bg, adjust, stop):
        kw = dict(alpha=alpha, corelength=c, ww1=ww1, ww2=ww2, adjustprolines=adjust)
        if fg is not None and sum(fg[1:21]) > 0:
            kw["fg"] = np.array(fg)
        if bg is not None:
            kw["bgcounts"] = np.array(bg)
        P = oracle.build_params(**kw)
        if not np.all(np.isfinite(np.array(P.llr))):  # a zero frequency: log(0) poisons both sides alike, nothing to learn
            return
        if stop:
            seqs = [s + [21] for s in seqs]
        codes = np.array([c_ for s in seqs for c_ in s], dtype=np.uint8)
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        _compare_with_twin(oracle, P, codes, offs, kw)

    run_case()
