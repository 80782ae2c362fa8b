"""GPU tests of the command-line host (bin/plaac): the plaac.jar flag surface end to end.
Expected text = the oracle's rows pushed through the same C++ formatter (formatting itself is covered on
the CPU by test_host_io.py), plus literal spot checks of BASELINE config 1 (Sup35p, -c 60 -a 1.0)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read_fasta_simple

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "bin", "plaac")
FA4 = os.path.join(GOLDEN, "four_classic_prions.fasta")
KAT = os.path.join(GOLDEN, "kat28.fasta")


def run(*args):
    assert os.path.exists(BIN), "bin/plaac missing: run make / __graft_entry__.build()"
    r = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout.split("\n")


def expected_rows(oracle, io, path, **kw):
    names, codes, offs = io.read_fasta(path)
    rows = oracle.score_batch(oracle.build_params(**kw), codes, offs)
    out = []
    for i, nm in enumerate(names):
        line = io.format_summary_row(rows[i], nm, codes[int(offs[i]):int(offs[i + 1])],
                                     corelength=kw.get("corelength", 60), ww2=kw.get("ww2", 41))
        if line:
            out.append(line)
    return out, oracle.histogram(codes, offs)


@pytest.fixture(scope="module")
def io(native):
    from plaac_amd import hostio
    return hostio


def test_config1_sup35_summary(oracle, io):
    lines = run("-i", FA4, "-c", 60, "-a", "1.0")
    assert lines[0].startswith("####") and lines[1].startswith("## alpha=1.0; corelength=60; ww1=41; ww2=41; ww3=41;")
    body = [l for l in lines if l and not l.startswith("#")]
    assert body[0] == io.summary_header()
    want, counts = expected_rows(oracle, io, FA4, bgcounts=None)
    # alpha = 1: the input background does not influence the scores, only the '## bg_input' line
    assert body[1:] == want
    sup = dict(zip(body[0].split("\t"), body[1].split("\t")))
    assert (sup["SEQid"], sup["COREstart"], sup["COREend"], sup["PRDstart"], sup["PRDend"], sup["PROTlen"]) == \
        ("Sup35p", "5", "64", "1", "133", "685")
    assert sup["COREscore"] == "51.215" and sup["HMMall"] == "81.820"
    bg_line = [l for l in lines if l.startswith("## bg_input")][0]
    c = counts.astype(float)
    c[0] = c[21] = 0
    assert "A=%.5f;" % (c[1] / c.sum()) in bg_line


def test_flags_alpha_core_windows_and_skipping_header(oracle, io):
    _, counts = expected_rows(oracle, io, KAT)
    kw = dict(alpha=0.5, corelength=30, ww1=21, ww2=31, bgcounts=counts.astype(float))
    lines = run("-i", KAT, "-a", 0.5, "-c", 30, "-w", 21, "-W", 31, "-s")
    assert lines[0] == io.summary_header()  # -s: no parameter block
    want, _ = expected_rows(oracle, io, KAT, **kw)
    assert [l for l in lines[1:] if l] == want


def test_background_tools_roundtrip(oracle, io, tmp_path):
    dump = run("-b", KAT)  # -b without -i: print counts and exit
    _, counts = expected_rows(oracle, io, KAT)
    assert [l for l in dump if l] == ["%.6f # %s" % (counts[i], a) for i, a in enumerate("XACDEFGHIKLMNPQRSTVWY*")]
    bgf = tmp_path / "bg.txt"
    bgf.write_text("\n".join(dump))
    a = run("-i", FA4, "-B", bgf, "-a", 0.0, "-s")
    b = run("-i", FA4, "-b", KAT, "-a", 0.0, "-s")
    assert a == b
    want, _ = expected_rows(oracle, io, FA4, alpha=0.0, bgcounts=counts.astype(float))
    assert [l for l in a[1:] if l] == want


def test_config3_the_way_baseline_words_it_reference_written_human_background(oracle, io, native, tmp_path):
    """BASELINE config 3: "-a 0.5 mixed background" = the web app's real invocation `-i <fasta> -c 60 -a 0.5 -B
    bg_freqs_HUMAN.txt` (web/lib/server.rb:152-155) with the background file the reference itself wrote (plaac.jar -b,
    tests/golden/bg_freqs/, make_bg_freqs.py). A human-proteome-shaped synthetic FASTA (incl. a 34,350-residue record)
    through bin/plaac against the oracle-formatted table; the parameter block echoes the file's frequencies."""
    from plaac_amd import synth
    bgfile = os.path.join(GOLDEN, "bg_freqs", "bg_freqs_HUMAN.txt")
    bgcounts = io.read_aa_params(bgfile)[0]
    P = native.make_params(alpha=0.5, bgcounts=bgcounts)
    codes, offs = synth.make_batch(3, nprot=2000, seed=33, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
    fa = tmp_path / "human_like.fa"
    _write_fasta(fa, codes, offs)
    lines = run("-i", fa, "-c", 60, "-a", "0.5", "-B", bgfile)
    body = [l for l in lines if l and not l.startswith("#")]
    want, _ = expected_rows(oracle, io, str(fa), alpha=0.5, bgcounts=bgcounts)
    assert body[0] == io.summary_header() and body[1:] == want
    head = [l for l in lines if l.startswith("## ")]
    assert head[0].startswith("## alpha=0.5; corelength=60;")
    c = bgcounts.copy()
    c[0] = c[21] = 0
    bg_line = [l for l in head if l.startswith("## bg_input")][0]
    assert "A=%.5f;" % (c[1] / c.sum()) in bg_line and "L=%.5f;" % (c[10] / c.sum()) in bg_line
    # the same background through -b <fasta> would need the proteome; the file IS what the reference counted from it


def test_fg_override_reproduces_kat28(io, kat28):
    """-F prd_freq_scer_04.txt (fixed: the reference reads the -B file here) -> the 28 annotated PrDs"""
    lines = run("-i", KAT, "-F", os.path.join(GOLDEN, "prd_freq_scer_04.txt"), "-p", "all", "-s")
    assert lines[0] == io.tracks_header()
    recs, rows = kat28
    vit = {}
    for l in lines[1:]:
        if l and not l.startswith("#"):
            f = l.split("\t")
            vit.setdefault(f[1], []).append(int(f[4]))
    from conftest import runs_of_ones
    for gene, orf, s, e in rows:
        assert (s, e) in runs_of_ones(vit[orf]), gene


def test_plot_list_and_invalid_alpha(io, tmp_path):
    lst = tmp_path / "list.txt"
    lst.write_text("Rnq1p\tRNQ1 display\nSup35p\nnot-there\n")
    lines = run("-i", FA4, "-p", lst, "-a", 3, "-zzz", "x")
    # the reference's loop stops before a trailing token that is not -d/-s, so only -zzz is reported (:337-353)
    assert "# skipping unknown option -zzz" in lines and "# skipping unknown option x" not in lines
    assert "# warning: invalid alpha; using alpha = 1.0" in lines
    body = [l.split("\t") for l in lines if l and not l.startswith("#")]
    assert body[0] == io.tracks_header().split("\t")
    firsts = [f for f in body[1:] if f[2] == "1"]
    # file order, ORDER = line number in the list, SEQid = display name when given
    assert [(f[0], f[1]) for f in firsts] == [("2", "Sup35p"), ("1", "RNQ1 display")]
    assert sum(1 for l in lines if l == "#" * 56) == 2


def test_usage_and_missing_input():
    out = run()
    assert any("USAGE" in l for l in out)
    out = run("-i", "/nonexistent.fa", "-s")
    assert "# Couldn't open /nonexistent.fa" in out


def run_env(env, *args):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    return r.stdout


def _write_fasta(path, codes, offs, stop_every=0):
    letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
    with open(path, "wb") as fh:
        for i in range(len(offs) - 1):
            seq = letters[int(offs[i]):int(offs[i + 1])].tobytes()
            fh.write(b">rec%05d some description\n" % i)
            for k in range(0, len(seq), 60):  # wrapped lines, as real FASTA files are
                fh.write(seq[k:k + 60] + b"\n")


def test_streamed_pipeline_is_independent_of_batching_and_contexts(native, tmp_path):
    """The CLI streams the input in batches over every scoring context: the table must not depend on the batch size,
    on the number of contexts, on the device list, or on whether the scoring pass replays the batches the background
    pass kept (small inputs) or reads the file again (large inputs). Reference order: plaac.java:755 (file order)."""
    from plaac_amd import synth
    P = native.make_params()
    codes, offs = synth.make_batch(3, nprot=1500, seed=12, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.1)
    fa = tmp_path / "in.fa"
    _write_fasta(fa, codes, offs)
    args = ("-i", fa, "-a", 0.5, "-c", 40)
    base = run_env({"PLAAC_DEVICES": "0"}, *args)  # one context, one batch
    assert base.count(b"\n") > 1500
    envs = ({"PLAAC_BATCH_RECORDS": 37}, {"PLAAC_BATCH_BYTES": 20000, "PLAAC_DEVICES": "0,0,0"},
            {"PLAAC_BATCH_RECORDS": 100, "PLAAC_KEEP_BYTES": 1}, {"PLAAC_BATCH_RECORDS": 1, "PLAAC_DEVICES": "0,0"},
            {"PLAAC_BATCH_RECORDS": 100, "PLAAC_KEEP_BYTES": 100000, "PLAAC_CTX_PER_DEVICE": 3},
            {"PLAAC_FAST_EXIT": 1, "PLAAC_BATCH_RECORDS": 500}, {"PLAAC_TEARDOWN": 1, "PLAAC_BATCH_RECORDS": 300})
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as pool:  # (four processes on the card at a time: a run is mostly HIP start-up)
        for env, out in zip(envs, pool.map(lambda e: run_env(e, *args), envs)):
            assert out == base, env
    # track mode streams too (8 MiB batches by default): every record, then a list
    tbase = run_env({"PLAAC_DEVICES": "0"}, "-i", fa, "-p", "all", "-s")
    assert run_env({"PLAAC_BATCH_RECORDS": 64, "PLAAC_DEVICES": "0,0"}, "-i", fa, "-p", "all", "-s") == tbase
    lst = tmp_path / "list.txt"
    lst.write_text("rec01400 some description\trenamed\nrec00003 some description\n")
    t1 = run_env({"PLAAC_DEVICES": "0"}, "-i", fa, "-p", lst, "-s")
    assert run_env({"PLAAC_BATCH_RECORDS": 10}, "-i", fa, "-p", lst, "-s") == t1
    order = [l.split(b"\t")[:2] for l in t1.split(b"\n") if l and not l.startswith(b"#") and l.split(b"\t")[2:3] == [b"1"]]
    assert order == [[b"2", b"rec00003 some description"], [b"1", b"renamed"]]


def test_background_dump_streams_and_missing_background_is_reported(native, tmp_path):
    from plaac_amd import synth
    P = native.make_params()
    codes, offs = synth.make_batch(2, nprot=700, seed=3, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.3)
    fa = tmp_path / "bg.fa"
    _write_fasta(fa, codes, offs)
    a = run_env({}, "-b", fa)
    assert run_env({"PLAAC_BATCH_RECORDS": 33, "PLAAC_DEVICES": "0,0"}, "-b", fa) == a
    out = run_env({}, "-b", "/nonexistent_bg.fa").decode()
    assert "# Couldn't open /nonexistent_bg.fa" in out and "0.000000 # A" in out


def test_track_table_reads_like_the_R_consumer(io, tmp_path):
    """The consumer of `-p` output is cli/R/plaac_plot_util.r: it drops every line that starts with '#'
    (:371-372), reads the rest with read.table(header=TRUE, sep="\\t", quote="", comment.char="") (:373), keys
    on column NAMES - ORDER (:401, :410), AA (:124), VIT / MAP (:186-187), the columns matching ^HMM (:115-117), the
    tracks CHARGE HYDRO FI PLAAC PAPA FIx2 PLAACx2 PAPAx2 (:240-262) - and groups rows by ORDER. Parsed here the same way."""
    lst = tmp_path / "list.txt"
    lst.write_text("Rnq1p\tRNQ1 #1 display\nSup35p\n")  # a '#' inside a display name must survive (quote="", :369)
    lines = run("-i", FA4, "-p", lst)
    assert any(l.startswith("##") for l in lines)          # the parameter block is there and is dropped below
    kept = [l for l in lines if l and not l.startswith("#")]  # datRaw[!grepl("^#", datRaw)]
    header = kept[0].split("\t")
    # read.table(header=TRUE) passes names through make.names: '-' becomes '.'
    rnames = [h.replace("-", ".") for h in header]
    for need in ("ORDER", "SEQid", "AANUM", "AA", "VIT", "MAP", "CHARGE", "HYDRO", "FI", "PLAAC", "PAPA", "FIx2",
                 "PLAACx2", "PAPAx2"):
        assert need in rnames, need
    hmm = [h for h in rnames if h.startswith("HMM")]           # grep("^HMM", colnames(dat))
    assert [h.replace("HMM.", "", 1) for h in hmm] == ["background", "PrD.like"]
    rows = [l.split("\t") for l in kept[1:]]
    assert all(len(r) == len(header) for r in rows)            # read.table refuses ragged lines
    col = {h: [r[i] for r in rows] for i, h in enumerate(rnames)}

    def rnum(x):  # type.convert: NaN / NA / Inf spellings R accepts
        return float("nan") if x in ("NaN", "NA") else float(x)
    for name in ("CHARGE", "HYDRO", "FI", "PLAAC", "PAPA", "FIx2", "PLAACx2", "PAPAx2") + tuple(hmm):
        vals = np.array([rnum(x) for x in col[name]])          # every cell numeric -> the column is numeric in R
        assert np.isfinite(vals).sum() > 0
    assert set(col["VIT"]) <= {"0", "1"} and set(col["MAP"]) <= {"0", "1"}
    order = [int(x) for x in col["ORDER"]]
    groups = {}
    for o_, nm, an, aa in zip(order, col["SEQid"], col["AANUM"], col["AA"]):
        groups.setdefault(o_, []).append((nm, int(an), aa))
    assert sorted(groups) == [1, 2]                            # sort(unique(datAll$ORDER)): plot order = list order
    assert {g[0][0] for g in groups.values()} == {"RNQ1 #1 display", "Sup35p"}
    seqs = dict(read_fasta_simple(FA4))
    for o_, nm in ((1, "Rnq1p"), (2, "Sup35p")):
        g = groups[o_]
        assert [a for _, a, _ in g] == list(range(1, len(g) + 1))        # AANUM runs 1..n inside a group
        assert "".join(aa for _, _, aa in g) == seqs[nm].rstrip("*")
        post = np.array([[rnum(r[rnames.index(h)]) for h in hmm] for r in rows if int(r[0]) == o_])
        assert np.all(np.abs(post.sum(axis=1) - 1.0) < 1e-3)          # posteriors of the two states


def test_single_pass_writes_the_same_bytes_as_the_two_passes(native, tmp_path):
    """bin/plaac folds the reference's counting pass (plaac.java:377-384) into its scoring pass when alpha = 1 (round 5:
    plaac_score_begin_counting; the parameter block waits for the final counts, everything behind it is held back): stdout
    must be byte-identical to the two-pass run (PLAAC_SINGLE_PASS=0) for every way of asking - plain, with the column notes,
    an invalid alpha (replaced by 1.0 with the warning line), -b naming the input itself, without the parameter block, with
    the dot export, many small batches over two contexts, an empty file - and alpha < 1 / -B / -p keep the two passes."""
    from plaac_amd import synth
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=3000, seed=90, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.2)
    fa = tmp_path / "in.fa"
    _write_fasta(fa, codes, offs)
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    dot = tmp_path / "hmm.dot"
    # K1 (round 5): the single pass parses on the device (plaac_score_begin_text) and, with one context, formats the rows there
    # too (plaac_score_end_text_table; PLAAC_DEVICE_FORMAT=0: the host's formatter); a file with every line-end / blank-line /
    # name-trimming quirk of fastareader (:4302-4375), cut into batches at any record, must come out the same
    from conftest import quirky_fasta
    quirks = tmp_path / "quirks.fa"
    quirks.write_bytes(quirky_fasta(seed=8, nrec=1200))
    cases = [["-i", fa], ["-i", fa, "-d"], ["-i", fa, "-a", "3"], ["-i", fa, "-b", fa], ["-i", fa, "-s"], ["-i", fa, "-h", dot],
             ["-i", fa, "-c", "30", "-W", "21"], ["-i", empty], ["-i", FA4, "-a", "1.0"], ["-i", quirks], ["-i", fa, "-a", "0.5"],
             ["-i", fa, "-B", os.path.join(GOLDEN, "bg_freqs", "bg_freqs_YEAST.txt")]]
    from concurrent.futures import ThreadPoolExecutor
    for ci, args in enumerate(cases):
        envs = ({"PLAAC_SINGLE_PASS": "0", "PLAAC_HUGE_PAGES": "0"}, {}, {"PLAAC_BATCH_RECORDS": "257", "PLAAC_DEVICES": "0,0"},
                {"PLAAC_DEVICE_PARSE": "0"}, {"PLAAC_DEVICE_FORMAT": "0"}, {"PLAAC_DEVICE_FORMAT": "0", "PLAAC_BATCH_RECORDS": "100"},
                {"PLAAC_BATCH_RECORDS": "1"} if args[1] is quirks else {"PLAAC_BATCH_BYTES": "4096"})
        # (every way of running it for the plain call, the column notes, the classic four, the quirky file and the two-pass
        #  alpha; for the other options the two-pass run, the default and the host's formatter: a run is mostly HIP start-up)
        if ci not in (0, 1, 8, 9, 10):
            envs = (envs[0], envs[1], envs[4])

        def run(env, args=args):
            return subprocess.run([BIN] + [str(a) for a in args], capture_output=True, timeout=300, env=dict(os.environ, PLAAC_TIMING="1", **env))

        # (the variants of a case side by side, four processes on the card at a time: a run is mostly HIP start-up;
        #  -h writes a file, so those go one by one)
        with ThreadPoolExecutor(max_workers=1 if "-h" in args else 4) as pool:
            rs = list(pool.map(run, envs))
        outs = []
        for r in rs:
            assert r.returncode == 0, r.stderr.decode(errors="replace")
            outs.append((r.stdout, r.stderr.decode(errors="replace")))
        assert all(o[0] == outs[0][0] for o in outs[1:]), "single pass differs from two passes for %s" % args
        single = "single pass" in outs[1][1]
        assert "single pass" not in outs[0][1]
        expect_single = not any(str(a) in ("0.5", "-B") for a in args)
        assert single == expect_single, (args, outs[1][1])
    # a single-pass table is still the oracle's (bg_input line from the counts of the whole input)
    lines = outs[1][0].decode().split("\n")
    assert any(l.startswith("## bg_input") for l in lines)


def test_single_pass_into_a_file_writes_the_table_in_place(native, tmp_path):
    """Single pass with stdout a plain file (round 5): the table is written where it belongs while the parameter block in front
    of it still waits for the counts - a placeholder of the block's length first, the block over it at the end. Byte-identical
    to the two-pass run: plain, with column notes, without the block, from an offset inside a file that already has content,
    appending (no placing then), an input without a single valid residue (the block prints NaN: shorter than its placeholder,
    the table moves), an input of one residue type (frequency 1.00000), an empty input."""
    from plaac_amd import synth
    from conftest import quirky_fasta
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=5000, seed=91, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.2)
    fa = tmp_path / "in.fa"
    _write_fasta(fa, codes, offs)
    allx = tmp_path / "allx.fa"
    allx.write_bytes(b"".join(b">x%d\n%s\n" % (i, b"X" * (50 + i) + (b"*" if i % 2 else b"")) for i in range(300)))
    polyq = tmp_path / "polyq.fa"
    polyq.write_bytes(b"".join(b">q%d\n%s\n" % (i, b"Q" * (80 + i)) for i in range(100)))
    quirks = tmp_path / "quirks.fa"
    quirks.write_bytes(quirky_fasta(seed=9, nrec=900))
    empty = tmp_path / "empty.fa"
    empty.write_text("")

    import itertools
    serial = itertools.count()

    def to_file(args, env, mode="wb", lead=b""):
        out = tmp_path / ("out%d.tsv" % next(serial))
        if mode == "ab":
            out.write_bytes(lead)
        with open(out, mode) as fh:
            if mode == "wb" and lead:
                fh.write(lead)
                fh.flush()
            r = subprocess.run([BIN] + [str(a) for a in args], stdout=fh, stderr=subprocess.PIPE, timeout=300,
                               env=dict(os.environ, PLAAC_TIMING="1", **env))
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        return out.read_bytes(), r.stderr.decode(errors="replace")

    for args in (["-i", fa], ["-i", fa, "-d"], ["-i", fa, "-s"], ["-i", allx], ["-i", allx, "-d"], ["-i", polyq], ["-i", quirks], ["-i", empty]):
        from concurrent.futures import ThreadPoolExecutor
        envs = ({}, {"PLAAC_PLACED_WRITE": "0"}, {"PLAAC_BATCH_RECORDS": "300"}, {"PLAAC_DEVICE_FORMAT": "0"})
        with ThreadPoolExecutor(max_workers=4) as pool:  # (four processes on the card at a time; every run its own output file)
            f_want = pool.submit(to_file, args, {"PLAAC_SINGLE_PASS": "0"})
            f_envs = [pool.submit(to_file, args, env) for env in envs]
            f_lead = pool.submit(to_file, args, {}, "wb", b"something the shell wrote before\n")
            f_app = pool.submit(to_file, args, {}, "ab", b"appended to\n")
            want, _ = f_want.result()
            for env, f in zip(envs, f_envs):
                got, err = f.result()
                assert "single pass" in err
                assert got == want, (args, env)
            assert f_lead.result()[0] == b"something the shell wrote before\n" + want, args
            assert f_app.result()[0] == b"appended to\n" + want, args
    assert b"NaN" in to_file(["-i", allx], {})[0]


def test_per_residue_table_from_the_device_is_the_hosts(native, tmp_path):
    """bin/plaac -p: the lines come from the device (plaac_score_tracks_table) unless PLAAC_DEVICE_FORMAT=0 - the same bytes either
    way, for -p all and for a list with display names and an order column, over files with stops, short records, records
    without a sequence, and small batches."""
    from plaac_amd import synth
    from conftest import quirky_fasta
    P = native.make_params()
    codes, offs = synth.make_batch(4, nprot=400, seed=93, fg=np.array(P.fg), bg=np.array(P.bg), stop_fraction=0.3)
    fa = tmp_path / "in.fa"
    _write_fasta(fa, codes, offs)
    quirks = tmp_path / "quirks.fa"
    quirks.write_bytes(quirky_fasta(seed=10, nrec=300))
    lst = tmp_path / "list.txt"
    lst.write_text("rec00007 some description\tShown as seven\nrec00003 some description\nnot-there\nrec00100 some description\tA hundred\n")
    for args in (["-i", fa, "-p", "all"], ["-i", fa, "-p", lst], ["-i", quirks, "-p", "all", "-s"], ["-i", fa, "-p", "all", "-a", "0.5"]):
        outs = []
        envs = ({"PLAAC_DEVICE_FORMAT": "0"}, {}, {"PLAAC_BATCH_BYTES": "20000"}, {"PLAAC_DEVICES": "0,0", "PLAAC_BATCH_RECORDS": "50"})
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=4) as pool:  # (four processes on the card at a time: a run is mostly HIP start-up)
            rs = list(pool.map(lambda env: subprocess.run([BIN] + [str(a) for a in args], capture_output=True, timeout=300,
                                                          env=dict(os.environ, **env)), envs))
        for r in rs:
            assert r.returncode == 0, r.stderr.decode(errors="replace")
            outs.append(r.stdout)
        assert all(o == outs[0] for o in outs[1:]), args
        assert outs[0].count(b"#" * 56 + b"\n") >= 1
