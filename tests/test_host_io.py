"""CPU tests of the host text I/O in libplaac_native.so (include/plaac_host.h): fastareader quirks
(SURVEY.md §9.A), java.util.Formatter number formatting (§9.F), the summary / track / parameter text."""
import math
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def io(native):
    from plaac_amd import hostio
    hostio._lib()
    return hostio


def test_host_header_exports(native):
    L = native.load()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "plaac_host.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(plaac_[a-z0-9_]+)\s*\(", txt)))
    from plaac_amd import hostio
    assert declared == sorted(hostio.HOST_EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_fixed_formatting_short_cut_agrees_with_the_digit_string_path_and_with_decimal(io):
    """the arithmetic short cut of %.Nf (values away from a tie) against the digit-string path and against an independent
    statement of the rule: shortest round-trip digits (Python's repr), then ROUND_HALF_UP at the requested place"""
    import decimal
    import numpy as np
    rng = np.random.default_rng(314)
    vals = [0.0, -0.0, 0.0005, 0.00049999999999999994, 0.0625, 0.1875, 1.0005, 2.5e-4, 123456.7895, 999999.9995,
            4.5e6, 1e9 - 1e-3, 1e9, 3.3e15, 5e-324, 1e-300, 0.9995, 0.99949999999999994]
    vals += list(rng.normal(0, 50, 20000)) + list(rng.random(20000) * 10.0 ** rng.integers(-6, 9, 20000))
    for k in rng.integers(0, 10 ** 7, 20000):  # exact and nearly exact ties at three and four decimals
        for d in (1000.0, 10000.0):
            t = (k + 0.5) / d
            vals += [t, np.nextafter(t, 0.0), np.nextafter(t, np.inf)]
    for v in vals:
        for dec in (3, 4, 8):
            got = io.format_fixed(v, dec)
            assert got == io.format_fixed_reference(v, dec), (v, dec)
            want = decimal.Decimal(repr(abs(float(v)))).quantize(decimal.Decimal(1).scaleb(-dec), decimal.ROUND_HALF_UP)
            assert got == ("-" if np.signbit(v) else "") + format(want, "f"), (v, dec, got)


def test_java_fixed_formatting(io):
    f = io.format_fixed
    assert f(51.21464835140835, 3) == "51.215"
    assert f(float("nan"), 3) == "NaN"
    assert f(float("inf"), 3) == "Infinity" and f(float("-inf"), 3) == "-Infinity"
    assert f(0.0, 3) == "0.000" and f(-0.0, 3) == "-0.000"
    # HALF_UP on exact binary ties (C printf would round half-to-even: 0.062)
    assert f(0.0625, 3) == "0.063" and f(-0.0625, 3) == "-0.063" and f(0.1875, 3) == "0.188"
    # HALF_UP on the SHORTEST decimal repr, like java.util.Formatter: 0.0045 is 0.004499999... in binary
    assert f(0.0045, 3) == "0.005" and f(9 / 2000, 3) == "0.005" and f(1.0005, 3) == "1.001"
    assert f(1.005, 2) == "1.01"
    assert f(0.0004, 3) == "0.000" and f(0.0005, 3) == "0.001" and f(-0.0004, 3) == "-0.000"
    assert f(0.9996, 3) == "1.000" and f(9.9995, 3) == "10.000" and f(999.9999, 3) == "1000.000"
    assert f(1e-9, 3) == "0.000" and f(123456789.125, 3) == "123456789.125"
    assert f(-1.151, 3) == "-1.151" and f(2.0, 4) == "2.0000" and f(0.00000001234, 8) == "0.00000001"
    assert f(1e21, 3) == "1000000000000000000000.000"
    rng = np.random.default_rng(0)
    for v in np.concatenate([rng.normal(0, 50, 2000), rng.random(2000) * 1e-3]):
        for d in (3, 4, 8):
            # away from decimal ties Java and C agree
            assert f(v, d) == ("%.*f" % (d, v)), (v, d)


def test_java_double_tostring(io):
    t = io.double_tostring
    assert [t(v) for v in (1.0, 0.5, 0.25, 0.1, 0.0, 100.0, 0.001, 1234567.0)] == \
        ["1.0", "0.5", "0.25", "0.1", "0.0", "100.0", "0.001", "1234567.0"]
    assert t(0.0001) == "1.0E-4" and t(1e7) == "1.0E7" and t(1.5e-5) == "1.5E-5"


def test_fastareader_quirks(io, tmp_path):
    p = tmp_path / "q.fa"
    p.write_bytes(b"junk before first header\n"
                  b">first  \r\n"           # first name is trimmed, \r\n is a line end
                  b"MKV\n"
                  b"QQ N\n"                 # embedded blank stays (-> X)
                  b">second \n"             # later names are NOT trimmed
                  b"acd\n"
                  b"\n"                     # blank line ends the record ...
                  b"EFG\n"                  # ... and the rest is skipped
                  b">third\n"
                  b">fourth\n"              # third has an empty sequence
                  b"WY*\n"
                  b"   \n"                  # a line of blanks is sequence, not a terminator
                  b"K")                     # no final newline
    names, codes, offs = io.read_fasta(p)
    assert names == [b"first", b"second ", b"third", b"fourth"]
    seqs = [bytes(codes[int(offs[i]):int(offs[i + 1])]) for i in range(4)]
    enc = lambda s: bytes(__import__("plaac_amd").native.encode(s))
    assert seqs == [enc("MKVQQ N"), enc("acd"), b"", enc("WY*   K")]
    assert seqs[0][5] == 0  # the blank became X


def test_fasta_matches_simple_reader_on_fixtures(io, classic4, kat28, native):
    for path, recs in ((os.path.join(GOLDEN, "four_classic_prions.fasta"), classic4),
                       (os.path.join(GOLDEN, "kat28.fasta"), kat28[0])):
        names, codes, offs = io.read_fasta(path)
        assert [n.decode() for n in names] == [n for n, _ in recs]
        c2, o2 = native.pack([s for _, s in recs])
        assert np.array_equal(codes, c2) and np.array_equal(offs, o2)


def test_missing_file_is_an_io_error(io, native):
    with pytest.raises(native.PlaacError) as e:
        io.read_fasta("/nonexistent/x.fa")
    assert e.value.status == native.PLAAC_ERR_IO


def test_read_aa_params(io, tmp_path, oracle):
    vec, warn = io.read_aa_params(os.path.join(GOLDEN, "prd_freq_scer_04.txt"))
    assert np.array_equal(vec, oracle.const_tables()["fg04"]) and not warn.any()
    p = tmp_path / "bg.txt"
    p.write_text("".join("%d.000000 # %s\n" % (i * 7, a) for i, a in enumerate("XACDEFGHIKLMNPQRSTVWY*")))
    vec, warn = io.read_aa_params(p)
    assert vec.tolist() == [i * 7.0 for i in range(22)] and not warn.any()
    p.write_text("".join("%d # %s\n" % (i, a) for i, a in enumerate("XACDEFGHIKLMNPQRSTVYW*")))
    _, warn = io.read_aa_params(p)
    assert warn.nonzero()[0].tolist() == [19, 20]
    # round trip with the -b dump format (print_aa_params)
    txt = io.format_aa_params(np.arange(22) * 1000.5)
    assert txt.splitlines()[1] == "1000.500000 # A" and len(txt.splitlines()) == 22
    p.write_text(txt)
    assert np.array_equal(io.read_aa_params(p)[0], np.arange(22) * 1000.5)


def test_reference_written_background_files_round_trip_byte_for_byte(io, oracle):
    """tests/golden/bg_freqs/*.txt are the reference's OWN output (`plaac.jar -b`, print_aa_params plaac.java:2665-2669,
    copied by tests/golden/make_bg_freqs.py): plaac_read_aa_params (the -B reader, :2684-2713) followed by
    plaac_format_aa_params (the -b writer) must reproduce every file byte for byte; the counts are integers, the stop bin
    counts the terminal stops (countaas does not trim them, :1698-1706); and the oracle's table setup accepts them."""
    import glob
    files = sorted(glob.glob(os.path.join(GOLDEN, "bg_freqs", "bg_freqs_*.txt")))
    assert len(files) == 12
    for f in files:
        raw = open(f, "rb").read()
        vec, warn = io.read_aa_params(f)
        assert not warn.any(), f
        assert io.format_aa_params(vec).encode() == raw, f
        assert np.array_equal(vec, np.round(vec)) and vec[1:21].min() > 0
        Po = oracle.build_params(alpha=0.5, bgcounts=vec)
        bg = np.array(Po.bg)
        assert abs(bg.sum() - 1.0) < 1e-12 and bg[0] > 0 and bg[21] > 0  # eps on X and * (:490-495)
    human = io.read_aa_params(os.path.join(GOLDEN, "bg_freqs", "bg_freqs_HUMAN.txt"))[0]
    assert human[1] == 2428201.0 and human[0] == 6721.0  # (first lines of the file as the reference wrote them)


def test_summary_row_text(io, oracle, classic4):
    """format a Sup35p row produced by the oracle (tests may use it) and check every column"""
    codes, offs = oracle.pack([s for _, s in classic4])
    rows = oracle.score_batch(oracle.build_params(), codes, offs)
    line = io.format_summary_row(rows[0], "Sup35p", codes[int(offs[0]):int(offs[1])])
    col = dict(zip(io.summary_header().split("\t"), line.split("\t")))
    assert len(line.split("\t")) == 38 == len(io.summary_header().split("\t"))
    assert col["SEQid"] == "Sup35p" and col["PROTlen"] == "685"
    assert (col["COREstart"], col["COREend"], col["CORElen"]) == ("5", "64", "60")
    assert (col["PRDstart"], col["PRDend"], col["PRDlen"]) == ("1", "133", "133")
    assert col["COREscore"] == "51.215" and col["LLR"] == "51.215" and col["NLLR"] == "0.854"
    assert col["HMMall"] == "81.820" and col["HMMvit"] == "79.598" and col["PRDscore"] == "89.773"
    seq = classic4[0][1]
    assert col["COREaa"] == seq[4:64] and col["PRDaa"] == seq[0:133]
    assert col["STARTaa"] == seq[0:15] and col["ENDaa"] == seq[118:133]
    cen = int(col["PAPAcen"])
    assert col["PAPAaa"] == seq[cen - 1 - 20:cen - 1 + 21]
    assert (col["MWstart"], col["MWend"], col["MWlen"]) == ("4", "83", "80") and col["MW"] == "38"


def test_summary_row_sentinels(io, oracle):
    """no core / n < c: start 0, end -1, len 0, NaN scores, '-' strings, PAPAaa = first <= 20 residues"""
    codes, offs = oracle.pack(["MKVLAAGIVGLDEE", "mkvl*", "*"])
    rows = oracle.score_batch(oracle.build_params(), codes, offs)
    cols = io.format_summary_row(rows[0], "short", codes[int(offs[0]):int(offs[1])]).split("\t")
    col = dict(zip(io.summary_header().split("\t"), cols))
    assert (col["LLR"], col["LLRstart"], col["LLRend"], col["LLRlen"], col["NLLR"]) == ("NaN", "0", "-1", "0", "NaN")
    assert (col["COREscore"], col["COREstart"], col["COREend"], col["CORElen"]) == ("NaN", "0", "-1", "0")
    assert (col["PRDscore"], col["PRDstart"], col["PRDend"], col["PRDlen"]) == ("0.000", "0", "-1", "0")
    assert (col["COREaa"], col["STARTaa"], col["ENDaa"], col["PRDaa"]) == ("-", "-", "-", "-")
    assert col["PAPAcombo"] == "NaN" and col["PAPAprop"] == "NaN" and col["PAPAcen"] == "0"
    assert col["PAPAaa"] == "MKVLAAGIVGLDEE" and col["MWlen"] == "14"
    # lower case is printed upper case, the trimmed stop is not part of the protein
    line = io.format_summary_row(rows[1], "lc", codes[int(offs[1]):int(offs[2])])
    assert line.split("\t")[-1] == "MKVL" and line.split("\t")[19] == "4"
    # a stop-only record yields no line
    assert io.format_summary_row(rows[2], "stop", codes[int(offs[2]):int(offs[3])]) == ""


def test_track_rows_text(io, oracle):
    seq = "MQNQQNYQQGGYNNSS" * 6
    codes, offs = oracle.pack([seq])
    rows, tr = oracle.score_batch(oracle.build_params(), codes, offs, tracks=True)
    txt = io.format_track_rows(tr, 0, codes, len(seq), "7", "my name")
    lines = txt.split("\n")
    assert lines[-1] == "" and lines[-2] == "#" * 56 and len(lines) == len(seq) + 2
    assert len(io.tracks_header().split("\t")) == 16
    f = lines[0].split("\t")
    assert len(f) == 16 and f[:4] == ["7", "my name", "1", "M"] and f[4] in "01" and f[5] in "01"
    assert f[11] == "NaN" and f[12] == "NaN" and f[13] == "NaN"  # x2 tracks are NaN within 20 of the ends
    mid = lines[48].split("\t")
    assert mid[2] == "49" and mid[3] == seq[48]
    assert mid[6] == io.format_fixed(tr["charge"][48], 4) and mid[8] == io.format_fixed(tr["fi"][48], 8)
    assert abs(float(mid[14]) + float(mid[15]) - 1.0) < 2e-3


def test_param_block_text(io, native):
    P = native.make_params(alpha=0.5, bgcounts=np.arange(22.0) + 1, corelength=30, ww1=21, ww2=31)
    lines = io.format_param_block(P).split("\n")
    assert lines[0].startswith("####") and "parameters at run-time" in lines[0]
    assert lines[1] == "## alpha=0.5; corelength=30; ww1=21; ww2=31; ww3=31; adjustprolines=true;"
    assert lines[2].startswith("## fg_used: {X=0.00001;A=0.04865;C=0.00219;") and lines[2].endswith(";}")
    assert [l.split(":")[0] for l in lines[2:8]] == ["## fg_used", "## bg_scer", "## bg_input", "## bg_used",
                                                     "## plaac_llr", "## papa_lods"]
    assert lines[4].startswith("## bg_input: {X=0.00000;A=0.00870;")  # X and * zeroed, rest normalised
    assert lines[8] == "#" * 87


def _py_fastareader(data):
    """straight Python restatement of fastareader (:4302-4375) for cross-checking the parallel C++ reader"""
    import re as _re
    lines = _re.split(b"\r\n|\n|\r", data)
    if lines and lines[-1] == b"":
        lines.pop()
    recs, i, ondeck, name = [], 0, False, None
    jtrim = lambda b: b.strip(bytes(range(0, 33)))
    while True:
        if not ondeck:
            found = False
            while i < len(lines):
                ln = lines[i]
                i += 1
                if len(ln) > 0 and ln[:1] == b">":
                    name = jtrim(ln)[1:]
                    found = True
                    break
            if not found:
                break
        seq, ondeck, nextname = [], False, None
        while i < len(lines):
            ln = lines[i]
            i += 1
            if len(ln) == 0:
                break
            if ln[:1] == b">":
                ondeck, nextname = True, ln[1:]
                break
            seq.append(ln)
        recs.append((name, b"".join(seq)))
        if ondeck:
            name = nextname
    return recs


def test_parallel_fasta_reader_matches_serial_semantics(io, native, tmp_path, monkeypatch):
    rng = np.random.default_rng(12)
    aas = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdXB* -", dtype=np.uint8)
    out = [b"leading junk\n"]
    for r in range(7000):
        out.append(b">rec%d some text%s" % (r, b"  " if r % 7 == 0 else b"") + (b"\r\n" if r % 5 == 0 else b"\n"))
        for _ in range(int(rng.integers(0, 4))):
            out.append(bytes(rng.choice(aas, int(rng.integers(1, 70)))) + (b"\r\n" if r % 11 == 0 else b"\n"))
        if r % 13 == 0:
            out.append(b"\nskipped tail\nmore skipped\n")  # blank line: rest of the record is ignored
    data = b"".join(out)[:-1]  # no final newline
    p = tmp_path / "big.fa"
    p.write_bytes(data)
    want = _py_fastareader(data)
    for nthreads in ("1", "5"):
        monkeypatch.setenv("PLAAC_THREADS", nthreads)
        names, codes, offs = io.read_fasta(p)
        assert len(names) == len(want) == 7000
        assert names == [n for n, _ in want]
        got = [bytes(codes[int(offs[i]):int(offs[i + 1])]) for i in range(len(names))]
        assert got == [bytes(native.encode(s)) for _, s in want]


@pytest.mark.parametrize("max_records,max_bytes", [(1, 1 << 30), (3, 1 << 30), (1000, 1 << 30), (1 << 20, 1), (1 << 20, 700),
                                                   (1 << 20, 50000), (977, 33333)])
def test_streamed_batches_concatenate_to_the_whole_file_read(io, tmp_path, max_records, max_bytes):
    """plaac_fasta_open/_next (bounded-memory reader): any batch size gives the records, names (trimming depends on how
    the PREVIOUS record ended, also across a batch boundary) and residues of plaac_fasta_read"""
    rng = np.random.default_rng(5)
    aas = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdXB* ->", dtype=np.uint8)  # '>' inside a line is not a header
    out = [b"leading junk\n>first header after junk  \n"]
    for r in range(1500):
        out.append(b">rec%d text%s" % (r, b" \t " if r % 3 == 0 else b"") + (b"\r\n" if r % 5 == 0 else b"\n"))
        for _ in range(int(rng.integers(0, 4))):
            out.append(b"A" + bytes(rng.choice(aas, int(rng.integers(1, 70)))) + (b"\r" if r % 11 == 0 else b"\n"))
        if r % 4 == 0:
            out.append(b"\nskipped\n")
    p = tmp_path / "s.fa"
    p.write_bytes(b"".join(out))
    names, codes, offs = io.read_fasta(p)
    got_names, got_seqs, nb = [], [], 0
    for bn, bc, bo in io.stream_fasta(p, max_records, max_bytes):
        assert 1 <= len(bn) <= max_records
        nb += 1
        got_names += bn
        got_seqs += [bytes(bc[int(bo[i]):int(bo[i + 1])]) for i in range(len(bn))]
    assert got_names == names
    assert got_seqs == [bytes(codes[int(offs[i]):int(offs[i + 1])]) for i in range(len(names))]
    if max_records == 1:
        assert nb == len(names)


def test_streamed_reader_on_empty_and_headerless_files(io, tmp_path):
    for name, data in (("e.fa", b""), ("j.fa", b"no header at all\nMKV\n")):
        p = tmp_path / name
        p.write_bytes(data)
        assert list(io.stream_fasta(p)) == []
        assert io.read_fasta(p)[0] == []


def test_hmm_dot_export(io, native):
    """-h: the GraphViz text of hmm.dottify(file, true) for prionhmm1 (plaac.java:4209-4287)"""
    P = native.make_params()
    lines = io.format_hmm_dot(P).split("\n")
    assert lines[0] == "Digraph G {" and lines[-2] == "}" and lines[-1] == ""
    assert '  n0 [label="background", shape=circle, height=1.2];' in lines
    assert '  n1 [label="PrD-like", shape=circle, height=1.2];' in lines
    assert '  start -> n0 [label="0.952", color=gray];' in lines and '  start -> n1 [label="0.048", color=gray];' in lines
    assert '  n0:w -> n0:w [label="0.999", color=gray];' in lines and '  n1:e -> n1:e [label="0.980", color=gray];' in lines
    assert '  n0 -> n1 [label="0.001", color=gray, constraint=false];' in lines
    assert '  n1 -> n0 [label="0.020", color=gray, constraint=false];' in lines
    assert '  n0 -> n1 [label="spacerlabel", color=gray, constraint=false, style=invis];' in lines
    rec1 = [l for l in lines if l.startswith("rec1 ")][0]
    assert rec1.startswith('rec1 [shape=record, label="{ <fs> AA|A|C|D|E|') and "|Y}|{ <f1> prob|" in rec1
    probs = rec1.split("prob|")[1].split("}")[0].split("|")
    assert len(probs) == 20 and probs[11] == io.format_fixed(P.fg[12] / sum(P.fg), 4)  # N is the 12th code
    assert lines[-4:-2] == ["  n0 -> rec0 [style=dashed];", "  n1 -> rec1 [style=dashed];"]


def test_text_batches_locate_the_records_the_parser_finds(io, tmp_path):
    """plaac_fasta_next_text (the host half of the device-side parse): the same records as plaac_fasta_read - their starts are
    the '>' of a line, their untrimmed names the header text - for any batch cut; junk before the first header belongs to no
    batch; the batches tile the file from the first header on."""
    rng = np.random.default_rng(3)
    out = [b"junk\r\n>a  \n"]
    for r in range(800):
        out.append(b">r%d  \t" % r + (b"\r\n" if r % 3 == 0 else b"\n") + b"ACD>EF\n" * int(rng.integers(0, 3)) + (b"\n" if r % 5 == 0 else b""))
    data = b"".join(out)
    p = tmp_path / "t.fa"
    p.write_bytes(data)
    names, codes, offs = io.read_fasta(p)
    for mr, mb in ((1 << 20, 1 << 30), (7, 1 << 30), (1 << 20, 100)):
        pos, n, untrimmed = data.index(b">"), 0, []
        for text, starts, trim in io.stream_fasta_text(p, mr, mb):
            assert data[pos:pos + len(text)] == text and len(starts) - 1 <= mr
            for i in range(len(starts) - 1):
                assert text[int(starts[i]):int(starts[i]) + 1] == b">"
            nm, _ = trim(np.zeros(len(starts) - 1, np.uint8), 0)  # (no trimming: raw header text)
            untrimmed += nm
            pos += len(text)
            n += len(starts) - 1
        assert pos == len(data) and n == len(names)
        # the reader's own look at each batch's ends: the file's first batch follows "a blank line", a batch's last record ends
        # in one iff an empty line follows its header before the next record (r % 5 == 0 above), and the flags chain
        flags = [trim.flags for _, _, trim in io.stream_fasta_text(p, mr, mb)]
        assert flags[0][0] == 1 and all(a[1] == b[0] for a, b in zip(flags, flags[1:]))
        k = 0
        for (text, starts, trim) in io.stream_fasta_text(p, mr, mb):
            k += len(starts) - 1
            last = k - 2  # (record k - 1 of the file is ">r<k-2>", the first one is ">a")
            assert trim.flags[1] == (1 if (last >= 0 and last % 5 == 0) else 0), (k, trim.flags)
        assert [u.rstrip(bytes(range(33))) if (i == 0) else u for i, u in enumerate(untrimmed)][:1] == names[:1]
        assert all(u.startswith(nm) for u, nm in zip(untrimmed, names))


def test_summary_row_written_straight_into_a_buffer(io):
    """plaac_format_summary_row_n (round 5: bin/plaac's formatter threads write rows straight into their output buffers): the
    name by length (not NUL-terminated), no terminating NUL written, -1 unless the buffer covers the longest a row can be,
    and the same bytes as plaac_format_summary_row - also for NaN / infinite fields and values on a %.3f tie."""
    import ctypes as C
    from plaac_amd import native
    L = native.load()
    L.plaac_format_summary_row_n.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.plaac_format_summary_row_n.restype = C.c_long
    rng = np.random.default_rng(12)
    codes = rng.integers(1, 21, 400).astype(np.uint8)
    rows = np.zeros(6, dtype=native.ROW_DTYPE)
    for i, r in enumerate(rows):
        r["prot_len"] = 400
        for f in ("llr_score", "core_score", "prd_score", "hmm_all", "hmm_vit", "fi_meanhydro", "fi_meancharge", "fi_meancombo",
                  "papa_combo", "papa_prop", "papa_fi", "papa_llr", "papa_llr2"):
            r[f] = rng.normal(0, 30)
        r["mw_start"], r["mw_end"], r["llr_start"], r["llr_end"] = 5, 84, 10, 69
        r["core_start"], r["core_end"], r["prd_start"], r["prd_end"], r["papa_cen"] = 20, 79, 15, 140, 100 + i
    rows[1]["core_score"] = np.nan
    rows[2]["llr_score"] = -np.inf
    rows[2]["papa_combo"] = -np.inf
    rows[3]["fi_meancharge"] = 0.0625      # an exact tie at three decimals
    rows[4]["fi_meanhydro"] = 0.0475       # the shortest digits sit on the tie, the binary value below it
    rows[5]["prd_start"], rows[5]["prd_end"] = -1, -2
    name = b"sp|P1|NAME one two"
    for r in rows:
        one = np.array([r], dtype=native.ROW_DTYPE)
        want = io.format_summary_row(r, name, codes)
        buf = C.create_string_buffer(b"\xff" * 16384, 16384)
        k = L.plaac_format_summary_row_n(one.ctypes.data, name + b"JUNK BEHIND THE NAME", len(name), codes.ctypes.data, len(codes), 60, 41, buf, 16384)
        assert k == len(want.encode()) and buf.raw[:k] == want.encode() and buf.raw[k:k + 1] == b"\xff"
        assert L.plaac_format_summary_row_n(one.ctypes.data, name, len(name), codes.ctypes.data, len(codes), 60, 41, buf, 2000) == -1
    assert "0.063" in io.format_summary_row(rows[3], name, codes) and "0.048" in io.format_summary_row(rows[4], name, codes)


def test_text_batches_outlive_the_close_of_their_stream(tmp_path):
    """bin/plaac's no-GPU exit closed the FASTA stream while text batches were still queued; freeing them then read
    the deleted stream (and told the kernel to drop pages of a mapping that was gone). The stream now counts the batches it has
    handed out: plaac_fasta_close only gives up the handle, the file image stays mapped until the last batch is freed - the
    batch's text stays readable in between."""
    import ctypes as C
    from plaac_amd import hostio
    L = hostio._lib()
    L.plaac_fasta_next_text.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(hostio._FastaText))]
    L.plaac_fasta_text_free.argtypes = [C.POINTER(hostio._FastaText)]
    L.plaac_fasta_text_free.restype = None
    p = tmp_path / "big.fa"
    rec = b">name of record %07d\n" + b"ACDEFGHIKLMNPQRSTVWY" * 30 + b"\n"
    p.write_bytes(b"".join(rec % i for i in range(40000)))  # 25 MB: mapped, several pages per batch
    h = C.c_void_p()
    assert L.plaac_fasta_open(str(p).encode(), C.byref(h)) == 0
    batches = []
    for _ in range(3):
        pt = C.POINTER(hostio._FastaText)()
        assert L.plaac_fasta_next_text(h, 5000, 1 << 30, C.byref(pt)) == 0 and pt
        batches.append(pt)
    L.plaac_fasta_close(h)  # (the handle is gone; three batches are still alive)
    for k, pt in enumerate(batches):
        t = pt.contents
        text = C.string_at(t.text, int(t.len))
        assert t.nrec == 5000 and text.startswith(b">name of record %07d\n" % (5000 * k))
        L.plaac_fasta_text_free(pt)  # (the last one unmaps the file)
