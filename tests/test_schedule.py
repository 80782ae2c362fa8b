"""CPU: the scheduling of a scoring call as data (plaac_amd/csrc/schedule.hip.inc through plaac_debug_schedule).

A call is some forty kernel launches on up to ten streams; which kernels, in which forms and on which streams is a pure
function of the call's kind, the words the planning kernels send back, the context's history and the environment
switches. This test walks that decision table - batch shape x mode x overlap x sweep x knobs - WITHOUT a device and
replays every schedule (and, for overlapping calls, several consecutive ones) in a small happens-before model:

  * streams are in-order; `R stream event` records, `W stream event` waits for the LAST recording of that event;
  * a wait for an event that is only recorded LATER in the same call can never be meant (HIP would wait for the previous
    call's recording): flagged unless the event is one that crosses calls by design (previous joins, ka_done, tail);
  * every launch lists the buffers it reads (r), writes (w) or appends to atomically (a); two accesses conflict unless
    both are reads or both are atomic appends; every pair of conflicting accesses - within a call and across the calls of
    a pipeline - must be ordered by the stream / event graph.

The reference has nothing of the kind (one thread, plaac.java:755); this is the contract the overlapping, multi-stream
replacement has to keep."""
import itertools
import os

import pytest

from conftest import ROOT  # noqa: F401

EK = dict(T=0, PK=1, J=2, F=3, L=4, KA=5, TAIL=6, KB=7, TF=8, TB=9, TP=10, G=11, GJ=12, PREVJOIN=13, R=14)
E_JOIN = 10
# buffers (schedule.hip.inc, enum Buf)
BUF = ["NEFF", "HIST", "ORDER", "GROW", "HUGE", "HPIN", "PACKED", "BITS", "LMARG", "H0", "VEND", "CORELIST", "CORECOUNT",
       "COREFLAGS", "COREP", "COREPART", "FWDP", "BWDP", "CLIST", "CCOUNT", "FBLIST", "FBCOUNT", "ROW_VIT", "ROW_CORE",
       "ROW_HMMVIT", "ROW_HMMALL", "ROW_MW", "ROW_LLR", "ROW_KB", "ROW_KBLLR", "TRK_WIN", "TRK_POST", "TRK_VIT"]
PER_PARITY = {"NEFF", "HIST", "ORDER", "GROW", "PACKED", "BITS", "LMARG", "H0", "VEND", "CORELIST", "COREFLAGS", "COREP", "COREPART",
              "CLIST", "FBLIST"}
FOUR_SETS = {"HUGE", "FBCOUNT", "CCOUNT", "CORECOUNT"}  # four copies used in turn: the tail of call k reads its own while call k+2 plans
PER_CALL = {b for b in BUF if b.startswith("ROW_") or b.startswith("TRK_")} | {"HPIN"}  # the caller's buffers: distinct per call


def query(native, nprot, residues, rows_first, long_groups=0, long_rows=0, total_rows=None, points=(1,), kb_base=None,
          lane=True, fast20=True, wmax=20, tracks=False, overlap=False, ncalls=0, last=(0, 0, 0), old_tail=0, marks=None):
    q = native.SchedQuery()
    q.nprot, q.residues, q.rows_first, q.long_groups, q.long_rows = nprot, residues, rows_first, long_groups, long_rows
    q.total_rows = total_rows if total_rows is not None else max(1, residues // 16 + nprot // 64)
    ngroups = (nprot + 63) // 64
    for k in range(7):
        q.run_mark[k] = (marks[k] if marks else min(ngroups - 1, (k + 1) * ngroups // 8)) if ngroups > 8 else 0xffffffff
    q.npoints = sum(points)
    q.ngroups_sweep = len(points)
    for g, m in enumerate(points):
        q.group_members[g] = m
        q.kb_base[g] = -1 if kb_base is None else kb_base[g]
    q.lane_possible, q.fast20, q.wmax, q.core_par_tables = int(lane), int(fast20), wmax, 1
    q.tracks, q.overlap, q.ncalls = int(tracks), int(overlap), ncalls
    q.last_chain_bound, q.last_mixed, q.last_single_summary = last
    q.old_tail = old_tail
    return q


def parse(text):
    forms, alias, ops = {}, {}, []
    for line in text.splitlines():
        t = line.split()
        if t[0] == "F":
            forms = {k: int(v) for k, v in (x.split("=") for x in t[1:])}
        elif t[0] == "ALIAS":
            alias[int(t[1])] = int(t[2])
        elif t[0] == "X":
            raise AssertionError("schedule dump reports: " + line)  # an access list overflowed: the hazard check would be unsound
        elif t[0] in "RW":
            k, i = t[2].split(".")
            ops.append((t[0], int(t[1]), (int(k), int(i))))
        else:
            acc = [(a[0], int(a[1:].split(".")[0]), int(a.split(".")[1])) for a in t[16:]]
            ops.append(("L", int(t[1]), t[2], acc, line))
    return forms, alias, ops


class Model:
    """happens-before over the operations of consecutive calls of one context"""

    def __init__(self):
        self.clock = {}      # stream -> vector clock {stream: last op number known to have happened}
        self.n = {}          # stream -> ops issued
        self.events = {}     # event key -> vector clock at its last recording
        self.acc = {}        # buffer key -> list of (mode, stream, op number, description)
        self.errors = []

    def run(self, callno, forms, alias, ops, what):
        par = callno & 1
        recorded_later = {}
        for i, o in enumerate(ops):
            if o[0] == "R":
                recorded_later.setdefault(self.ekey(o[2], callno, par), i)
        for i, o in enumerate(ops):
            s = alias.get(o[1], o[1])
            if forms["serial"]:
                s = 0
            vc = self.clock.setdefault(s, {})
            self.n[s] = self.n.get(s, 0) + 1
            vc[s] = self.n[s]
            if o[0] == "R":
                self.events[self.ekey(o[2], callno, par)] = dict(vc)
            elif o[0] == "W":
                key = self.ekey(o[2], callno, par)
                if key in self.events:
                    for k, v in self.events[key].items():
                        vc[k] = max(vc.get(k, 0), v)
                first = recorded_later.get(key)
                crosses = o[2][0] in (EK["PREVJOIN"], EK["KA"], EK["TAIL"])
                if first is not None and first > i and not crosses:
                    self.errors.append("%s: op %d waits for event %s that this call only records later (op %d)" % (what, i, o[2], first))
            else:
                for mode, b, q in o[3]:
                    name = BUF[b]
                    key = (name,) + ((par,) if name in PER_PARITY else (callno & 3,) if name in FOUR_SETS else (callno,) if name in PER_CALL else ())
                    for (m2, q2, s2, n2, d2) in self.acc.get(key, []):
                        if (mode == "r" and m2 == "r") or (mode == "a" and m2 == "a") or (q != q2 and 255 not in (q, q2)):
                            continue  # (q = 255: every part of the buffer)
                        if vc.get(s2, 0) < n2:
                            self.errors.append("%s: %s (%s %s.%d) on stream %d is not ordered behind %s (%s .%d)" % (
                                what, o[2], mode, key, q, s, d2, m2, q2))
                    self.acc.setdefault(key, []).append((mode, q, s, self.n[s], "%s of call %d" % (o[2], callno)))

    @staticmethod
    def ekey(e, callno, par):
        kind, idx = e
        if kind == EK["T"]:
            return ("T", callno, idx)
        if kind == EK["PREVJOIN"]:
            return ("T", callno - idx, E_JOIN)
        if kind == EK["L"]:
            return (kind, par)
        return (kind, idx)


SHAPES = {
    # name: (nprot, residues, rows_first, long_groups, long_rows)
    "tiny": (3, 900, 25, 0, 0),
    "cfg2": (5880, 2_900_000, 307, 1, 307),
    "cfg3": (20600, 11_400_000, 2147, 10, 4000),
    "share": (1_250_000, 364_000_000, 2250, 45, 9000),
    "full": (10_000_000, 2_914_000_000, 2250, 360, 72000),
    "flat": (300_000, 60_000_000, 20, 0, 0),
    "huge": (3000, 1_000_000, 4400, 1, 4400),
}


def consecutive(native, shape, n=4, monkey=None, **kw):
    """schedules of n consecutive calls of one kind, the history threaded from call to call"""
    out, last, tails = [], (0, 0, 0), {}
    for c in range(n):
        q = query(native, *SHAPES[shape], ncalls=c, last=last, old_tail=tails.get(c - 2, 0), **kw)
        forms, alias, ops = parse(native.debug_schedule(q))
        out.append((c, forms, alias, ops))
        single_summary = int(q.npoints == 1 and not q.tracks)
        last = (forms["chain_bound"], forms["mixed"], single_summary)
        tails[c] = int(any(o[0] == "R" and o[2][0] == EK["TAIL"] for o in ops))
    return out


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("mode", ["summary", "tracks", "sweep3x3", "sweep2"])
def test_no_wait_can_deadlock_and_conflicting_accesses_are_ordered(native, shape, overlap, mode):
    kw = dict(overlap=overlap)
    if mode == "tracks":
        kw.update(tracks=True)
    elif mode == "sweep3x3":
        kw.update(points=(3, 3, 3), kb_base=(-1, 0, 0))
    elif mode == "sweep2":
        kw.update(points=(1, 5), kb_base=(-1, -1))
    m = Model()
    for c, forms, alias, ops in consecutive(native, shape, 6, **kw):
        m.run(c, forms, alias, ops, "%s/%s/overlap=%s call %d" % (shape, mode, overlap, c))
    assert not m.errors, "\n".join(m.errors[:12])


KNOBS = [{"PLAAC_SWEEP_SPREAD": "0", "PLAAC_SWEEP_REST_ASIDE": "1"}, {"PLAAC_SWEEP_SPREAD": "0", "PLAAC_SWEEP_CORE_ASIDE": "0"}, {"PLAAC_SWEEP_CORE_ASIDE": "0"}, {"PLAAC_SWEEP_CHAINS_FIRST": "1"}, {"PLAAC_SWEEP_CORE_ASIDE": "0", "PLAAC_SWEEP_CHAINS_FIRST": "1"}, {"PLAAC_TRACK_ONE_PASS": "1"}, {"PLAAC_TRACK_KB_LATE": "1"}, {"PLAAC_TRACK_VIT_EARLY": "0"}, {"PLAAC_SERIAL_STREAMS": "1"}, {"PLAAC_KB_SIDE": "0"}, {"PLAAC_FINISH_KERNEL": "1"}, {"PLAAC_KB_PRIO": "0"}, {"PLAAC_TRACK_FUSED": "0"}, {"PLAAC_TRACK_VIT_MIXED": "0"}, {"PLAAC_TRACK_CONSEC": "0"}, {"PLAAC_TRACK_CKPT": "0"}, {"PLAAC_TRACK_SEGMENTS": "2"},
         {"PLAAC_MIXED_MIN_REST": "1", "PLAAC_TRACK_SEGMENTS": "4", "PLAAC_SEGMENT_MIN_ROWS": "1"}, {"PLAAC_MIXED": "0"}, {"PLAAC_LATENCY_MODE": "1"}, {"PLAAC_LATENCY_MODE": "0"},
         {"PLAAC_KB_LANE": "0"}, {"PLAAC_KB_FILTER": "0"}, {"PLAAC_CORE_LIST": "0"}, {"PLAAC_MIXED_GROUPS": "3", "PLAAC_MIXED_MIN_REST": "1"},
         {"PLAAC_KB_LANE_MIN_GROUPS": "1"}, {"PLAAC_PIPE_SEGMENTS": "4", "PLAAC_SEGMENT_MIN_ROWS": "1"},
         {"PLAAC_TRACK_SEGMENTS": "4", "PLAAC_SEGMENT_MIN_ROWS": "1"}, {"PLAAC_SWEEP_SPREAD": "0"}, {"PLAAC_SWEEP_LATENCY": "0"},
         {"PLAAC_KB_PER_PROTEIN": "1"}, {"PLAAC_GENERIC_TRACKS": "1"}, {"PLAAC_CORE_PAR": "0"}, {"PLAAC_FI_INT": "0"}]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_every_knob_keeps_the_schedule_sound(native, monkeypatch, knobs):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for shape, overlap, mode in itertools.product(("tiny", "cfg3", "share", "flat"), (False, True), ("summary", "tracks", "sweep")):
        kw = dict(overlap=overlap)
        if mode == "tracks":
            kw.update(tracks=True)
        elif mode == "sweep":
            kw.update(points=(3, 3, 3), kb_base=(-1, 0, 0))
        m = Model()
        for c, forms, alias, ops in consecutive(native, shape, 3, **kw):
            m.run(c, forms, alias, ops, "%s/%s/overlap=%s/%s call %d" % (shape, mode, overlap, knobs, c))
        assert not m.errors, "\n".join(m.errors[:12])


def test_calls_of_different_kinds_in_a_row_stay_ordered(native):
    """the transitions: throughput-bound -> chain-bound (mixed) -> tracks -> sweep -> summary again, overlapping"""
    kinds = [("full", {}), ("share", {}), ("share", {}), ("cfg2", {}), ("cfg3", dict(tracks=True)),
             ("cfg3", dict(points=(3, 3, 3), kb_base=(-1, 0, 0))), ("share", {}), ("flat", {}), ("full", {}), ("share", {})]
    m, last, tails = Model(), (0, 0, 0), {}
    for c, (shape, kw) in enumerate(kinds):
        q = query(native, *SHAPES[shape], ncalls=c, last=last, old_tail=tails.get(c - 2, 0), overlap=True, **kw)
        forms, alias, ops = parse(native.debug_schedule(q))
        m.run(c, forms, alias, ops, "mixed sequence call %d (%s)" % (c, shape))
        last = (forms["chain_bound"], forms["mixed"], int(q.npoints == 1 and not q.tracks))
        tails[c] = int(any(o[0] == "R" and o[2][0] == EK["TAIL"] for o in ops))
    assert not m.errors, "\n".join(m.errors[:12])


def test_the_decision_table(native):
    """the forms the library picks for BASELINE's configurations: config 2 / 3 chain-bound, all wave-groups in the
    long run; the 1.25 M-sequence share chain-bound in mixed forms; the 10 M batch throughput-bound with the core list"""
    f = {s: parse(native.debug_schedule(query(native, *SHAPES[s])))[0] for s in SHAPES}
    assert f["cfg2"]["chain_bound"] and f["cfg2"]["mixed"] and f["cfg2"]["gl"] == 92 and not f["cfg2"]["use_core_list"]
    assert f["cfg3"]["mixed"] and f["cfg3"]["gl"] == 322 and f["cfg3"]["core_long"]
    assert f["share"]["mixed"] and f["share"]["gl"] == 45 and f["share"]["use_core_list"] and f["share"]["runs"] == 2
    assert f["share"]["kb_after_pack"] and f["share"]["kb_deferred"] and not f["share"]["maybe_huge"]
    assert not f["full"]["chain_bound"] and not f["full"]["mixed"] and f["full"]["use_core_list"] and f["full"]["runs"] == 1
    assert not f["flat"]["chain_bound"] and f["flat"]["kb_after_pack"]
    # overlapping calls: the head of every call after the first runs aside; single-point summary calls only
    o = [c[1] for c in consecutive(native, "share", 3, overlap=True)]
    assert [x["head_aside"] for x in o] == [0, 1, 1] and all(x["tail_allowed"] for x in o)
    t = [c[1] for c in consecutive(native, "share", 3, overlap=True, tracks=True)]
    assert not any(x["head_aside"] or x["tail_allowed"] for x in t)
    # throughput-bound single-point calls write HMMall / HMMvit from k_fwd / k_vit (no k_finish); chain-bound ones finish the long
    # run with k_finish and let the throughput-form runs write their own
    full = consecutive(native, "full", 2, overlap=True)[1][3]
    assert not any(o[0] == "L" and o[2] == "k_finish" for o in full)
    assert any(o[0] == "L" and o[2] == "k_fwd" and "ext=0" in o[4] for o in full)
    share = consecutive(native, "share", 2, overlap=True)[1][3]
    assert sum(1 for o in share if o[0] == "L" and o[2] == "k_finish") == 1
    assert any(o[0] == "L" and o[2] == "k_fwd" and "ext=0" in o[4] for o in share)
    # a call has at most ~120 operations (the host's share of a 3 ms step)
    assert all(len(parse(native.debug_schedule(query(native, *SHAPES[s], overlap=True, ncalls=3, last=(1, 1, 1))))[2]) < 130 for s in SHAPES)


def test_the_model_sees_a_missing_wait(native):
    """the checker itself: take the share's schedule and drop, in turn, the wait of the long stream for the Viterbi side's
    event, the waits for the packed copy, and the record of a join event - every removal must be reported"""
    calls = consecutive(native, "share", 3, overlap=True)

    def errors_without(pred):
        m = Model()
        for c, forms, alias, ops in calls:
            m.run(c, forms, alias, [o for o in ops if not pred(o)], "call %d" % c)
        return m.errors

    assert not errors_without(lambda o: False)
    e = errors_without(lambda o: o[0] == "W" and o[2] == (EK["L"], 0))
    assert any("k_finish" in x and "VEND" in x for x in e), e[:3]
    e = errors_without(lambda o: o[0] == "W" and o[2][0] == EK["PK"])
    assert any("PACKED" in x for x in e), e[:3]
    # the head of call k+2 no longer waits for the window kernels of call k, which read the plan it overwrites
    e = errors_without(lambda o: o[0] == "W" and o[2][0] == EK["TAIL"])
    assert any("call 2" in x and ("ORDER" in x or "PACKED" in x or "GROW" in x) for x in e), e[:3]


def _field(line, name):
    return int(next(x for x in line.split() if x.startswith(name + "=")).split("=")[1])


@pytest.mark.parametrize("members", [(300,), (1, 700), (257, 3, 4096 - 260)])
def test_every_member_of_a_large_sweep_group_is_served_exactly_once(native, members):
    """Op::m0 was a byte, so members >= 256 of a sweep group were never written (and members 0.. were
    rewritten). Every per-core-length kernel must cover first..first+members-1 of its group exactly once, for groups of
    more than 256 core lengths and point numbers up to the JNI limit of 4096 (Access.q = point * 16 + part)."""
    q = query(native, *SHAPES["cfg3"], points=members, kb_base=tuple([-1] * len(members)))
    forms, alias, ops = parse(native.debug_schedule(q))
    for g, m in enumerate(members):
        for kern, first in (("k_vit", 0), ("k_win", 0), ("k_core_list", 0), ("k_replicate", 1)):
            seen = []
            for o in ops:
                if o[0] == "L" and o[2] == kern and _field(o[4], "group") == g:
                    if kern == "k_core_list" and not forms["use_core_list"]:
                        continue
                    m0, nc = _field(o[4], "m0"), _field(o[4], "nc")
                    seen += list(range(m0, m0 + nc))
            if kern == "k_core_list" and not seen:
                continue
            runs = max(1, forms["runs"]) if kern in ("k_vit", "k_win") else 1
            want = sorted(list(range(first, m)) * (len(seen) // max(1, m - first)))
            assert m - first == 0 or (sorted(seen) == want and len(seen) % (m - first) == 0), (kern, g, sorted(set(range(first, m)) - set(seen))[:5])
    # the row parts a launch writes carry the point number: no wrap at point 4096
    top = max(q_ for o in ops if o[0] == "L" for (_, b, q_) in o[3] if BUF[b].startswith("ROW_"))
    assert top // 16 == sum(members) - 1
    if sum(members) <= 400:  # (the happens-before model is quadratic in the accesses to a buffer)
        m = Model()
        m.run(0, forms, alias, ops, "large sweep")
        assert not m.errors, "\n".join(m.errors[:12])
