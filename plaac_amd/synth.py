"""Seeded synthetic proteomes for tests and bench.py (BASELINE.md §4 / SURVEY.md §8(d) M2).

Residues are sampled from the generative 2-state prion/background HMM itself (transition and
initial probabilities of prionhmm1, plaac.java:970-971; emissions = the default fg/bg tables),
plus 0.05 % 'X', so that ~5 % of residues are prion-like and cores / parses are non-trivial.
The length laws are THIS repo's definition of the BASELINE.json configs, not reference data.
"""
import math

import numpy as np

SEED0 = 20140512

# inverse-CDF knots of the yeast-proteome length law (config 2)
_YEAST_Q = np.array([0.0, 0.05, 0.25, 0.50, 0.75, 0.95, 0.99, 1.0])
_YEAST_L = np.array([16, 95, 238, 407, 630, 1220, 1796, 4910], dtype=np.float64)

CONFIG_NPROT = {1: 1, 2: 5880, 3: 20600, 4: 10_000_000, 5: 10_000_000}

# probability of: flip state | force state 0 | keep, as thresholds on one uniform (see _states)
_P01 = 0.1 / 100   # background -> PrD
_P10 = 2.0 / 100   # PrD -> background
_PINIT1 = 0.0476
_PX = 0.0005


def lengths(config, nprot, rng):
    """protein lengths (int64) for a BASELINE config index"""
    if config == 2:
        u = rng.random(nprot)
        return np.maximum(16, np.rint(np.interp(u, _YEAST_Q, _YEAST_L))).astype(np.int64)
    if config == 3:
        ln = np.exp(rng.normal(math.log(415.0), 0.85, nprot))
        out = np.clip(np.rint(ln), 25, 35000).astype(np.int64)
        if nprot > 0:
            out[nprot // 2] = 34350  # one titin-sized record
        return out
    if config in (4, 5):
        ln = np.exp(rng.normal(math.log(210.0), 0.80, nprot))
        out = np.clip(np.rint(ln), 11, 8192).astype(np.int64)
        tail = rng.random(nprot) < 1e-4
        out[tail] = rng.integers(8193, 36001, int(tail.sum()))
        return out
    raise ValueError("config must be 2..5")


def _emission_cdfs(fg, bg):
    """cumulative emission tables over codes 1..20 for state 0 (bg) and state 1 (fg)"""
    c0 = np.cumsum(np.asarray(bg, dtype=np.float64)[1:21])
    c1 = np.cumsum(np.asarray(fg, dtype=np.float64)[1:21])
    return c0 / c0[-1], c1 / c1[-1]


def residues(lens, fg, bg, rng, stop_fraction=0.0):
    """codes u8 + offsets u64 for the given lengths. stop_fraction of the records get a trailing '*'."""
    lens = np.asarray(lens, dtype=np.int64)
    nprot = len(lens)
    stops = rng.random(nprot) < stop_fraction if stop_fraction > 0 else np.zeros(nprot, dtype=bool)
    rec = lens + stops
    offsets = np.zeros(nprot + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(rec)
    total = int(offsets[-1])
    first = np.zeros(total, dtype=bool)
    first[offsets[:-1][rec > 0].astype(np.int64)] = True
    # state chain: one uniform per residue encodes a map {flip, reset-to-0, keep}; maps compose associatively:
    # state = (state at the last reset) xor (parity of flips since). Record starts are resets to the initial draw.
    u = rng.random(total)
    flip = u < _P01                       # flips from either state (P01 < P10)
    reset = (u >= _P01) & (u < _P10)      # PrD -> background, background stays
    init1 = rng.random(total) < _PINIT1
    flip = flip & ~first
    reset = reset | first
    idx = np.arange(total, dtype=np.int64)
    last_reset = np.maximum.accumulate(np.where(reset, idx, -1))
    cf = np.cumsum(flip, dtype=np.int64)
    base = np.where(first, init1, False)
    state = (base[last_reset] ^ (((cf - cf[last_reset]) & 1) == 1))
    c0, c1 = _emission_cdfs(fg, bg)
    v = rng.random(total)
    codes = np.where(state, np.searchsorted(c1, v, side="right"), np.searchsorted(c0, v, side="right")) + 1
    codes = np.minimum(codes, 20).astype(np.uint8)
    codes[rng.random(total) < _PX] = 0
    if stops.any():
        codes[(offsets[1:][stops] - 1).astype(np.int64)] = 21
    return codes, offsets


def make_batch(config, nprot=None, fg=None, bg=None, seed=None, stop_fraction=0.0):
    """numpy batch for a BASELINE config (2..5). fg/bg: 22-vectors (default tables if None)."""
    from . import native
    if fg is None or bg is None:
        P = native.make_params()
        fg = np.array(P.fg) if fg is None else fg
        bg = np.array(P.bg) if bg is None else bg
    nprot = CONFIG_NPROT[config] if nprot is None else nprot
    rng = np.random.default_rng(SEED0 + config if seed is None else seed)
    lens = lengths(config, nprot, rng)
    return residues(lens, fg, bg, rng, stop_fraction)


def make_batch_torch(config, nprot, fg, bg, device, seed=None, max_len=None):
    """Same generator on a torch device (bench.py builds its multi-hundred-megabyte batch in HBM).
    Returns (codes uint8[total], offsets int64[nprot+1]) as torch tensors on `device`.
    max_len: clip every length (config 4 / 5 with max_len = 8192: the same length law without its 0.01 % tail of 8,193 -
    36,000-residue records - a proteome no single chain bounds; the random draws stay the same, only the lengths change)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(SEED0 + config if seed is None else seed)
    f64 = torch.float64
    if config == 2:
        u = torch.rand(nprot, generator=g, device=device, dtype=f64)
        q = torch.tensor(_YEAST_Q, device=device)
        ln = torch.tensor(_YEAST_L, device=device)
        k = torch.clamp(torch.bucketize(u, q, right=True) - 1, 0, len(_YEAST_Q) - 2)
        frac = (u - q[k]) / (q[k + 1] - q[k])
        lens = torch.clamp(torch.round(ln[k] + frac * (ln[k + 1] - ln[k])), min=16).to(torch.int64)
    elif config == 3:
        z = torch.randn(nprot, generator=g, device=device, dtype=f64)
        lens = torch.clamp(torch.round(torch.exp(math.log(415.0) + 0.85 * z)), 25, 35000).to(torch.int64)
        lens[nprot // 2] = 34350
    elif config in (4, 5):
        z = torch.randn(nprot, generator=g, device=device, dtype=f64)
        lens = torch.clamp(torch.round(torch.exp(math.log(210.0) + 0.80 * z)), 11, 8192).to(torch.int64)
        tail = torch.rand(nprot, generator=g, device=device) < 1e-4
        tl = torch.randint(8193, 36001, (nprot,), generator=g, device=device)
        lens = torch.where(tail, tl, lens)
    else:
        raise ValueError("config must be 2..5")
    if max_len is not None:
        lens = torch.clamp(lens, max=int(max_len))
    offsets = torch.zeros(nprot + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(lens, 0)
    total = int(offsets[-1].item())
    first = torch.zeros(total, dtype=torch.bool, device=device)
    first[offsets[:-1]] = True
    u = torch.rand(total, generator=g, device=device)
    flip = (u < _P01) & ~first
    reset = ((u >= _P01) & (u < _P10)) | first
    del u
    init1 = (torch.rand(total, generator=g, device=device) < _PINIT1) & first
    # position of the last reset at or before every residue (a record's first residue is one): the reset positions, indexed by
    # the running count of resets - a prefix sum and a gather. (torch.cummax over the int64 positions gave the same values at
    # 1.2 s per 1.25 M-sequence piece on MI355X: 9 of the 12 s bench.py took to build its 10 M-sequence batch.)
    last_reset = torch.nonzero(reset).reshape(-1)[torch.cumsum(reset.to(torch.int32), 0, dtype=torch.int32).to(torch.int64) - 1]
    del reset
    cf = torch.cumsum(flip.to(torch.int32), 0)
    del flip
    state = init1[last_reset] ^ (((cf - cf[last_reset]) & 1) == 1)
    del cf, last_reset, init1
    c0, c1 = _emission_cdfs(fg, bg)
    c0 = torch.tensor(c0, device=device, dtype=torch.float32)
    c1 = torch.tensor(c1, device=device, dtype=torch.float32)
    v = torch.rand(total, generator=g, device=device)
    codes = torch.where(state, torch.bucketize(v, c1, right=True), torch.bucketize(v, c0, right=True)) + 1
    del state, v
    codes = torch.clamp(codes, max=20).to(torch.uint8)
    codes[torch.rand(total, generator=g, device=device) < _PX] = 0
    return codes, offsets
