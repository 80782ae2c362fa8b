"""Pure-Python twin of oracle/plaac_oracle.c — TEST INFRASTRUCTURE ONLY, small cases only.

A second, independently written restatement of the same reference arithmetic (cli/src/plaac.java),
used by tests to cross-check the C oracle bit for bit (Python floats are IEEE doubles and CPython
never fuses a*b+c). Parity status is the oracle's: Viterbi pinned by the 28-domain KAT, floats unpinned.
"""
import math

NAN = float("nan")
NINF = float("-inf")
ALPHABET = "XACDEFGHIKLMNPQRSTVWY*"  # plaac.java:26


def aatoint(ch):  # plaac.java:1508-1534
    u = ch.upper()
    if u == "X":
        return 0
    k = ALPHABET.find(u)
    return k if k > 0 else 0


def encode(s):
    return [aatoint(ch) for ch in s]


def logeapeb(lut, a, b):  # plaac.java:1024-1047
    if a > b:
        c = a - b
        if not (c < 40):
            return a
        dex = int(math.floor(100 * c))
        return a + ((100 * c - dex) * lut[dex + 1] + (dex + 1 - 100 * c) * lut[dex])
    if b > a:
        c = b - a
        if not (c < 40):
            return b
        dex = int(math.floor(100 * c))
        return b + ((100 * c - dex) * lut[dex + 1] + (dex + 1 - 100 * c) * lut[dex])
    return a + math.log(2)


def fixed_window(seq, L):  # hss2 with min == max == L, plaac.java:1206-1257
    n = len(seq)
    if L > n:
        return -1, -2, NINF
    ps = [0.0]
    for v in seq:
        ps.append(ps[-1] + v)
    best, bs = ps[L], 0
    for i in range(L, n):
        s = i - L + 1
        d = ps[i + 1] - ps[s]
        if d > best:
            best, bs = d, s
    return bs, bs + L - 1, best


def mean_shrink(arr, ww, seq=None):  # slidingaverage(..., true, false[, 13, seq]) :2585-2662
    n = len(arr)
    w = ww // 2
    if w >= n:
        w = n - 1
    out = []
    for i in range(n):
        score = 0.0
        denom = 0.0
        for p in range(i - w, i + w + 1):
            if 0 <= p < n:
                denom = denom + 1.0
                if seq is not None and seq[p] == 13 and ((p >= 1 and seq[p - 1] == 13) or
                                                          (p >= 2 and seq[p - 2] == 13)):
                    continue
                score = score + 1.0 * arr[p]
        out.append(score / denom)
    return out


def mean_weighted(arr, ww):  # slidingaverage(..., false, true) :2585-2622
    n = len(arr)
    w = ww // 2
    if w >= n:
        w = n - 1
    out = [NAN] * n
    for i in range(w, n - w):
        score = 0.0
        denom = 0.0
        for p in range(i - w, i + w + 1):
            wt = 1.0 + min(p, w) + min(n - p - 1, w)
            denom = denom + wt
            score = score + wt * arr[p]
        out[i] = score / denom
    return out


def viterbi(h, x):  # plaac.java:3077-3121; h = dict(lt, li, le, lf) of nested lists
    n = len(x)
    s = [[h["li"][i] + h["le"][i][x[0]] for i in (0, 1)]]
    tb = [[0, 0]]
    for t in range(1, n):
        cur, arg = [], []
        for i in (0, 1):
            best, k = h["lt"][0][i] + s[-1][0], 0
            if h["lt"][1][i] + s[-1][1] > best:
                best, k = h["lt"][1][i] + s[-1][1], 1
            cur.append(best + h["le"][i][x[t]])
            arg.append(k)
        s.append(cur)
        tb.append(arg)
    best, k = s[-1][0] + h["lf"][0], 0
    if s[-1][1] + h["lf"][1] > best:
        best, k = s[-1][1] + h["lf"][1], 1
    path = [0] * n
    path[-1] = k
    for t in range(n - 2, -1, -1):
        path[t] = tb[t + 1][path[t + 1]]
    return path, best


def forward_backward(h, lut, x):  # plaac.java:3349-3411, :4032-4045
    n = len(x)
    a = [[h["li"][i] + h["le"][i][x[0]] for i in (0, 1)]]
    for t in range(1, n):
        cur = []
        for i in (0, 1):
            sc = NINF
            for k in (0, 1):
                sc = logeapeb(lut, sc, h["lt"][k][i] + a[-1][k])
            cur.append(sc + h["le"][i][x[t]])
        a.append(cur)
    tot = NINF
    for i in (0, 1):
        tot = logeapeb(lut, tot, a[-1][i] + h["lf"][i])
    b = [None] * n
    b[n - 1] = [h["lf"][0], h["lf"][1]]
    for t in range(n - 2, -1, -1):
        cur = []
        for i in (0, 1):
            sc = NINF
            for k in (0, 1):
                sc = logeapeb(lut, sc, h["lt"][i][k] + b[t + 1][k] + h["le"][k][x[t + 1]])
            cur.append(sc)
        b[t] = cur
    lp = NINF
    for i in (0, 1):
        lp = logeapeb(lut, lp, a[0][i] + b[0][i])
    pp = [[math.exp((a[t][i] + b[t][i]) - lp) for i in (0, 1)] for t in range(n)]
    mp = [1 if pp[t][1] > pp[t][0] else 0 for t in range(n)]
    return tot, pp, mp


def score_protein(P, x):
    """P: dict of plain Python tables (see tests/test_oracle.py::params_to_dict). x: codes after the stop trim.
    Returns (row dict, tracks dict) with the same field names as oracle_row / oracle_tracks."""
    n = len(x)
    c = P["corelength"]
    row = {"prot_len": n}
    s, e, sc = fixed_window([1.0 if k in (12, 14) else 0.0 for k in x], min(80, n))
    row.update(mw_score=int(sc), mw_start=s, mw_end=e)
    llrs = [P["llr"][k] for k in x]
    s, e, sc = fixed_window(llrs, c)
    row.update(llr_score=sc, llr_start=s, llr_end=e)
    vit, lv1 = viterbi(P["hmm1"], x)
    lm1, pp, mp = forward_backward(P["hmm1"], P["loglut"], x)
    _, lv0 = viterbi(P["hmm0"], x)
    lm0, _, _ = forward_backward(P["hmm0"], P["loglut"], x)
    row.update(hmm_all=lm1 - lm0, hmm_vit=lv1 - lv0)
    longest, cur = 0, 0
    for v in vit:
        cur = cur + 1 if v else 0
        longest = max(longest, cur)
    row["vit_maxrun"] = longest
    s, e, sc = fixed_window([llrs[i] if vit[i] == 1 else -1000000.0 for i in range(n)], c)
    if sc > -500000.0:
        a, z = s, e
        while a >= 0 and vit[a] == 1:
            a -= 1
        a += 1
        while z < n and vit[z] == 1:
            z += 1
        z -= 1
        prd = 0.0
        for k in range(a, z + 1):
            prd = prd + llrs[k]
        row.update(core_score=sc, core_start=s, core_end=e, prd_score=prd, prd_start=a, prd_end=z)
    else:
        row.update(core_score=NAN, core_start=-1, core_end=-2, prd_score=0.0, prd_start=-1, prd_end=-2)
    # disorderreport :4866-5068
    cc = P["cc"]
    hy = [P["hydro2"][k] for k in x]
    ch = [P["charge"][k] for k in x]
    mh = 0.0
    for v in hy:
        mh = mh + v
    mh = (1.0 * mh) / n
    mc = 0.0
    for v in ch:
        mc = mc + v
    mc = (1.0 * mc) / n
    hydro = mean_shrink(hy, P["ww1"])
    charge = mean_shrink(ch, P["ww1"])
    fi = [cc[0] * hydro[i] + cc[1] * abs(charge[i]) + cc[2] for i in range(n)]
    row.update(fi_meanhydro=mh, fi_meancharge=mc, fi_meancombo=cc[2] + cc[1] * abs(mc) + cc[0] * mh)
    pllr = mean_shrink(llrs, P["ww3"])
    papa = mean_shrink([P["lodpapa"][k] for k in x], P["ww2"], x if P["adjustprolines"] else None)
    papax2 = mean_weighted(papa, P["ww2"])
    pllrx2 = mean_weighted(pllr, P["ww3"])
    fix2 = mean_weighted(fi, P["ww1"])
    best, cen = NINF, -1
    for k in range((P["ww2"] - 1) // 2, n - (P["ww2"] - 1) // 2):
        if papax2[k] > best and fix2[k] < 0:
            best, cen = papax2[k], k
    row.update(papa_combo=best, papa_cen=cen, papa_prop=NAN, papa_fi=NAN, papa_llr=NAN, papa_llr2=NAN)
    if cen >= 0:
        row.update(papa_prop=papax2[cen], papa_fi=fix2[cen], papa_llr=pllr[cen], papa_llr2=pllrx2[cen])
    halfw = min((P["ww1"] - 1) // 2, n // 2)
    num, mx, i = 0, 0, halfw
    while i < n - halfw:
        if fi[i] < 0:
            st = i
            while i < n - halfw and fi[i] < 0:
                i += 1
            en = i - 1
            if st == halfw:
                st = 0
            if en == n - halfw - 1:
                en = n - 1
            if en - st + 1 >= 5:
                num += en - st + 1
                mx = max(mx, en - st + 1)
        else:
            i += 1
    row.update(fi_numaa=num, fi_maxrun=mx)
    tracks = dict(vit=vit, map=mp, charge=charge, hydro=hydro, fi=fi, plaacllr=pllr, papa=papa, fix2=fix2,
                  plaacllrx2=pllrx2, papax2=papax2, post0=[q[0] for q in pp], post1=[q[1] for q in pp])
    return row, tracks
