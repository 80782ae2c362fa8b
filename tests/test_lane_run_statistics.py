"""CPU check of the bit-parallel run statistics of k_tracksL (no GPU)."""


def test_run_statistics_of_sixteen_steps_at_once_equal_the_per_step_form():
    """k_tracksL (kernels_windows_lane.hip.inc, kl_runs16) takes the FoldIndex run statistics (plaac.java:5010-5059: residues in
    runs of five or more negative positions, longest such run) of sixteen steps from their sign bits with population counts
    and a binary search over doubled masks instead of nine instructions per step. The algorithm, restated here line by line,
    against the per-step form on random blocks with every kind of carry; the device code itself is covered by the GPU parity
    tests (fields fi_numaa / fi_maxrun)."""
    import random

    def per_step(flags, cur, numaa, maxlen):
        for f in flags:
            ended = 0 if f else cur
            cnt = ended if ended >= 5 else 0
            numaa += cnt
            maxlen = max(maxlen, cnt)
            cur = cur + 1 if f else 0
        return cur, numaa, maxlen

    def block(flags, cur, numaa, maxlen):
        r = sum(1 << i for i, f in enumerate(flags) if f)
        nz = ~r & 0xffff
        if nz == 0:
            return cur + 16, numaa, maxlen
        lead = (nz & -nz).bit_length() - 1
        len0 = cur + lead
        cnt0 = len0 if len0 >= 5 else 0
        tail = 16 - nz.bit_length()
        mid = r & ~((1 << lead) - 1) & ((1 << (16 - tail)) - 1)
        a = mid & (mid >> 1)
        b = a & (a >> 2)
        m5 = b & (mid >> 4)
        sp = m5 | (m5 << 1)
        sp |= sp << 2
        sp |= m5 << 4
        e2 = m5 & (m5 >> 1)
        e4 = e2 & (e2 >> 2)
        e8 = e4 & (e4 >> 4)
        length, y = 0, 0xffff
        for step, e in ((8, e8), (4, e4), (2, e2), (1, m5)):
            c = y & (e >> length)
            if c:
                y, length = c, length + step
        lint = length + 4 if m5 else 0
        return tail, numaa + bin(sp).count("1") + cnt0, max(maxlen, cnt0, lint)

    rng = random.Random(5010)
    for _ in range(60000):
        p = rng.choice([0.05, 0.3, 0.5, 0.7, 0.9, 0.98])
        flags = [rng.random() < p for _ in range(16)]
        state = (rng.choice([0, 0, 1, 3, 4, 5, 17, 2000]), rng.randrange(1000), rng.choice([0, 5, 9, 300]))
        assert per_step(flags, *state) == block(flags, *state), (flags, state)
    for flags in ([True] * 16, [False] * 16, [True] * 15 + [False], [False] + [True] * 15, [True] * 5 + [False] + [True] * 10):
        for cur in (0, 4, 5):
            assert per_step(flags, cur, 0, 0) == block(flags, cur, 0, 0)
