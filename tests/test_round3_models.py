"""CPU: restatements, in plain Python, of the two index schemes round 3 added to the sort path — checked against the oracle's
item enumeration (Lv1FillOffsets / Lv2ExtractSubString of stage 1, reference src/sorting/read_to_sdbg_s1.cpp:208-366) and against
a stable sort (kmlib::kmsort's contract per radix level, src/kmlib/kmsort.h:45-122).  They follow the device code statement by
statement (megahit_amd/csrc/s1.hip: S1GenBlocked::get_unit, s1_item_from_window; sort_kernels.h: k_radix_onesweep_u), so that
the arithmetic — slot -> offset mapping, window words shared by a run of consecutive items, the read boundary inside a run,
the clamping of threads beyond the last item, the store's first two bases, positions inside a unit and the sliding window — is
pinned for read lengths and k the GPU tests do not visit (those test the compiled kernels: tests/test_gpu_round3_knobs.py,
tests/test_gpu_sort_unit_runs.py)."""
import numpy as np
import pytest

import oracle_binding as ob

M64 = (1 << 64) - 1
SENT = 4
NI, UT, WAVES = 8, 3, 4
TILE = 256 * NI


def funnel_l(hi, lo, sh):
    return ((hi << sh) | (lo >> (32 - sh))) & 0xFFFFFFFF if sh else hi


def rc64(x, n):
    r = int("{:064b}".format(x)[::-1], 2)
    r = ((r >> 1) & 0x5555555555555555) | ((r & 0x5555555555555555) << 1)
    return ((~r & M64) << (64 - 2 * n)) & M64


def comp_or_sentinel(c):
    return SENT if c == SENT else 3 - c


def item_from_window(win, q, forced, L, k, a):
    km1 = k - 1
    head_b, tail_b = (win >> 60) & 3, (win >> (58 - 2 * km1)) & 3
    f = ((win << 4) & M64) & ((M64 << (64 - 2 * km1)) & M64)
    rc = rc64(f, km1)
    head = head_b if q >= 1 else SENT
    tail = tail_b if q + k - 1 < L else SENT
    if forced >= 0:
        strand = forced
    else:
        strand = 1 if f > rc else (0 if f < rc else (0 if head <= 3 - tail else 1))
    key = (rc | (comp_or_sentinel(tail) << 3) | comp_or_sentinel(head)) if strand else (f | (head << 3) | tail)
    return (key >> 32, key & 0xFFFFFFFF, a & 0xFFFFFFFF)


def blocked_unit(seq, L, per, k, unit_base, n):
    """S1GenBlocked::get_unit for all 256 threads of a workgroup -> {item index: record}"""
    tile_q, tile_r = TILE // per, TILE % per
    qlast, jf = L - k + 1, L - k + 2
    out = {}
    for w in range(WAVES):
        for lane in range(64):
            g00 = unit_base + (w * 64 + lane) * NI
            r = g00 // per
            jt, bt = [g00 - r * per], [r * L]
            for t in range(1, UT):
                jn, bn = jt[t - 1] + tile_r, bt[t - 1] + tile_q * L
                if jn >= per:
                    jn -= per
                    bn += L
                jt.append(jn)
                bt.append(bn)
            for t in range(UT):
                if g00 + t * TILE >= n:
                    jt[t], bt[t] = 0, 0
            for t in range(UT):
                q0 = min(jt[t] - 1 if jt[t] > 0 else 0, qlast)
                a0 = bt[t] + q0
                b0 = a0 - 2 if a0 >= 2 else 0
                wc, wn = b0 >> 4, (bt[t] + L - 2) >> 4
                assert wn + 3 < len(seq) and wc + 3 < len(seq), "window words outside the padded store"
                c = [int(seq[wc + x]) for x in range(4)]
                nx = [int(seq[wn + x]) for x in range(4)]
                g0, j, base = g00 + t * TILE, jt[t], bt[t]
                for i in range(NI):
                    q = min(j - 1 if j > 0 else 0, qlast)
                    forced = j if j < 2 else (j - jf if j >= jf else -1)
                    a = base + q
                    b = a - 2 if a >= 2 else 0
                    down = 0 if a >= 2 else (2 - a) * 2
                    second = (b >> 4) != wc
                    assert (b >> 4) - wc in (0, 1), "a run of eight windows starts in at most two words"
                    sh = (b & 15) * 2
                    x0, x1, x2 = (c[1], c[2], c[3]) if second else (c[0], c[1], c[2])
                    win = ((funnel_l(x0, x1, sh) << 32) | funnel_l(x1, x2, sh)) >> down
                    if g0 + i < n:
                        out[g0 + i] = item_from_window(win, q, forced, L, k, a)
                    j += 1
                    if j == per:
                        j, base, c, wc = 0, base + L, list(nx), wn
    return out


@pytest.mark.parametrize("L,k,n_reads", [(100, 21, 150), (30, 21, 700), (60, 25, 90), (50, 29, 40), (36, 29, 600), (33, 29, 900), (150, 22, 20), (40, 15, 8)])
def test_blocked_generator_makes_the_oracles_items(L, k, n_reads):
    rng = np.random.default_rng(L * 100 + k)
    reads = [rng.integers(0, 4, size=L, dtype=np.uint8) for _ in range(n_reads)]
    reads[3] = reads[3] * 0                      # poly-A: palindromic (k-1)-mers never occur, equal strands do not either; hot keys
    reads[5][:] = np.tile(np.array([0, 3], dtype=np.uint8), L)[:L]   # AT repeats: palindromes at even k-1
    pkg = ob.Package(reads, reverse=True)
    want = ob.s1_items(pkg, k)                   # read-major, slot order: item g = read g // per, slot g % per
    per = L - k + 4
    assert per >= NI and want.shape == (n_reads * per, 4)
    n = want.shape[0]
    seq = np.concatenate([pkg.words(), np.zeros(40, dtype=np.uint32)])  # the device store is padded with >= 32 zero words
    got = {}
    n_units = (n + TILE * UT - 1) // (TILE * UT) + 1   # one unit beyond the input, as the rounded-up grid has them
    for u in range(n_units):
        got.update(blocked_unit(seq, L, per, k, u * TILE * UT, n))
    assert sorted(got) == list(range(n))
    pos = ((want[:, 2].astype(np.uint64) << np.uint64(32)) | want[:, 3].astype(np.uint64)) >> np.uint64(7)
    for g in range(n):
        assert got[g] == (int(want[g, 0]), int(want[g, 1]), int(pos[g])), "item %d (read %d, slot %d)" % (g, g // per, g % per)


@pytest.mark.parametrize("unit_n,skew", [(TILE * UT, False), (TILE * UT - 1, True), (TILE + 5, False), (1, False), (700, True), (TILE * 2, True)])
def test_unit_wide_positions_are_the_stable_order(unit_n, skew):
    """k_radix_onesweep_u: rank inside (tile, wave, digit) in (round, lane) order, starts of the cells by an exclusive scan over
    (digit, tile, wave), position = start + rank; the window rounds then write position p to g_off[digit] + p."""
    rng = np.random.default_rng(unit_n)
    dig = rng.choice([3, 3, 3, 7, 200], size=TILE * UT) if skew else rng.integers(0, 256, size=TILE * UT)
    cnt = np.zeros((UT, WAVES, 256), dtype=np.int64)
    rank = {}
    for t in range(UT):
        for j in range(NI):
            for tid in range(256):
                w, lane = tid // 64, tid % 64
                gi = t * TILE + w * (64 * NI) + j * 64 + lane
                if gi < unit_n:
                    rank[gi] = cnt[t, w, dig[gi]]
                    cnt[t, w, dig[gi]] += 1
    tot = cnt.sum(axis=(0, 1))
    start = np.concatenate([[0], np.cumsum(tot)[:-1]])
    cell = np.zeros_like(cnt)
    for d in range(256):
        run = start[d]
        for t in range(UT):
            for w in range(WAVES):
                cell[t, w, d] = run
                run += cnt[t, w, d]
    pos = np.full(unit_n, -1, dtype=np.int64)
    for gi in range(unit_n):
        t, r = divmod(gi, TILE)
        pos[gi] = cell[t, r // (64 * NI), dig[gi]] + rank[gi]
    assert sorted(pos) == list(range(unit_n))
    assert np.array_equal(np.argsort(pos), np.argsort(dig[:unit_n], kind="stable"))
    # window rounds: every position is staged in exactly one round, at p - round * TILE
    staged = np.zeros(unit_n, dtype=np.int64)
    for r in range(UT):
        lo = r * TILE
        if lo >= unit_n:
            break
        rel = pos - lo
        staged += (rel >= 0) & (rel < TILE)
    assert np.all(staged == 1)
