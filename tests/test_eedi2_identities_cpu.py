"""The arithmetic identities the EEDI2 kernels of round 3 rest on, checked on the CPU over their whole domains.

The GPU parity tests compare every scratch plane of a pass with the oracle on synthetic frames - content decides
which corners of an expression they reach.  The rewrites below replace an expression of the reference by a cheaper
one that is claimed to be the SAME function; each claim is restated here (numpy, the kernel's integer / float32
operations transcribed one for one) and tried on every input the kernels can form, or on all of a reduced domain.
Nothing here runs a kernel: csrc/eedi2.hip, csrc/eedi2_16.hip and csrc/eedi2_vote.h name the test that covers them.
"""
import itertools

import numpy as np

LIMLUT = [6, 6, 7, 7, 8, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 12, 12, 12,
          12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, -1, -1]            # eedi2.c:21-25


def test_filter_map_step_as_sign_arithmetic_8bit():
    """k_filter_map's fm_step (eedi2.hip): (|s-ref| > lim && s != 255) || (c == 255 && s == 255) || (|c-ref| > lim &&
    c != 255) is the sign of (lim-|s-ref| & s-255) | (lim-|c-ref| & c-255) | (509-c-s) - every s, c, ref and the limits
    a direction value can give (lim = max(2 |dir|, 48), |dir| <= 32)."""
    s = np.arange(256, dtype=np.int32)[:, None, None]
    c = np.arange(256, dtype=np.int32)[None, :, None]
    ref = np.arange(256, dtype=np.int32)[None, None, :]
    for lim in sorted({max(2 * abs(d), 48) for d in range(-32, 32)}):
        want = ((np.abs(s - ref) > lim) & (s != 255)) | ((c == 255) & (s == 255)) | ((np.abs(c - ref) > lim) & (c != 255))
        got = (((lim - np.abs(s - ref)) & (s - 255)) | ((lim - np.abs(c - ref)) & (c - 255)) | (509 - c - s)) < 0
        assert np.array_equal(want, got), lim


def test_filter_map_step_as_sign_arithmetic_16bit():
    """q_filter_map's qfm_step (eedi2_16.hip) at 10 and 12 bits: the same with peak = 2^depth - 1 and 2 peak - 1 - c - s;
    random samples plus the values around the peak and the limits."""
    rng = np.random.default_rng(7)
    for depth in (10, 12):
        peak, shift = (1 << depth) - 1, depth - 8
        n = 400_000
        edge = np.array([0, 1, peak - 1, peak, peak // 2, peak // 2 + 1], dtype=np.int64)
        s = np.concatenate([rng.integers(0, peak + 1, n), rng.choice(edge, n)])
        c = np.concatenate([rng.choice(edge, n), rng.integers(0, peak + 1, n)])
        ref = rng.integers(0, peak + 1, 2 * n)
        d = rng.integers(-(1 << (depth - 3)), 1 << (depth - 3), 2 * n)
        lim = np.maximum(np.abs(d) * 2, 12 << (2 + shift))
        want = ((np.abs(s - ref) > lim) & (s != peak)) | ((c == peak) & (s == peak)) | ((np.abs(c - ref) > lim) & (c != peak))
        got = (((lim - np.abs(s - ref)) & (s - peak)) | ((lim - np.abs(c - ref)) & (c - peak)) | (2 * peak - 1 - c - s)) < 0
        assert np.array_equal(want, got), depth


def test_filter_map_ranges_unified():
    """The four walk ranges of eedi2_template.c:565-620 as [max(-x, neg), min(w-x-1, pos)] above and
    [max(-x, -pos), min(w-x-1, -neg)] below, neg = min(dir, 0), pos = max(dir, 0), for every interior x."""
    for width in (3, 4, 9, 17, 40):
        for x in range(1, width - 1):
            for d in range(-8, 9):
                if d < 0:
                    top = (max(-x, d), 0)
                    bot = (0, min(width - x - 1, abs(d)))
                else:
                    top = (0, min(width - x - 1, d))
                    bot = (max(-x, -d), 0)
                neg, pos = min(d, 0), max(d, 0)
                assert top == (max(-x, neg), min(width - x - 1, pos))
                assert bot == (max(-x, -pos), min(width - x - 1, -neg))


def test_calc_directions_vote_limit_closed_form():
    """max(limlut[|mid|] >> 2, 2) == (|mid| >= 13 ? 3 : 2) for every |mid| the LDS search can produce (<= 30)."""
    for m in range(31):
        assert max(LIMLUT[m] >> 2, 2) == (3 if m >= 13 else 2)


def test_vote_average_is_an_integer_floor():
    """(int)((float)a / (float)b + 0.5f) == (2a + b) // (2b) for a = sum + mid <= 2559, b = count + 1 <= 10
    (csrc/eedi2_vote.h; the device side of it, v_rcp_f32 included, runs in tests/test_eedi2_gpu.py)."""
    a = np.arange(2560, dtype=np.int64)
    for b in range(1, 11):
        ref = (a.astype(np.float32) / np.float32(b) + np.float32(0.5)).astype(np.int32)
        assert np.array_equal(ref, (2 * a + b) // (2 * b)), b
        # the reciprocal form with a reciprocal off by up to two units in the last place, and its one correction
        n, m = (2 * a + b).astype(np.float32), np.float32(2 * b)
        r0 = np.float32(1.0) / m
        for ulp in (-2, -1, 0, 1, 2):
            r = r0
            for _ in range(abs(ulp)):
                r = np.nextafter(r, np.float32(np.inf if ulp > 0 else -np.inf))
            q = (n * r).astype(np.int64)
            q += ((2 * a + b) - q * (2 * b)) >= 2 * b
            assert np.array_equal(q, (2 * a + b) // (2 * b)), (b, ulp)


def _sorted_mid(v):
    """eedi2.c:65-80 on the present values."""
    v = sorted(v)
    n = len(v)
    return v[n >> 1] if n & 1 else (v[(n - 1) >> 1] + v[n >> 1] + 1) >> 1


def test_midpoint_selection_and_unsorted_vote():
    """mid9's two middle entries off the same two comparisons (n <= 5, n <= 7), absent slots as a value above every
    present one; and the vote (a sum and a count of the slots within lim of the midpoint) taken on the slots as they
    are instead of the sorted ones."""
    rng = np.random.default_rng(3)
    ABSENT = 1000
    for _ in range(20000):
        n = int(rng.integers(4, 10))
        present = [int(x) for x in rng.integers(0, 255, n)]
        slots = present + [ABSENT] * (9 - n)
        rng.shuffle(slots)
        v = sorted(slots)
        n5, n7 = n <= 5, n <= 7
        hi = v[2] if n5 else (v[3] if n7 else v[4])
        lo = v[1] if n5 else (v[2] if n7 else v[3])
        mid = hi if n & 1 else (lo + hi + 1) >> 1
        assert mid == _sorted_mid(present)
        lim = int(rng.integers(0, 13))
        in_sorted = [x for x in v if abs(x - mid) <= lim]
        in_slots = [x for x in slots if abs(x - mid) <= lim]
        assert (sum(in_sorted), len(in_sorted)) == (sum(in_slots), len(in_slots))
        assert ABSENT not in in_slots


def _compose(later, earlier):
    """lr_compose (eedi2.hip): the 2-state map that applies `earlier` first; bit s = outcome for incoming state s."""
    return ((later >> (earlier & 1)) & 1) | (((later >> ((earlier >> 1) & 1)) & 1) << 1)


def test_lattice_resolve_scan_equals_the_serial_walk():
    """k_lattice_resolve: the outcome of a pixel depends on its left neighbour's outcome only, so a row is a chain of
    2-state maps.  The kernel composes the maps of a thread's four pixels, scans the threads' maps inside a wave,
    chains the waves and carries the last outcome into the next pass; resolved that way a row must come out as it does
    walked pixel by pixel."""
    rng = np.random.default_rng(11)
    T, PX, W = 256, 4, 64                                     # threads per workgroup, pixels per thread, wave size
    for width in (1, 3, 4, 5, 63, 64, 255, 960, 1023, 1024, 1025, 1920, 2500):
        maps = rng.integers(0, 4, width)
        # the serial walk: pixel 0's incoming state is irrelevant for the kernel (both bits equal there); force that
        maps[0] = 3 * int(rng.integers(0, 2))
        state, serial = 0, []
        for m in maps:
            state = (int(m) >> state) & 1
            serial.append(state)
        out, carry = [], 0
        for x0 in range(0, width, T * PX):
            pm = np.zeros((T, PX), dtype=np.int64)
            for t in range(T):
                for k in range(PX):
                    x = x0 + PX * t + k
                    m = int(maps[x]) if x < width else 0
                    pm[t, k] = m if k == 0 else _compose(m, int(pm[t, k - 1]))
            tm = pm[:, PX - 1].copy()
            for w in range(T // W):                            # inclusive scan inside each wave
                seg = tm[w * W:(w + 1) * W]
                off = 1
                while off < W:
                    prev = seg.copy()
                    for lane in range(off, W):
                        seg[lane] = _compose(int(prev[lane]), int(prev[lane - off]))
                    off <<= 1
            win, state = [], carry
            for w in range(T // W):
                win.append(state)
                state = (int(tm[w * W + W - 1]) >> state) & 1
            for t in range(T):
                lane, w = t % W, t // W
                before = 2 if lane == 0 else int(tm[t - 1])
                sin = (before >> win[w]) & 1
                for k in range(PX):
                    x = x0 + PX * t + k
                    if x < width:
                        o = (int(pm[t, k]) >> sin) & 1
                        out.append(o)
                        if x == min(x0 + T * PX, width) - 1:
                            carry = o
        assert out == serial, width


def test_row_pairs_cover_every_row_once():
    """The _2x passes with a thread row per PAIR of rows (2r, 2r + 1): the row with the rebuilt rows' parity goes through
    the pass (or is copied when it is outside y0 .. height - 2), the other is copied - every row of the plane exactly
    once, the rebuilt ones exactly the reference's (y0, y0 + 2, ... < height - 1), for both field parities."""
    for height, tff in itertools.product((2, 3, 4, 5, 8, 9, 270, 271, 540, 1080, 1081), (0, 1)):
        y0 = 2 - tff
        work, copied = [], []
        for r in range((height + 1) // 2):
            y, yc = 2 * r + (y0 & 1), 2 * r + 1 - (y0 & 1)
            if yc < height:
                copied.append(yc)
            if y < height:
                (work if y0 <= y < height - 1 else copied).append(y)
        assert sorted(work + copied) == list(range(height))
        assert work == list(range(y0, height - 1, 2))


def test_limlut_closed_form():
    """csrc/eedi2.hip:limlut2 - the 8-bit limlut (eedi2.c:21-25 cast to `pixel`: its two -1 entries read 255) as
    min(12, 6 + ((i - [i >= 8]) >> 1)) below 31 and 255 from there on, with [a < b] taken as the borrow bit of a
    16-bit difference (both operands below 2^15)."""
    table = [6, 6, 7, 7, 8, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
             12, -1, -1]
    lt = lambda a, b: ((a - b) & 0xffff) >> 15
    for i in range(0, 8192):                                   # |mid - 128| >> 2 of any 15-bit midpoint
        g = lt(7, i)
        lim = max(min(((i - g) >> 1) + 6, 12), lt(30, i) * 255)
        want = (table[i] & 0xff) if i < 33 else 255
        assert lim == want, i
    # the indicator itself: exact for operands below 2^15
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1 << 15, 20000); b = rng.integers(0, 1 << 15, 20000)
    assert np.array_equal(((a - b) & 0xffff) >> 15, (a < b).astype(np.int64))


def test_pair_vote_slots():
    """csrc/eedi2.hip:dir_map_pair - an absent slot (peak 255 pushed to 0x7fff) sorts behind every value, never comes
    within any limit (also 255) of a midpoint of present values, and the midpoint selection by the number of absent
    slots picks the entries mid9 picks."""
    for mid in range(0, 255):
        for lim in (6, 12, 255):
            assert abs(0x7fff - mid) > lim
    for n in range(4, 10):
        absent = 9 - n
        m5, m7, odd = absent >= 4, absent >= 2, (absent & 1) == 0
        hi = 2 if m5 else (3 if m7 else 4)
        lo = 1 if m5 else (2 if m7 else 3)
        assert hi == n >> 1 and (odd == bool(n & 1)) and (odd or lo == (n - 1) >> 1)


def test_lattice_search_steps_side_by_side():
    """lattice_stage_b_win (eedi2.hip, eedi2_16.hip): the reference's loop over u (eedi2.c:1249-1290) accepts a step when its diff
    is below the last accepted one and its other tests pass - tests that never look at what an earlier step left behind - so
    the step it ends on is the first one with the smallest diff among those that pass: the minimum of (diff << 3) | step."""
    rng = np.random.default_rng(11)
    nt8 = 160
    for _ in range(20000):
        diff = rng.integers(0, 200, 5)
        ok = rng.integers(0, 2, 5).astype(bool)
        mn, last = nt8, -1
        for i in range(5):                                   # the loop as the reference writes it
            if diff[i] < mn and ok[i]:
                mn, last = diff[i], i
        keys = [(int(diff[i]) << 3) | i if (ok[i] and diff[i] < nt8) else 0x7fffffff for i in range(5)]
        best = min(keys)
        assert (last == -1) == (best == 0x7fffffff)
        if last >= 0:
            assert best & 7 == last and best >> 3 == mn


def test_lattice_half_steps_are_constants_of_the_step():
    """Counted from the even number at or below dir - 2 (de - 2 with de = dir & ~1: six steps, the first or the last outside
    dir -+ 2), step i has u = de - 2 + i, u >> 1 = hb + (i >> 1) and (u + 1) >> 1 = hb + ((i + 1) >> 1) with hb = (de >> 1) - 1;
    the windows' sample positions follow: x - u - 1 sits 5 - i behind x - de - 4, x + u - 1 sits i behind x + de - 3."""
    for d in range(-32, 33):
        de, par = d & ~1, d & 1
        hb = (de >> 1) - 1
        active = []
        for i in range(6):
            u = de - 2 + i
            assert u >> 1 == hb + (i >> 1) and (u + 1) >> 1 == hb + ((i + 1) >> 1)
            if (i == 0 and par) or (i == 5 and not par):
                continue
            active.append(u)
            x = 1000
            assert (x - u - 1) - (x - de - 4) == 5 - i and (x + u - 1) - (x + de - 3) == i
            assert (x + (u >> 1) - 1) - (x + hb - 1) == i >> 1 and (x - (u >> 1) - 1) - (x - hb - 4) == 3 - (i >> 1)
            assert (x + ((u + 1) >> 1)) - (x + hb) == (i + 1) >> 1 and (x - ((u + 1) >> 1)) - (x - hb - 3) == 3 - ((i + 1) >> 1)
        assert active == list(range(d - 2, d + 3))


def test_within_limit_as_one_unsigned_compare():
    """|a - b| <= lim as (a - b + lim) <= 2 lim in unsigned 32-bit arithmetic (the lattice search's pair tests): every pair of
    8-bit values and limit of the table, and 12-bit samples with the 16-bit table's limits (its -1 entries are 65532 / 65520)."""
    a = np.arange(256, dtype=np.int64)[:, None]
    b = np.arange(256, dtype=np.int64)[None, :]
    for lim in sorted(set(l & 0xff for l in LIMLUT)):
        got = ((a - b + lim) & 0xffffffff) <= 2 * lim
        assert np.array_equal(got, np.abs(a - b) <= lim), lim
    rng = np.random.default_rng(3)
    for shift in (2, 4):
        peak = (256 << shift) - 1
        a = rng.integers(0, peak + 2, 300_000)
        b = rng.integers(0, peak + 2, 300_000)
        for l in sorted(set(LIMLUT)):
            lim = ((l & 0xffff) << shift) & 0xffff
            got = ((a - b + lim) & 0xffffffff) <= 2 * lim
            assert np.array_equal(got, np.abs(a - b) <= lim), (shift, lim)


def test_a_direction_value_can_exceed_the_peak_at_16_bits():
    """Directions are stored as neutral + (dir << (2 + shift)) in the sample type (eedi2.c:1289 and its siblings).  With
    |dir| <= 32 the 8-bit value wraps (128 + 128 = 0); the 10 / 12-bit one does not: it is peak + 1 - so "is the peak" is
    an equality in every 16-bit kernel (dir_map_pair16, lat16_near8), never `< peak`."""
    assert (128 + (32 << 2)) & 0xff == 0
    for depth in (10, 12):
        shift, peak, neutral = depth - 8, (1 << depth) - 1, 1 << (depth - 1)
        assert (neutral + (32 << (2 + shift))) & 0xffff == peak + 1


def test_lapsharp_mix_as_one_float_multiply():
    """lap_float_mix (sharpen.hip): (((double)sum * coef) - centre) * strength, truncated (lapsharp.c:174-175), equals
    trunc((float)(sum - k centre) * c) for a float c next to coef * strength - for every (sum, centre) the 3x3 taps can
    produce - at the strengths of the presets; for others (0.35, 0.7) no such float exists and the kernel keeps the doubles.
    Same search as the host code runs at init, in numpy."""
    def float_form(taps, coef, strength):
        ki = round(1 / coef)
        lo = sum(255 * t for t in taps if t < 0)
        hi = sum(255 * t for t in taps if t > 0)
        acc = np.arange(lo, hi + 1, dtype=np.int64)[:, None]
        cen = np.arange(256, dtype=np.int64)[None, :]
        ref = np.trunc(((acc.astype(np.float64) * coef) - cen.astype(np.float64)) * strength).astype(np.int64)
        q = (acc - ki * cen).astype(np.float32)
        s = np.float32(coef * strength)
        for c in (s, np.nextafter(s, np.float32(np.inf)), np.nextafter(s, np.float32(-np.inf))):
            if np.array_equal(np.trunc(q * c).astype(np.int64), ref):
                return True
        return False
    lap = [0, -1, 0, -1, 5, -1, 0, -1, 0]
    iso = [-1, -4, -1, -4, 25, -4, -1, -4, -1]
    for st in (0.2, 0.3, 0.04, 0.15, 0.5, 1.0, 1.5):
        assert float_form(lap, 1.0, st) and float_form(iso, 0.2, st), st
    assert not float_form(iso, 0.2, 0.35) and not float_form(lap, 1.0, 0.7)


def test_edge_mask_flatness_and_iy_are_one_range():
    """mask_tile / qmask_tile (eedi2.hip, eedi2_16.hip): build_edge_mask's flatness test of a column - all three pairwise
    differences of the samples above / at / below below ten (eedi2_template.c:141-150) - is max - min < ten, and Iy, the
    largest pairwise difference (:172-173), is that same max - min.  Every triple of 8-bit samples; 10-bit triples on a
    grid that holds every residue and both ends."""
    p = np.arange(256, dtype=np.int32)[:, None, None]
    c = np.arange(256, dtype=np.int32)[None, :, None]
    n = np.arange(256, dtype=np.int32)[None, None, :]
    rng = np.maximum(np.maximum(p, c), n) - np.minimum(np.minimum(p, c), n)
    assert np.array_equal((np.abs(p - c) < 10) & (np.abs(c - n) < 10) & (np.abs(p - n) < 10), rng < 10)
    assert np.array_equal(np.maximum(np.maximum(np.abs(p - n), np.abs(p - c)), np.abs(c - n)), rng)
    g = np.unique(np.concatenate([np.arange(0, 1024, 7), np.arange(0, 48), np.arange(976, 1024)])).astype(np.int32)
    p, c, n = g[:, None, None], g[None, :, None], g[None, None, :]
    rng = np.maximum(np.maximum(p, c), n) - np.minimum(np.minimum(p, c), n)
    for shift in (2, 4):                                                  # `ten` = 10 << (depth - 8), Iy >> shift
        ten = 10 << shift
        assert np.array_equal((np.abs(p - c) < ten) & (np.abs(c - n) < ten) & (np.abs(p - n) < ten), rng < ten)
        assert np.array_equal(np.maximum(np.maximum(np.abs(p - n), np.abs(p - c)), np.abs(c - n)) >> shift, rng >> shift)


def test_edge_mask_laplacian_as_two_unsigned_sads():
    """mask_tile (eedi2.hip, 8-bit): |Ixx| + |Iyy| (:183-186) with Ixx = C0 - 2 C1 + C2 and Iyy = P1 - 2 C1 + N1 is
    |(C0 + C2) - 2 C1| + |(P1 + N1) - 2 C1| on non-negative operands - two v_sad_u32 -, and P1 + N1 is the column sum
    the variance test already has, less C1."""
    r = np.random.default_rng(5)
    c0, c1, c2, p1, n1 = (r.integers(0, 256, 1 << 20, dtype=np.int64) for _ in range(5))
    for arr in (c0, c1, c2, p1, n1):
        arr[:4096] = r.choice([0, 255], 4096)                              # the corners
    want = np.abs(c0 - 2 * c1 + c2) + np.abs(p1 - 2 * c1 + n1)
    cs = p1 + c1 + n1
    sad = lambda a, b, acc: np.abs(a.astype(np.uint32).astype(np.int64) - b.astype(np.uint32).astype(np.int64)) + acc
    got = sad(c0 + c2, 2 * c1, sad(cs - c1, 2 * c1, 0))
    assert np.array_equal(want, got)
