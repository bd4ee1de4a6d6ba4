"""The two exactness arguments the round-3 flank-extension engines rest on (tidehunter_amd/csrc/th_ksw.h: th_adx_bound /
th_ext_bound_w and th_ksw_ext_band_wave), checked on thousands of small random pairs against the plain full-matrix extension
(ksw_extz2_sse with KSW_EZ_EXTZ_ONLY, no band, no z-drop: src/ksw2_align.c:153-173 -- match 1, mismatch -2, gap 2 + 1 per base).
The model below restates the engine's pruning rules with the strip width as a parameter (the device uses 64), so that small matrices
exercise them: it is a check of the ARGUMENT (bounds, off-by-ones), the device code itself is checked by TH_ALN_MODE=2 on the GPU
and by tests/test_emul_parity.py."""
import random

Q, E, NEG = 2, 1, -(1 << 29)


def sc(a, b):
    return 1 if a == b else -2


def full_ext(q, t):
    """H of every cell (rows = t, columns = q), extension from the corner"""
    n, m = len(t), len(q)
    H = [[NEG] * m for _ in range(n)]
    Ecol = [-(Q + (j + 1) * E) - Q - E for j in range(m)]            # E entering row 0 of column j
    Hup = [-(Q + (j + 1) * E) for j in range(m)]                      # H(-1, j)
    for i in range(n):
        hleft, F = -(Q + (i + 1) * E), -(Q + (i + 1) * E) - Q - E     # H(i, -1) and F entering column 0
        hdiag = 0 if i == 0 else -(Q + i * E)                         # H(i-1, -1)
        for j in range(m):
            h = max(hdiag + sc(t[i], q[j]), Ecol[j], F)
            hdiag = Hup[j]
            Hup[j] = h
            Ecol[j] = max(Ecol[j], h - Q) - E
            F = max(F, h - Q) - E
            H[i][j] = h
    return H


def best_cells(H):
    mx = max(max(r) for r in H)
    if mx <= 0:
        return 0, None
    cells = [(i + j, i, j) for i, r in enumerate(H) for j, h in enumerate(r) if h == mx]
    r0 = min(c[0] for c in cells)
    return mx, sorted(c for c in cells if c[0] == r0)


def bound_w(W, qlen, tlen):
    d = W + 1                                                         # th_ext_bound_w
    return max(min(tlen - d, qlen), min(qlen - d, tlen)) - Q - d * E


def banded_ext(q, t, W):
    """cells with |i - j| > W are minus infinity (the borders keep their values)"""
    return pruned_ext(q, t, W, NEG, len(q) + len(t) + 8)


def pruned_ext(q, t, W, slb, S):
    """th_ksw_ext_band_wave with strips of S columns: strip j0 sweeps rows rs..re only -- the band |i - j| <= W and, once the top border
    has left the band (j0 > W), the rows that live cells of the parked column j0 - 1 can reach; everything else is minus infinity"""
    n, m = len(t), len(q)
    H = [[NEG] * m for _ in range(n)]
    parkH, parkF, prs, ppre = {}, {}, 0, -1            # column j0 - 1 of the previous strip: valid on its rows prs..ppre only
    for j0 in range(0, m, S):
        rs, re = max(0, j0 - W), min(n - 1, j0 + S - 1 + W)
        if j0 > W and slb > NEG:
            vmin, vmax, B = 1 << 30, -1, m - j0
            for row in range(prs, ppre + 1):
                h, A = parkH[row], n - 1 - row
                if h + min(A, B) >= slb:
                    vmin = min(vmin, row)
                    vmax = max(vmax, row + S + min(h + B - slb, (h + A - slb) >> 1))
            rs, re = max(rs, vmin), min(re, vmax)
        if rs > re:
            break
        cols = range(j0, min(j0 + S, m))
        rowH = {j: NEG for j in cols}                  # H of the row above (minus infinity above rs; the border above row 0)
        rowE = {j: NEG for j in cols}                  # E entering this row
        newH, newF = {}, {}
        for i in range(rs, re + 1):
            if i == 0:
                for j in cols:
                    rowH[j] = -(Q + (j + 1) * E)
                    rowE[j] = rowH[j] - Q - E
            if j0 == 0:
                left_h, F = -(Q + (i + 1) * E), -(Q + (i + 1) * E) - Q - E
                diag = 0 if i == 0 else -(Q + i * E)
            else:
                left_h, F = (parkH[i], parkF[i]) if prs <= i <= ppre else (NEG, NEG)
                diag = -(Q + j0 * E) if i == 0 else (parkH[i - 1] if prs <= i - 1 <= ppre else NEG)
            for j in cols:
                h = max(diag + sc(t[i], q[j]), rowE[j], F)
                diag = rowH[j]
                if abs(i - j) <= W:
                    rowH[j], rowE[j], F = h, max(rowE[j], h - Q) - E, max(F, h - Q) - E
                else:
                    rowH[j], rowE[j], F = NEG, NEG, NEG
                H[i][j] = rowH[j]
            newH[i], newF[i] = rowH[cols[-1]], F
        parkH, parkF, prs, ppre = newH, newF, rs, re
    return H


def mutate(rng, s, p):
    out = []
    for c in s:
        x = rng.random()
        if x < p / 3:
            out.append(rng.choice("ACGT"))
        elif x < 2 * p / 3:
            out.append(c); out.append(rng.choice("ACGT"))
        elif x < p:
            continue
        else:
            out.append(c)
    return "".join(out)


def test_band_bound_covers_every_cell_a_band_can_get_wrong():
    """th_ext_bound_w / th_adx_bound: every cell outside the band, and every cell whose banded H differs from the true one, has true
    H <= the bound -- so a banded maximum above the bound is the true one, with the true tie candidates"""
    rng = random.Random(7)
    for _ in range(150):
        base = "".join(rng.choice("ACGT") for _ in range(rng.randint(20, 70)))
        q, t = mutate(rng, base, 0.25) or "A", mutate(rng, base, 0.25) or "C"
        H = full_ext(q, t)
        for W in (2, 5, 9, 14, 22):
            Hb = banded_ext(q, t, W)
            bnd = bound_w(W, len(q), len(t))
            for i in range(len(t)):
                for j in range(len(q)):
                    assert Hb[i][j] <= H[i][j]
                    if abs(i - j) > W or Hb[i][j] != H[i][j]:
                        assert H[i][j] <= bnd, (q, t, W, i, j, H[i][j], bnd)


def test_pruned_strip_engine_is_exact_for_any_lower_bound():
    """th_ksw_ext_band_wave: with slb <= the true maximum (the true one, a looser one, the banded maximum of a narrow probe) and W from
    th_ext_bound_w(W) < slb, the pruned sweep reports the full matrix's maximum on the same earliest anti-diagonal cells"""
    rng = random.Random(11)
    n_pruned = 0
    for it in range(400):
        base = "".join(rng.choice("ACGT") for _ in range(rng.randint(30, 110)))
        p = rng.choice((0.08, 0.2, 0.35))
        q, t = mutate(rng, base, p) or "A", mutate(rng, base, p) or "C"
        if it % 5 == 0:
            t = t[: max(1, len(t) // 2)]
        if it % 7 == 0:
            q = q + "".join(rng.choice("ACGT") for _ in range(20))
        H = full_ext(q, t)
        mx, cells = best_cells(H)
        if mx <= 0:
            continue
        probe = best_cells(banded_ext(q, t, 4))[0]
        for S in (4, 8):
            for slb in {mx, max(1, mx - rng.randint(1, 6)), max(1, probe)}:
                W = S
                while W < len(q) + len(t) and bound_w(W, len(q), len(t)) >= slb:
                    W += S
                Hp = pruned_ext(q, t, W, slb, S)
                n_pruned += sum(1 for i in range(len(t)) for j in range(len(q)) if Hp[i][j] == NEG and abs(i - j) <= W)
                for i in range(len(t)):
                    for j in range(len(q)):
                        assert Hp[i][j] <= H[i][j]
                        if H[i][j] >= slb:
                            assert Hp[i][j] == H[i][j], (q, t, S, slb, W, i, j)
                assert best_cells(Hp) == (mx, cells), (q, t, S, slb, W)
    assert n_pruned > 10000          # the viability rule did prune inside the band


# ---------------------------------------------------------------- the second band of the unit alignments (th_ksw_global_band_wave, slb)
def live_new(h, a, b, slb):
    """the engine's liveness test of a parked cell (th_ksw.h, th_ksw_global_band_wave): a path may cross the parked column INSIDE an open
    horizontal gap, so only |a - b| further extension bases may be charged"""
    return h + min(a, b) - abs(a - b) * E >= slb


def live_round3(h, a, b, slb):
    """round 3's rule, one extension base too strict (kept to show that the structured cases below have teeth)"""
    d = abs(a - b)
    return h + min(a, b) - ((d + 1) * E if d else 0) >= slb


def global_dp(q, t, W, slb, S, live=live_new):
    """H of the global alignment's matrix (ksw2 with flag 0: src/ksw2_align.c:117-151) restricted as th_ksw_global_band_wave restricts it:
    diagonals lo..hi = min(0, D) - W .. max(0, D) + W and, with a lower bound slb of the optimum, strips of S columns that sweep only the
    rows live cells of the parked column can reach"""
    n, m = len(t), len(q)
    D = n - m
    lo, hi = min(0, D) - W, max(0, D) + W
    H = [[NEG] * m for _ in range(n)]
    parkH, parkF, prs, ppre = {}, {}, 0, -1
    for j0 in range(0, m, S):
        rs, re = max(0, j0 + lo), min(n - 1, j0 + S - 1 + hi)
        if slb > NEG and j0 + lo > 0:
            vmin, vmax, b = 1 << 30, -1, m - j0
            for row in range(prs, ppre + 1):
                h, a = parkH[row], n - 1 - row
                if live(h, a, b, slb):
                    vmin = min(vmin, row)
                    vmax = max(vmax, row + S + min(h + b - slb, (h + a - slb) >> 1))
            rs, re = max(rs, vmin), min(re, vmax)
        if rs > re:
            return None
        cols = range(j0, min(j0 + S, m))
        rowH = {j: NEG for j in cols}
        rowE = {j: NEG for j in cols}
        newH, newF = {}, {}
        for i in range(rs, re + 1):
            if i == 0:
                for j in cols:
                    inb = -1 - j >= lo
                    rowH[j] = -(Q + (j + 1) * E) if inb else NEG
                    rowE[j] = rowH[j] - Q - E if inb else NEG
            if j0 == 0:
                inb = i + 1 <= hi
                left_f = -(Q + (i + 1) * E) - Q - E if inb else NEG
                diag = 0 if i == 0 else (-(Q + i * E) if i <= hi else NEG)
                F = left_f
            else:
                F = parkF[i] if prs <= i <= ppre else NEG
                diag = (-(Q + j0 * E) if -j0 >= lo else NEG) if i == 0 else (parkH[i - 1] if prs <= i - 1 <= ppre else NEG)
            for j in cols:
                h = max(diag + sc(t[i], q[j]), rowE[j], F)
                diag = rowH[j]
                if lo <= i - j <= hi:
                    rowH[j], rowE[j], F = h, max(rowE[j], h - Q) - E, max(F, h - Q) - E
                else:
                    rowH[j], rowE[j], F = NEG, NEG, NEG
                H[i][j] = rowH[j]
            newH[i], newF[i] = rowH[cols[-1]], F
        parkH, parkF, prs, ppre = newH, newF, rs, re
    return H


def test_pruned_global_band_keeps_every_cell_an_optimal_path_can_use():
    """th_ksw_global_band_wave with slb: the corner's score is the full matrix's, no cell exceeds its true value, and every cell through
    which a path of score >= slb can still pass -- H + what is left to the corner, counted as the engine counts it -- keeps its true H"""
    rng = random.Random(23)
    n_pruned = 0
    for it in range(300):
        base = "".join(rng.choice("ACGT") for _ in range(rng.randint(30, 100)))
        p = rng.choice((0.08, 0.2, 0.3))
        q, t = mutate(rng, base, p), mutate(rng, base, p)
        if len(q) < 4 or len(t) < 4:
            continue
        n, m = len(t), len(q)
        full = global_dp(q, t, n + m, NEG, n + m)
        opt = full[n - 1][m - 1]
        narrow = global_dp(q, t, 3, NEG, n + m)[n - 1][m - 1]          # a banded score: a lower bound of the optimum
        for S in (4, 8):
            for slb in {opt, opt - rng.randint(1, 5), narrow}:
                if slb <= NEG // 2:
                    continue
                for W in (n + m, max(4, abs(n - m) + rng.randint(4, 20))):
                    ref = global_dp(q, t, W, NEG, n + m)              # the band alone
                    got = global_dp(q, t, W, slb, S)
                    if ref[n - 1][m - 1] < slb:
                        continue                                    # (this band does not hold a path of that score: the caller never asks for it)
                    assert got is not None and got[n - 1][m - 1] == ref[n - 1][m - 1], (q, t, S, slb, W)
                    for i in range(n):
                        for j in range(m):
                            assert got[i][j] <= ref[i][j]
                            a, b = n - 1 - i, m - 1 - j
                            if ref[i][j] + min(a, b) - abs(a - b) * E >= slb:
                                assert got[i][j] == ref[i][j], (q, t, S, slb, W, i, j)
                            elif got[i][j] == NEG and ref[i][j] > NEG:
                                n_pruned += 1
    assert n_pruned > 10000


# ---------------------------------------------------------------- structured cases, judged by the unpruned full matrix only
def plain_global(q, t):
    """corner score of the plain global affine alignment (match 1, mismatch -2, gap 2 + 1 per base), written independently of the strip
    model above: three full matrices, no band, no pruning"""
    n, m = len(t), len(q)
    Hm = [[NEG] * (m + 1) for _ in range(n + 1)]
    Em = [[NEG] * (m + 1) for _ in range(n + 1)]
    Fm = [[NEG] * (m + 1) for _ in range(n + 1)]
    Hm[0][0] = 0
    for j in range(1, m + 1):
        Hm[0][j] = Fm[0][j] = -(Q + j * E)
    for i in range(1, n + 1):
        Hm[i][0] = Em[i][0] = -(Q + i * E)
        for j in range(1, m + 1):
            Em[i][j] = max(Em[i - 1][j], Hm[i - 1][j] - Q) - E
            Fm[i][j] = max(Fm[i][j - 1], Hm[i][j - 1] - Q) - E
            Hm[i][j] = max(Hm[i - 1][j - 1] + sc(t[i - 1], q[j - 1]), Em[i][j], Fm[i][j])
    return Hm[n][m]


def device_flow(q, t, S, W1, live):
    """the device's sequence for one unit alignment (th_k_align.hip): first band W1 -> banded score s1 -> not certified -> W2 from s1 ->
    the pruned second band with slb = s1 -> certificate.  Returns (status, corner): 'certified' / 'uncertified' / 'empty'"""
    n, m = len(t), len(q)
    D = n - m
    aD = abs(D)

    def certified(h, W):
        gmin = 2 * W + 2 + aD
        return 2 * h > (m + n - gmin) - 4 * Q - 2 * gmin * E

    first = global_dp(q, t, W1, NEG, n + m)
    s1 = first[n - 1][m - 1]
    if s1 <= NEG // 2:
        return "noband", None
    if certified(s1, W1):
        return "first", s1
    g2 = (m + n - 2 * s1 - 4 * Q) // (1 + 2 * E) + 2
    W2 = (g2 - 2 - aD + 1) // 2 + 8
    got = global_dp(q, t, max(W2, W1 + 1), s1, S, live)
    if got is None:
        return "empty", None
    h = got[n - 1][m - 1]
    return ("certified" if certified(h, max(W2, W1 + 1)) else "uncertified"), h


def structured_pairs(rng, S):
    """one insertion or deletion run placed across a strip boundary, an exact suffix after it, a few substitutions before it: the
    optimal path crosses the parked column inside its open gap and the rest is all matches"""
    for _ in range(4000):
        k = rng.randint(1, 4)                                  # strip boundary the gap straddles
        pre = k * S + rng.randint(-3, 3)
        glen = rng.randint(1, 6)
        suf = rng.randint(3, 4 * S)
        base = [rng.choice("ACGT") for _ in range(max(2, pre) + suf)]
        t = list(base)
        q = list(base)
        for _s in range(rng.randint(0, 3)):
            x = rng.randrange(0, max(1, pre - 1))
            q[x] = rng.choice([c for c in "ACGT" if c != q[x]])
        ins = [rng.choice("ACGT") for _ in range(glen)]
        if rng.random() < 0.5:
            q[pre:pre] = ins                                   # insertion in the query: a horizontal gap across column pre
        else:
            t[pre:pre] = ins
        yield "".join(q), "".join(t)


def test_structured_gap_across_a_strip_boundary_judged_by_the_full_matrix():
    """the liveness test of the parked column must not charge an extension base a path that crosses inside an open
    horizontal gap never pays.  Oracle = the plain full matrix (plain_global), not the liveness formula."""
    rng = random.Random(5)
    bad_new = bad_r3 = n = n_second = 0
    for S in (4, 8):
        for q, t in structured_pairs(rng, S):
            opt = plain_global(q, t)
            n += 1
            # (a) slb equal to the optimum, a band that holds the optimal path (diagonals 0 .. D and three either side)
            if global_dp(q, t, 3, NEG, len(q) + len(t))[len(t) - 1][len(q) - 1] == opt:
                got = global_dp(q, t, 3, opt, S)
                assert got is not None and got[len(t) - 1][len(q) - 1] == opt, (q, t, S)
                old = global_dp(q, t, 3, opt, S, live_round3)
                bad_r3 += old is None or old[len(t) - 1][len(q) - 1] != opt
            # (b) the device's flow with a narrow first band
            for W1 in (1, 2):
                st, h = device_flow(q, t, S, W1, live_new)
                if st in ("certified", "uncertified", "empty"):
                    n_second += 1
                    assert st != "empty", (q, t, S, W1)
                    if st == "certified":
                        bad_new += h != opt
                        assert h == opt, (q, t, S, W1, h, opt)
    assert bad_new == 0
    assert n_second > 500            # the second, pruned band was exercised
    assert bad_r3 > 0                # ... and these cases do catch round 3's rule


def test_advice_example_pair():
    """the pair the round-3 advisor reported: S = 4, first band such that s1 = opt = 12, W2 = 10 -- round 3's rule returned 9 and passed
    the certificate"""
    q, t = "GCTATCGTCCAGGAAGGGACGATGGACAAGCATCG", "GCTATCTTCCAGCAAGGGATGATGGACCG"
    opt = plain_global(q, t)
    assert opt == 12
    got = global_dp(q, t, 10, opt, 4)
    assert got is not None and got[len(t) - 1][len(q) - 1] == opt
    old = global_dp(q, t, 10, opt, 4, live_round3)
    assert old is None or old[len(t) - 1][len(q) - 1] != opt
