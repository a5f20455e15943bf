"""Index-logic model of the diagonal sweep (openvo_amd/csrc/sgbm_diag.inc) in numpy.

Not a test of the HIP code: it replays the kernel's decomposition -- strips of skewed columns
uu = x - y + H - 1, per-strip row ranges, own-column state / LDS row buffers / halo columns fed
through a boundary buffer, border state for pixels outside the image -- and compares the three
top-down path directions with a direct evaluation.  Run: python tools/sim_diag.py
"""
import numpy as np

MAXC = 32767


def step(C, Lp, P1, P2):
    """one path step: L(d) = C(d) + min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, min Lp + P2) - min Lp"""
    m = Lp.min()
    lm = np.concatenate(([MAXC], Lp[:-1])) + P1
    lp = np.concatenate((Lp[1:], [MAXC])) + P1
    return C + np.minimum(np.minimum(Lp, lm), np.minimum(lp, m + P2)) - m


def direct(C, P1, P2, rev):
    H, W, D = C.shape
    out = np.zeros((3, H, W, D), np.int64)
    rows = range(H - 1, -1, -1) if rev else range(H)
    dy = 1 if rev else -1
    for y in rows:
        for x in range(W):
            for k, dx in enumerate((-1, 0, 1)):
                xp, yp = x + dx, y + dy
                Lp = out[k, yp, xp] if (0 <= xp < W and 0 <= yp < H) else np.zeros(D, np.int64)
                out[k, y, x] = step(C[y, x], Lp, P1, P2)
    return out


def strips(C, P1, P2, rev, UW):
    H, W1, D = C.shape
    nstrips = -(-(W1 + H - 1) // UW)
    CW = UW + 2
    border = np.zeros(D, np.int64)
    out = np.zeros((3, H, W1, D), np.int64)
    bnd = {}                                   # (strip, row) -> (N[0], NE[0], NE[1])

    def rows_of(j):
        uu0 = j * UW
        return max(0, (H - 1) - (uu0 + UW - 1)), min(H, W1 + (H - 1) - uu0)

    for t in range(nstrips):                   # ticket order
        j = nstrips - 1 - t
        uu0 = j * UW
        ys, ye = rows_of(j)
        if ye <= ys:
            continue
        has_nb = j + 1 < nstrips
        nys, nye = rows_of(j + 1) if has_nb else (0, 0)
        buf = np.zeros((2, 2, CW, D), np.int64)   # [parity][N, NE][column][d], all border
        Lnw = np.zeros((UW, D), np.int64)

        def helper_import(r):
            need = has_nb and nys <= r < nye and r + 1 < ye
            p = r & 1
            if need:
                v = bnd[(j + 1, r)]
                buf[p, 0, UW], buf[p, 1, UW], buf[p, 1, UW + 1] = v
            else:
                buf[p, 0, UW] = border
                buf[p, 1, UW] = border
                buf[p, 1, UW + 1] = border

        helper_import(ys - 1)
        for y in range(ys, ye):
            if y - 1 >= ys:
                p1 = (y - 1) & 1
                bnd[(j, y - 1)] = (buf[p1, 0, 0].copy(), buf[p1, 1, 0].copy(), buf[p1, 1, 1].copy())
            helper_import(y)                    # (into buf[y & 1] halo: not read before the next row)
            par = y & 1
            yp = H - 1 - y if rev else y
            for sc in range(UW):
                x = uu0 + sc - (H - 1) + y
                live = 0 <= x < W1
                xc = min(max(x, 0), W1 - 1)
                Cv = C[yp, xc]
                a = step(Cv, Lnw[sc], P1, P2)
                b = step(Cv, buf[par ^ 1, 0, sc + 1], P1, P2)
                c = step(Cv, buf[par ^ 1, 1, sc + 2], P1, P2)
                if not live:
                    a = b = c = border
                buf[par, 0, sc] = b
                buf[par, 1, sc] = c
                Lnw[sc] = a
                if live:
                    out[0, yp, x], out[1, yp, x], out[2, yp, x] = a, b, c
        p1 = (ye - 1) & 1
        bnd[(j, ye - 1)] = (buf[p1, 0, 0].copy(), buf[p1, 1, 0].copy(), buf[p1, 1, 1].copy())
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for (H, W1, D, UW) in [(9, 13, 6, 4), (20, 7, 5, 8), (5, 40, 4, 12), (17, 17, 3, 4), (3, 3, 4, 28), (30, 11, 4, 8)]:
        C = rng.integers(0, 60, (H, W1, D)).astype(np.int64)
        for rev in (False, True):
            a = direct(C, 3, 11, rev)
            b = strips(C, 3, 11, rev, UW)
            assert np.array_equal(a, b), (H, W1, D, UW, rev)
    print("diagonal-sweep decomposition == direct evaluation")
