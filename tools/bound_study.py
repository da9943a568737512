#!/usr/bin/env python3
"""CPU study (container, numpy): how selective would a logarithm-free FIRST-level bound of the BOOST measure be on
the bench's own input (exact-count rows with the affy6 MAF spectrum, simulated phenotype)?

  measure = 2 ln 2 [ A + K_i + K_j + n lg tao ],   A = sum_cells (x lg x + y lg y - s lg s) = - sum_cells s h(x / s)

h = binary entropy in bits, x / y = case / control count of a cell, s = x + y.  With p0 = n_case / n and the
quadratic q(p) = h(p0) + h'(p0)(p - p0) - c (p - p0)^2 <= h(p) on [0, 1] (c = the smallest such constant),
sum x = p0 n gives   A <= -n h(p0) + c sum_cells (x - p0 s)^2 / s   -- nine reciprocals, no logarithm.
Prints the fraction of pairs whose first-level bound reaches the threshold (they would go on to the exact bound) and
the share of 64-pair groups (one per wavefront pass) holding at least one such pair.

Uses the oracle's generator (test infrastructure) for the rows; nothing here is part of the product."""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402


def h2(p):
    p = np.clip(p, 1e-300, 1.0)
    q = np.clip(1.0 - p, 1e-300, 1.0)
    return -(p * np.log2(p) + q * np.log2(q))


def quad_constant(p0: float) -> float:
    """smallest c with h(p0) + h'(p0)(p - p0) - c (p - p0)^2 <= h(p) on [0, 1] (dense grid + both ends + the limit)"""
    h0 = float(h2(np.array([p0]))[0])
    h1 = math.log2((1.0 - p0) / p0)
    p = np.concatenate([np.linspace(0.0, 1.0, 2_000_001), [0.0, 1.0]])
    p = p[np.abs(p - p0) > 1e-7]
    ratio = (h0 + h1 * (p - p0) - h2(p)) / (p - p0) ** 2
    lim = 1.0 / (2.0 * math.log(2.0) * p0 * (1.0 - p0))
    return max(float(ratio.max()), lim), int(np.argmax(ratio)), lim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=2000)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=20240607)
    ap.add_argument("--threshold", type=float, default=30.0)
    a = ap.parse_args()
    n, m = a.samples, a.snps
    table, g = O.sim_table_stream(a.seed, 0, m, n, 0, exact=True)
    cs, ct, ncase, nctrl = O.sim_phenotype_stream(a.seed, n)
    nb = g.data_blocks
    bits = lambda blocks: np.unpackbits(blocks.astype("<u2").view(np.uint8), bitorder="little").reshape(blocks.shape[0], -1)[:, :n]
    lo, hi = bits(table[:, 1:1 + nb]), bits(table[:, 1 + nb:1 + 2 * nb])
    codes = lo + 2 * hi  # 0 missing, 1, 2, 3
    case = bits(cs[None, :])[0].astype(bool)
    ctrl = bits(ct[None, :])[0].astype(bool)
    print(f"{m} SNPs x {n} samples, {case.sum()} cases, {ctrl.sum()} controls, missing {np.mean(codes == 0):.4f}")
    ind = [[(codes[:, grp] == k).astype(np.float32) for k in (1, 2, 3)] for grp in (case, ctrl)]
    # counts[g][a][b] = M x M
    cnt = [[[ind[gq][ka] @ ind[gq][kb].T for kb in range(3)] for ka in range(3)] for gq in range(2)]
    X = np.stack([np.stack(cnt[0][ka]) for ka in range(3)]).astype(np.float64)  # 3,3,M,M
    Y = np.stack([np.stack(cnt[1][ka]) for ka in range(3)]).astype(np.float64)
    S = X + Y
    nn = float(case.sum() + ctrl.sum())
    p0 = case.sum() / nn

    def xlgx(v):
        return np.where(v > 0, v * np.log2(np.maximum(v, 1.0)), 0.0)

    A = (xlgx(X) + xlgx(Y) - xlgx(S)).sum(axis=(0, 1))
    # the rest of the measure: (K_i + K_j + n lg tao), from the margins -- the kernel's identity
    # measure = 2 ln2 (G_ij - G_i - G_j) + 2 n ln tao  with  K_i = -(sum over i's classes of the same expression)
    mx, my = X.sum(axis=1), Y.sum(axis=1)            # 3,M,M : SNP i's class margins (same for every j)
    Ki = -(xlgx(mx) + xlgx(my) - xlgx(mx + my)).sum(axis=0)
    mxj, myj = X.sum(axis=0), Y.sum(axis=0)
    Kj = -(xlgx(mxj) + xlgx(myj) - xlgx(mxj + myj)).sum(axis=0)
    # tao (epistasis_func.cpp:424-470): sum_abc p(a,b) p(a,c) p(b,c) / (p(a) p(b) p(c)), c = group
    pab = S / nn
    pa, pb = pab.sum(axis=1), pab.sum(axis=0)       # 3,M,M
    tao = np.zeros_like(A)
    for grp, cnts, tot in ((0, X, case.sum()), (1, Y, ctrl.sum())):
        pc = tot / nn
        pac = cnts.sum(axis=1) / nn   # p(a, c)
        pbc = cnts.sum(axis=0) / nn
        for ka in range(3):
            for kb in range(3):
                with np.errstate(divide="ignore", invalid="ignore"):
                    term = pab[ka, kb] * pac[ka] * pbc[kb] / (pa[ka] * pb[kb] * pc)
                tao += np.nan_to_num(term)
    rest = Ki + Kj + nn * np.log2(tao) - nn * float(h2(np.array([case.sum() / nn]))[0])   # - n H(group)
    measure = 2.0 * math.log(2.0) * (A + rest)
    if m <= 1200:
        ref = O.boost_all_pairs(table, g, cs, ct, ncase, nctrl)
        mine = measure[np.triu_indices(m, 1)]
        fin = np.isfinite(mine) & np.isfinite(ref)
        print(f"identity vs oracle's measure on {fin.sum()} pairs: max |diff| {np.max(np.abs(mine[fin] - ref[fin])):.3e}")
    c, where, lim = quad_constant(p0)
    h0 = float(h2(np.array([p0]))[0])
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(S > 0, (X - p0 * S) ** 2 / np.maximum(S, 1.0), 0.0).sum(axis=(0, 1))
    A1 = -nn * h0 + c * q
    bound1 = 2.0 * math.log(2.0) * (A1 + rest)
    iu = np.triu_indices(m, 1)
    me, b1 = measure[iu], bound1[iu]
    ok = np.isfinite(me)
    print(f"p0 = {p0:.4f}  c = {c:.6f} (grid argmax {where}, limit {lim:.4f})  pairs {ok.sum()}")
    print(f"measure: mean {me[ok].mean():.3f}  max {me[ok].max():.2f}   >= thr: {np.mean(me[ok] >= a.threshold):.3e}")
    print(f"bound1 - measure: min {np.min((b1 - me)[ok]):.3e}  mean {np.mean((b1 - me)[ok]):.3f}  max {np.max((b1 - me)[ok]):.2f}")
    f = np.mean(b1[ok] >= a.threshold - 0.5)
    print(f"first-level bound >= thr - 0.5: {f:.3e} of the pairs")
    # round 4: the first level with  n lg tao <= n lg e (tao - 1)  (no logarithm at all): extra slack 2 n (tao - 1 - ln tao)
    t1 = tao[iu] - 1.0
    lin = 2.0 * nn * (t1 - np.log1p(t1))
    print(f"tao - 1: mean {t1[ok].mean():.2e}  std {t1[ok].std():.2e}  min {t1[ok].min():.3e}  max {t1[ok].max():.3e};  slack of the "
          f"linearised logarithm: mean {lin[ok].mean():.3f}  99.9 % {np.quantile(lin[ok], 0.999):.3f}  max {lin[ok].max():.2f}")
    f_lin = np.mean((b1 + lin)[ok] >= a.threshold - 0.5)
    print(f"first-level bound with the linearised logarithm >= thr - 0.5: {f_lin:.3e} of the pairs")
    # ... and with tao itself bounded from per-SNP numbers: tao - 1 = sum_c p(c) sum_ab p(ab) d_ac e_bc, d_ac = p(c|a)/p(c) - 1
    # for SNP i, e_bc for SNP j;  Cauchy-Schwarz over p(ab):  <= sum_c p(c) sqrt(sum_a p(a) d_ac^2) sqrt(sum_b p(b) e_bc^2)
    mxs, mys = X.sum(axis=1)[:, :, 0], Y.sum(axis=1)[:, :, 0]        # 3, M: per-SNP class counts of cases / controls
    tot = mxs + mys
    cs_bound = np.zeros((m, m))
    for cnts, pc in ((mxs, case.sum() / nn), (mys, ctrl.sum() / nn)):
        with np.errstate(divide="ignore", invalid="ignore"):
            d = np.where(tot > 0, cnts / np.maximum(tot, 1.0) / pc - 1.0, 0.0)       # 3, M
        r = np.sqrt(((tot / nn) * d * d).sum(axis=0))                               # M
        cs_bound += pc * np.outer(r, r)
    lin_cs = 2.0 * nn * cs_bound[iu]
    print(f"2 n x the Cauchy-Schwarz bound of tao - 1: mean {lin_cs[ok].mean():.3f}  99 % {np.quantile(lin_cs[ok], 0.99):.3f}  max {lin_cs[ok].max():.2f}"
          f"  (the true 2 n ln tao: mean {np.mean(2.0 * nn * np.log1p(t1[ok])):.4f})")
    b_cs = b1 - 2.0 * nn * np.log1p(t1) + lin_cs
    print(f"first-level bound with tao from per-SNP numbers >= thr - 0.5: {np.mean(b_cs[ok] >= a.threshold - 0.5):.3e} of the pairs; violations of the bound: {int(np.sum((me > b_cs + 1e-9)[ok]))}")
    # groups of 64 pairs as a 32 x 32 tile hands them to a wave: 64 consecutive (i, j) with j fastest, 32-wide tiles
    passing = (bound1 >= a.threshold - 0.5) & np.isfinite(measure)
    mm = (m // 32) * 32
    tiles = passing[:mm, :mm].reshape(mm // 32, 32, mm // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 16, 64)
    upper = np.triu(np.ones((mm // 32, mm // 32), bool), 1).reshape(-1)
    grp_any = tiles[upper].any(axis=2)
    print(f"64-pair groups with at least one pair going on: {grp_any.mean():.3e}   (expected for independent pairs: {1 - (1 - f) ** 64:.3e})")
    maf = np.minimum(pa[0, :, 0] * 0 + (2 * (codes == 1).sum(axis=1) + (codes == 2).sum(axis=1)) / (2.0 * n), 1.0)
    maf = np.minimum(maf, 1.0 - maf)
    low = maf < 0.05
    pl = passing[np.ix_(low, low)]
    ph = passing[np.ix_(~low, ~low)]
    print(f"MAF < 5 %: {low.mean():.3f} of the SNPs; passing among low x low {pl[np.triu_indices(pl.shape[0], 1)].mean():.3e}, high x high {ph[np.triu_indices(ph.shape[0], 1)].mean():.3e}")


if __name__ == "__main__":
    main()
