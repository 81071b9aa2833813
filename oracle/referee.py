"""referee.py -- an exact-arithmetic referee for esti_plane (include/common_lib.h:506-536 of the reference).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by bench.py's parity block, never by the product.

Why: the model half of the oracle restates Eigen's ColPivHouseholderQR (`A.colPivHouseholderQr().solve(b)`, common_lib.h:520); Eigen is
neither vendored nor installed here, so the restatement is pinned on the author's reading of Eigen ("parity unpinned").  This file
pins it on the MATHEMATICS instead, independent of that reading: whatever fp32 pivoted-QR implementation the reference links, it is
backward stable, i.e. its answer is the exact least-squares solution of a problem within a few rounding units of the given one.
The referee therefore
  1. solves the 5x3 problem  min || A x + 1 ||  (A = the five points, fp32 inputs taken as exact rationals) EXACTLY
     (fractions.Fraction, normal equations by Cramer's rule),
  2. measures the BACKWARD error of a candidate plane (n^, d) -- the smallest ||dA||_F / ||A||_F for which x^ = n^/d is an exact
     least-squares solution of (A + dA) x = -1 (Walden, Karlson & Sun 1995) -- in units of the fp32 unit roundoff u = 2^-24,
  3. bounds what such a perturbation can do to the residuals the two threshold tests read (|n^.p + d| > 0.1) and compares the
     candidate's inlier decisions with the exact ones wherever the exact margin exceeds that bound.
A restatement that is a correct backward-stable fp32 solve of THIS problem passes; a wrong reading of the algorithm (a sign, a pivot
rule, the normalisation, the meaning of d) does not."""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

U32 = 2.0 ** -24          # unit roundoff of fp32


def exact_solution(pts5) -> tuple:
    """x* = argmin ||A x + 1||_2 with A = pts5 (5 x 3 float32, taken as exact rationals).  Returns (x* as three Fractions or None when
    A^T A is singular, exact residuals (A x* + 1) as Fractions)."""
    A = [[Fraction(float(v)) for v in row] for row in np.asarray(pts5, np.float32).reshape(5, 3)]
    M = [[sum(A[k][i] * A[k][j] for k in range(5)) for j in range(3)] for i in range(3)]
    g = [-sum(A[k][i] for k in range(5)) for i in range(3)]           # A^T b, b = -1

    def det3(m):
        return (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))
    D = det3(M)
    if D == 0:
        return None, None
    x = []
    for c in range(3):
        Mc = [[g[r] if j == c else M[r][j] for j in range(3)] for r in range(3)]
        x.append(det3(Mc) / D)
    res = [sum(A[k][j] * x[j] for j in range(3)) + 1 for k in range(5)]
    return x, res


def backward_error(A64: np.ndarray, xhat: np.ndarray) -> float:
    """Walden-Karlson-Sun: the Frobenius norm of the smallest dA for which xhat is an exact least-squares solution of
    min ||(A + dA) x - b||, b = -1."""
    b = -np.ones(A64.shape[0])
    r = b - A64 @ xhat
    nr, nx = np.linalg.norm(r), np.linalg.norm(xhat)
    if nr == 0.0 or nx == 0.0:
        return 0.0
    eta1 = nr / nx
    P = np.eye(A64.shape[0]) - np.outer(r, r) / (nr * nr)
    smin = np.linalg.svd(np.hstack([A64, eta1 * P]), compute_uv=False)[-1]
    return float(min(eta1, smin))


def referee_case(pts5, plane4, ok: bool, threshold: float = 0.1, c_bound: float = 64.0) -> dict:
    """One neighbourhood: the exact solution against a candidate (n^x, n^y, n^z, d) + accept flag.  Returns the backward error in
    units of u * ||A||_F, the largest difference of the five signed distances to the exact ones against its bound, and whether the
    accept decision equals the exact one (None = inside the bound of the threshold: undecidable in fp32)."""
    A64 = np.asarray(pts5, np.float32).reshape(5, 3).astype(np.float64)
    x, res = exact_solution(pts5)
    out = {"singular": x is None}
    if x is None:
        return out
    n2 = float(x[0] * x[0] + x[1] * x[1] + x[2] * x[2])
    if n2 == 0.0:
        out["singular"] = True
        return out
    d_star = 1.0 / math.sqrt(n2)
    pd_star = np.array([float(r) for r in res]) * d_star                # n^*.p + d*, from the exact residuals
    p = np.asarray(plane4, np.float64)
    xhat = p[:3] / p[3]
    normA = np.linalg.norm(A64)
    be = backward_error(A64, xhat)
    sv = np.linalg.svd(A64, compute_uv=False)
    kappa = sv[0] / sv[-1] if sv[-1] > 0 else np.inf
    xs = np.array([float(v) for v in x])
    rs = np.array([float(r) for r in res])
    # a perturbation of relative size e = c u moves the residuals of a least-squares solution by <= e (||A|| ||x|| + kappa ||r||)
    # (Higham, Accuracy and Stability, Thm 20.1 without its second-order terms); times d*: metres
    bound = c_bound * U32 * (normA * np.linalg.norm(xs) + kappa * np.linalg.norm(rs)) * d_star
    pd_hat = A64 @ p[:3] + p[3]
    margin = np.abs(np.abs(pd_star) - threshold)
    exact_ok = bool((np.abs(pd_star) <= threshold).all())
    decidable = bool((margin > bound).all())
    out.update({"backward_error_over_u": be / (U32 * normA), "pd_max_diff": float(np.abs(pd_hat - pd_star).max()), "pd_bound": float(bound),
                "kappa": float(kappa), "exact_ok": exact_ok, "decidable": decidable,
                "decision_equal": (bool(ok) == exact_ok) if decidable else None, "min_margin": float(margin.min())})
    return out


def referee_batch_fp64(pts, planes, ok, threshold: float = 0.1, c_bound: float = 64.0) -> dict:
    """The same audit for MANY neighbourhoods (every plane of a 100 k-point scan) with the exact solve replaced by an fp64 Householder
    QR (numpy, batched): its own error (~1e-16 kappa) is nine orders below what is being judged; tests/test_oracle_math.py holds this
    route to the exact one on the 4 000 golden cases.  pts [N, 5, 3] float32, planes [N, 4], ok [N]."""
    A = np.asarray(pts, np.float32).astype(np.float64)
    N = len(A)
    planes = np.asarray(planes, np.float64)
    ok = np.asarray(ok).astype(bool)
    Q, R = np.linalg.qr(A)                                              # batched, reduced: Q [N, 5, 3], R [N, 3, 3]
    sv = np.linalg.svd(A, compute_uv=False)
    good = sv[:, -1] > 1e-9 * sv[:, 0]
    rhs = -(Q.transpose(0, 2, 1) @ np.ones((N, 5, 1)))
    xs = np.zeros((N, 3))
    xs[good] = np.linalg.solve(R[good], rhs[good])[:, :, 0]
    good &= np.linalg.norm(xs, axis=1) > 0
    d_star = np.where(good, 1.0 / np.maximum(np.linalg.norm(xs, axis=1), 1e-300), 0.0)
    rs = (A @ xs[:, :, None])[:, :, 0] + 1.0
    pd_star = rs * d_star[:, None]
    normA = np.linalg.norm(A, axis=(1, 2))
    kappa = np.where(sv[:, -1] > 0, sv[:, 0] / np.maximum(sv[:, -1], 1e-300), np.inf)
    bound = c_bound * U32 * (normA * np.linalg.norm(xs, axis=1) + kappa * np.linalg.norm(rs, axis=1)) * d_star
    pd_hat = (A @ planes[:, :3, None])[:, :, 0] + planes[:, 3:4]
    margin = np.abs(np.abs(pd_star) - threshold)
    exact_ok = (np.abs(pd_star) <= threshold).all(axis=1)
    decidable = good & (margin > bound[:, None]).all(axis=1)
    # backward error (WKS) per case: small SVDs in a loop over the candidate planes
    be = np.zeros(N)
    xhat = planes[:, :3] / planes[:, 3:4]
    for i in np.nonzero(good)[0]:
        be[i] = backward_error(A[i], xhat[i]) / (U32 * normA[i])
    sel = good
    return {"planes": int(N), "audited": int(sel.sum()), "max_backward_error_over_u": float(be[sel].max()) if sel.any() else 0.0,
            "median_backward_error_over_u": float(np.median(be[sel])) if sel.any() else 0.0,
            "pd_diff_over_bound_max": float((np.abs(pd_hat - pd_star)[sel].max(axis=1) / np.maximum(bound[sel], 1e-300)).max()) if sel.any() else 0.0,
            "max_abs_pd_diff_m": float(np.abs(pd_hat - pd_star)[sel].max()) if sel.any() else 0.0,
            "decidable": int(decidable.sum()), "decisions_equal": bool((ok[decidable] == exact_ok[decidable]).all()),
            "decisions_differing": int((ok[decidable] != exact_ok[decidable]).sum()),
            "undecidable_within_bound_of_threshold": int((sel & ~decidable).sum()), "c_bound": c_bound,
            "how": "exact solve replaced by a batched fp64 Householder QR (oracle/referee.py); backward error after Walden-Karlson-Sun in "
                   "units of u32 * ||A||_F; a decision counts where every point's exact margin to the 0.1 m threshold exceeds the "
                   "perturbation bound c u (||A|| ||x|| + kappa ||r||) d"}
