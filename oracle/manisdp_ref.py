"""ORACLE (test infrastructure, not product code) -- CPU restatement of the three
ManiSDP primal entry points, following the reference ``.m`` files line by line.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path never does.

Follows (paths relative to the reference tree):
  * src/primal/ManiSDP_onlyunitdiag.m:6-156
  * src/primal/ManiSDP_unitdiag.m:7-198
  * src/primal/ManiSDP_unittrace.m:7-177
  * manopt7.0/manopt/manifolds/sphere/spherefactory.m:83-153,220-232,249-254
  * src/primal/ManiSDP_multiblock.m, src/dual/ManiDSDP_unitdiag.m:8-220 (the "next" rows of SURVEY.md 8f-4)
with the RTR/tCG restatement in ``oracle/manopt_rtr.py``.

Parity status: the MATLAB reference cannot be executed in this environment; the
oracle is pinned by the known optimal values the reference ships
(data/sdplib/README:39-51,71-88,98-105) -- see tests/test_oracle_known_answers.py.

Array layout: all factors are NumPy ``(n, p)`` arrays.  For the two oblique entry
points this is the transpose of MATLAB's ``p x n`` (same bytes as MATLAB's
column-major storage); for ``unittrace`` it is MATLAB's own ``n x p``.
"""
from __future__ import annotations

import math
import time

import numpy as np
import scipy.sparse as sp

from .manopt_rtr import trustregions


# ----------------------------------------------------------------------- manifolds
class ObliqueNT:
    """``obliquefactoryNTrans(p, n)`` (ManiSDP_onlyunitdiag.m:132-156,
    ManiSDP_unitdiag.m:173-198) on (n, p) arrays with unit rows."""

    def __init__(self, p, n, inner_all=True):
        self.p, self.n = p, n
        self.inner_all = inner_all            # sum(.,'all') vs d1(:)'*d2(:)

    def dim(self):
        return (self.p - 1) * self.n

    def inner(self, x, d1, d2):
        if self.inner_all:
            return float(np.sum(d1 * d2))     # ManiSDP_onlyunitdiag.m:134
        return float(d1.ravel() @ d2.ravel())  # ManiSDP_unitdiag.m:176

    def norm(self, x, d):
        return float(np.linalg.norm(d))       # :136 norm(d,'fro')

    def typicaldist(self):
        return math.pi * math.sqrt(self.n)    # :137

    def proj(self, X, U):
        return U - X * np.sum(X * U, axis=1, keepdims=True)   # :138

    tangent = proj                            # :139

    def retr(self, x, d):
        xtd = x + d                           # :143-144
        return xtd / np.sqrt(np.sum(xtd ** 2, axis=1, keepdims=True))

    def zerovec(self, x):
        return np.zeros((self.n, self.p))     # :149

    def rand(self, rng):
        x = rng.standard_normal((self.n, self.p))               # :153-154
        return x / np.sqrt(np.sum(x ** 2, axis=1, keepdims=True))


class SphereF:
    """``spherefactory(n, p)``: unit Frobenius norm n x p matrices
    (spherefactory.m:85,87,111,113,151,220-232,249-254)."""

    def __init__(self, n, p):
        self.n, self.p = n, p

    def dim(self):
        return self.n * self.p - 1

    def inner(self, x, d1, d2):
        return float(d1.ravel() @ d2.ravel())

    def norm(self, x, d):
        return float(np.linalg.norm(d))

    def typicaldist(self):
        return math.pi

    def proj(self, x, d):
        return d - x * float(x.ravel() @ d.ravel())

    tangent = proj

    def retr(self, x, d):
        y = x + d
        return y / np.linalg.norm(y)

    def zerovec(self, x):
        return np.zeros((self.n, self.p))

    def rand(self, rng):
        x = rng.standard_normal((self.n, self.p))
        return x / np.linalg.norm(x)


# ------------------------------------------------------------------------ utilities
def _opt(options, name, default):
    if options is None:
        return default
    return options.get(name, default)


def _say(verbose, msg):
    if verbose:
        print(msg)


def dense_eig(S):
    """``eig(full(S),'vector')``: all eigenpairs, ascending."""
    if sp.issparse(S):
        S = S.toarray()
    dS, vS = np.linalg.eigh(S)
    return dS, vS


def _thin_svd_rank(Y, theta):
    """``svd(Y)`` + ``r = sum(e >= theta*e(1))`` on an (n,p) factor.  Returns
    (V_thin, e, r): the first r columns of V_thin scaled by e reproduce the
    reference's rank cut ``V(:,1:r)'.*e(1:r)`` (ManiSDP_onlyunitdiag.m:52-54,70-72)."""
    V, e, _ = np.linalg.svd(Y, full_matrices=False)
    r = int(np.sum(e >= theta * e[0]))
    return V, e, r


# ------------------------------------------------------------------ onlyunitdiag
class _OnlyUnitDiagProblem:
    """cost/grad/hess closures of ManiSDP_onlyunitdiag.m:117-130 with the shared
    variables ``YC, eG`` (quirk Q1: ``q1='reference'`` keeps the reference's stale
    state after a rejected step, ``q1='correct'`` restores the per-point state)."""

    def __init__(self, C, n, p, q1="reference"):
        self.C = C
        self.M = ObliqueNT(p, n, inner_all=True)
        self.YC = None
        self.eG = None
        self._bak = None
        self.q1 = q1
        self.nhess = 0

    def cost(self, Y):
        self._bak = (self.YC, self.eG)
        self.YC = self.C @ Y                              # :118  YC = Y*C  (C symmetric)
        self.eG = np.sum(self.YC * Y, axis=1, keepdims=True)   # :119
        return 0.5 * float(np.sum(self.eG))               # :120

    def grad(self, Y):
        return self.YC - Y * self.eG                      # :124

    def hess(self, Y, U):
        self.nhess += 1
        eH = self.C @ U                                   # :128
        return eH - Y * np.sum(Y * eH, axis=1, keepdims=True) - U * self.eG   # :129

    def on_reject(self):
        if self.q1 == "correct":
            self.YC, self.eG = self._bak


def hessvec_onlyunitdiag(C, Y, U):
    """One Hess-vec of ManiSDP_onlyunitdiag.m:127-130 at point Y (eG taken at Y)."""
    eG = np.sum((C @ Y) * Y, axis=1, keepdims=True)
    eH = C @ U
    return eH - Y * np.sum(Y * eH, axis=1, keepdims=True) - U * eG


def onlyunitdiag_rows(Crows, rows, Y, U=None):
    """The closures of ManiSDP_onlyunitdiag.m:117-130 on a SUBSET of rows -- for checks at sizes where the whole dense C (3.2 GB at
    n = 20000) is not wanted on the host: ``Crows = C[rows, :]``.  Returns (eG[rows], grad[rows], hess[rows] or None); the cost is
    ``0.5 * sum(eG)`` over ALL rows (:120), i.e. the sum of the first output over a partition of the rows."""
    Yr = Y[rows]
    YC = Crows @ Y                                        # :118 (C symmetric)
    eG = np.sum(YC * Yr, axis=1, keepdims=True)           # :119
    G = YC - Yr * eG                                      # :124
    H = None
    if U is not None:
        eH = Crows @ U                                    # :128
        H = eH - Yr * np.sum(Yr * eH, axis=1, keepdims=True) - U[rows] * eG   # :129
    return eG[:, 0], G, H


def ManiSDP_onlyunitdiag(C, options=None, rng=None, verbose=False, eig_fn=None, q1="reference"):
    """``[X, obj, data] = ManiSDP_onlyunitdiag(C, options)``; returns (Y, obj, data)
    with ``X = Y @ Y.T`` (``data['X']`` is only formed when n <= 4000).

    ``options['Y0']`` (an (n, p0) array) replaces the reference's ``randn`` start so
    that tests can drive the oracle and the HIP path from identical points."""
    o = dict(options or {})
    p0 = o.get("p0", 2); AL_maxiter = o.get("AL_maxiter", 20); tol = o.get("tol", 1e-8)
    theta = o.get("theta", 1e-1); delta = o.get("delta", 8); alpha = o.get("alpha", 0.5)
    tolgradnorm = o.get("tolgradnorm", 1e-8); TR_maxinner = o.get("TR_maxinner", 100)
    TR_maxiter = o.get("TR_maxiter", 40); line_search = o.get("line_search", 0)
    eig_fn = eig_fn or dense_eig
    rng = rng or np.random.default_rng(0)
    _say(verbose, "ManiSDP is starting...")
    n = C.shape[0]
    _say(verbose, f"SDP size: n = {n}, m = {n}")
    Csp = C.tocsr() if sp.issparse(C) else np.asarray(C)
    p = p0
    Y = o.get("Y0", None)
    U = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0,
            "eig_seconds": 0.0}
    t0 = time.time()
    dinf0 = None
    prob = _OnlyUnitDiagProblem(Csp, n, p, q1=q1)

    def co(Yv):                                            # :99-101
        return float(np.sum((Csp @ Yv) * Yv))

    def do_line_search(Yv, Uv):                            # :103-115
        a = 1.0
        cost0 = co(Yv)
        i = 1
        nY = Yv + a * Uv
        nY = nY / np.sqrt(np.sum(nY ** 2, axis=1, keepdims=True))
        while i <= 15 and co(nY) - cost0 > -1e-3:
            a = 0.8 * a
            nY = Yv + a * Uv
            nY = nY / np.sqrt(np.sum(nY ** 2, axis=1, keepdims=True))
            i += 1
        return nY

    obj = dinf = gradnorm = None
    z = S = None
    for it in range(1, AL_maxiter + 1):                    # :38
        prob.M = ObliqueNT(p, n, inner_all=True)           # :39
        if U is not None:
            Y = do_line_search(Y, U)                       # :40-42
        t1 = time.time()
        Y, _, info = trustregions(prob, Y, TR_maxiter, TR_maxinner, tolgradnorm, rng=rng)  # :43
        data["rtr_seconds"] += time.time() - t1
        data["hessvecs"] += info.hessvecs
        data["cost_evals"] += info.cost_evals
        data["rejected"] += info.rejected
        gradnorm = info.gradnorm                           # :44
        Y_eval = Y                                         # :45 X = Y'*Y is what the reference returns (:86)
        z = np.sum((Csp @ Y) * Y, axis=1)                  # :46-47 (z = sum(C.*X) = sum((Y*C).*Y))
        obj = float(np.sum(z))                             # :48
        S = Csp - sp.diags(z) if sp.issparse(Csp) else Csp - np.diag(z)   # :49
        t1 = time.time()
        dS, vS = eig_fn(S)                                 # :50
        data["eig_seconds"] += time.time() - t1
        dinf = max(0.0, -dS[0]) / (1.0 + dS[-1])           # :51
        V, e, r = _thin_svd_rank(Y, theta)                 # :52-54
        _say(verbose, "Iter %d, obj:%0.8f, dinf:%0.1e, r:%d, p:%d, time:%0.2fs"
             % (it, obj, dinf, r, p, time.time() - t0))
        data["iters"] = it
        if dinf < tol:                                     # :57-60
            _say(verbose, "Optimality is reached!")
            break
        if it % 20 == 0:                                   # :61-69
            if it > 50 and dinf > dinf0:
                data["status"] = 2
                _say(verbose, "Slow progress!")
                break
            else:
                dinf0 = dinf
        if r <= p - 1:                                     # :70-73
            Y = V[:, :r] * e[:r]
            p = r
        nne = max(min(int(np.sum(dS < 0)), delta), 1)      # :74
        if line_search == 1:                               # :75-77
            U = np.hstack([np.zeros((n, p)), vS[:, :nne]])
        p = p + nne                                        # :78
        if line_search == 1:
            Y = np.hstack([Y, np.zeros((n, nne))])         # :80
        else:
            Y = np.hstack([Y, alpha * vS[:, :nne]])        # :82
            Y = Y / np.sqrt(np.sum(Y ** 2, axis=1, keepdims=True))   # :83
    Y = Y_eval                                             # the loop's last pass has already widened its own copy
    data.update({"Y": Y, "S": S, "z": z, "dinf": dinf, "gradnorm": gradnorm,
                 "time": time.time() - t0, "p": Y.shape[1]})
    if n <= 4000:
        data["X"] = Y @ Y.T
    if data["status"] == 0 and dinf > tol:                 # :92-95
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y, obj, data


# ---------------------------------------------------------------------- unitdiag
def _as_dense_vec(v):
    if sp.issparse(v):
        return np.asarray(v.todense()).ravel()
    return np.asarray(v, dtype=np.float64).ravel()


class _UnitDiagProblem:
    """cost/grad/hess closures of ManiSDP_unitdiag.m:152-171.  Shared variables
    (parent workspace): ``Axb, eS``; ``YeG`` lives in the per-point store, which is
    equivalent to "value at the last accepted point" because ``grad`` is only ever
    called there."""

    def __init__(self, At, b, c, n, p):
        self.At = At.tocsc()
        self.A = self.At.T.tocsr()
        self.b = b
        self.c = c
        self.n = n
        self.M = ObliqueNT(p, n, inner_all=False)
        self.y = np.zeros(b.size)
        self.sigma = 1.0
        self.Axb = None
        self.eS = None
        self.YeG = None
        self.nhess = 0

    def _x(self, Y):
        X = Y @ Y.T                                        # :153  X = Y'*Y
        return X.ravel(order="F")

    def cost(self, Y):
        x = self._x(Y)
        self.Axb = self.A @ x - self.b - self.y / self.sigma      # :155
        return float(self.c @ x) + 0.5 * self.sigma * float(self.Axb @ self.Axb)   # :156

    def grad(self, Y):
        n = self.n
        self.eS = (self.c + self.sigma * (self.At @ self.Axb)).reshape((n, n), order="F")  # :160
        eG = 2.0 * (self.eS.T @ Y)                         # :161  eG = 2*Y*eS
        self.YeG = np.sum(Y * eG, axis=1, keepdims=True)   # :162
        return eG - Y * self.YeG                           # :163

    def hess(self, Y, U):
        self.nhess += 1
        n = self.n
        YU = Y @ U.T                                       # :167  YU = Y'*U
        AyU = (self.At @ (self.A @ YU.ravel(order="F"))).reshape((n, n), order="F")   # :168
        eH = 2.0 * (self.eS.T @ U) + 4.0 * self.sigma * (AyU.T @ Y)                    # :169
        return eH - Y * np.sum(Y * eH, axis=1, keepdims=True) - U * self.YeG           # :170


def ManiSDP_unitdiag(At, b, c, K, options=None, rng=None, verbose=False):
    """``[X, obj, data] = ManiSDP_unitdiag(At, b, c, K, options)``; returns
    (Y, obj, data).  ``options['Y0']`` optionally fixes the start point."""
    o = dict(options or {})
    n = int(K["s"])
    p0 = o.get("p0", 2); AL_maxiter = o.get("AL_maxiter", 300); gama = o.get("gama", 2)
    sigma0 = o.get("sigma0", 1e-3); sigma_min = o.get("sigma_min", 1e-2); sigma_max = o.get("sigma_max", 1e7)
    tol = o.get("tol", 1e-8); theta = o.get("theta", 1e-3); delta = o.get("delta", 8)
    alpha = o.get("alpha", 0.1); tolgradnorm = o.get("tolgradnorm", 1e-8)
    TR_maxinner = o.get("TR_maxinner", 20); TR_maxiter = o.get("TR_maxiter", 4)
    tau1 = o.get("tau1", 1); tau2 = o.get("tau2", 1); line_search = o.get("line_search", 0)
    rng = rng or np.random.default_rng(0)
    b = _as_dense_vec(b)
    c = _as_dense_vec(c)
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {n}, m = {b.size}")
    prob = _UnitDiagProblem(At, b, c, n, p0)
    A, Atc = prob.A, prob.At
    p = p0
    sigma = sigma0
    y = np.zeros(b.size)
    normb = 1.0 + np.linalg.norm(b)
    Y = o.get("Y0", None)
    U = None
    fac_size = []
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None

    def rownorm(Z):
        return Z / np.sqrt(np.sum(Z ** 2, axis=1, keepdims=True))

    def co(Yv):                                            # :131-136
        x = (Yv @ Yv.T).ravel(order="F")
        Axb = A @ x - b - y / sigma
        return float(c @ x) + sigma / 2.0 * float(Axb @ Axb)

    def do_line_search(Yv, Uv):                            # :138-150
        a = 1.0
        cost0 = co(Yv)
        i = 1
        nY = rownorm(Yv + a * Uv)
        while i <= 15 and co(nY) - cost0 > -1e-3:
            a = 0.8 * a
            nY = rownorm(Yv + a * Uv)
            i += 1
        return nY

    obj = gap = pinf = dinf = gradnorm = eta_kkt = None
    S = z = None
    for it in range(1, AL_maxiter + 1):                    # :51
        fac_size.append(p)
        prob.M = ObliqueNT(p, n, inner_all=False)          # :53
        prob.y, prob.sigma = y, sigma
        if U is not None:
            Y = do_line_search(Y, U)                       # :54-56
        t1 = time.time()
        Y, _, info = trustregions(prob, Y, TR_maxiter, TR_maxinner, tolgradnorm, rng=rng)  # :57
        data["rtr_seconds"] += time.time() - t1
        data["hessvecs"] += info.hessvecs
        data["cost_evals"] += info.cost_evals
        data["rejected"] += info.rejected
        gradnorm = info.gradnorm
        X = Y @ Y.T                                        # :59
        x = X.ravel(order="F")
        obj = float(c @ x)                                 # :61
        Axb = A @ x - b                                    # :62
        pinf = float(np.linalg.norm(Axb)) / normb          # :63
        y = y - sigma * Axb                                # :64
        eS = (c - Atc @ y).reshape((n, n), order="F")      # :65
        z = np.sum(X * eS, axis=0)                         # :66
        S = eS - np.diag(z)                                # :67
        dS, vS = np.linalg.eigh(S)                         # :68
        dinf = max(0.0, -dS[0]) / (1.0 + dS[-1])           # :69
        by = float(b @ y) + float(np.sum(z))               # :70
        gap = abs(obj - by) / (abs(by) + abs(obj) + 1.0)   # :71
        V, e, r = _thin_svd_rank(Y, theta)                 # :72-74
        _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs"
             % (it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
        eta_kkt = max(gap, pinf, dinf)                     # :77
        data["iters"] = it
        if eta_kkt < tol:
            _say(verbose, "Optimality is reached!")
            break
        if it % 50 == 0:                                   # :82-92
            if it > 100 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                data["status"] = 2
                _say(verbose, "Slow progress!")
                break
            else:
                gap0, pinf0, dinf0 = gap, pinf, dinf
        if r <= p - 1:                                     # :93-96
            Y = V[:, :r] * e[:r]
            p = r
        nne = max(min(int(np.sum(dS < 0)), delta), 1)      # :97
        if line_search == 1:
            U = np.hstack([np.zeros((n, p)), vS[:, :nne]])  # :99
        p = p + nne
        if line_search == 1:
            Y = np.hstack([Y, np.zeros((n, nne))])         # :103
        else:
            Y = rownorm(np.hstack([Y, alpha * vS[:, :nne]]))   # :105-106
        if pinf < tau1 * gradnorm:                         # :108-112
            sigma = max(sigma / gama, sigma_min)
        elif pinf > tau2 * gradnorm:
            sigma = min(sigma * gama, sigma_max)
    data.update({"Y": Y, "y": y, "S": S, "z": z, "gap": gap, "pinf": pinf, "dinf": dinf,
                 "gradnorm": gradnorm, "time": time.time() - t0, "fac_size": fac_size,
                 "X": X, "sigma": sigma})     # X of the evaluated point (:59), as the reference returns it
    if data["status"] == 0 and eta_kkt > tol:
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y, obj, data


# --------------------------------------------------------------------- unittrace
class _UnitTraceProblem:
    """cost/grad/hess closures of ManiSDP_unittrace.m:156-177; everything lives in
    the per-point ``store`` (so a rejected proposal leaves the current state intact)."""

    def __init__(self, At, b, c, n, p):
        self.At = At.tocsc()
        self.A = self.At.T.tocsr()
        self.b = b
        self.c = c
        self.n = n
        self.M = SphereF(n, p)
        self.y = np.zeros(b.size)
        self.sigma = 1.0
        self.cur = None
        self.prop = None
        self.nhess = 0

    def cost(self, Y):
        n = self.n
        X = Y @ Y.T                                        # :157
        x = X.ravel(order="F")
        Axb = self.A @ x - self.b - self.y / self.sigma    # :159
        f = float(self.c @ x) + self.sigma / 2.0 * float(Axb @ Axb)   # :160
        eS = (self.c + self.sigma * (self.At @ Axb)).reshape((n, n), order="F")   # :161
        zz = float(np.sum(X * eS))                         # :162
        G = 2.0 * (eS @ Y) - 2.0 * zz * Y                  # :163
        self.prop = {"z": zz, "G": G, "eS": eS}
        if self.cur is None:
            self.cur = self.prop
        return f

    def on_accept(self):
        self.cur = self.prop

    def grad(self, Y):
        return self.cur["G"]                               # :168

    def hess(self, Y, U):
        self.nhess += 1
        n = self.n
        YU = U @ Y.T                                       # :172
        AyU = (self.At @ (self.A @ YU.ravel(order="F"))).reshape((n, n), order="F")   # :173
        H = 2.0 * (self.cur["eS"] @ U) + 4.0 * self.sigma * (AyU @ Y)                  # :174
        return H - float(np.sum(H * Y)) * Y - 2.0 * self.cur["z"] * U                  # :176


def ManiSDP_unittrace(At, b, c, K, options=None, rng=None, verbose=False):
    """``[X, obj, data] = ManiSDP_unittrace(At, b, c, K, options)``; returns (Y, obj, data)."""
    o = dict(options or {})
    n = int(K["s"])
    p0 = o.get("p0", 1); AL_maxiter = o.get("AL_maxiter", 1000); gama = o.get("gama", 2)
    sigma0 = o.get("sigma0", 1e1); sigma_min = o.get("sigma_min", 1e2); sigma_max = o.get("sigma_max", 1e7)
    tol = o.get("tol", 1e-8); theta = o.get("theta", 1e-2); delta = o.get("delta", 8)
    alpha = o.get("alpha", 0.05); tolgradnorm = o.get("tolgradnorm", 1e-8)
    TR_maxinner = o.get("TR_maxinner", 40); TR_maxiter = o.get("TR_maxiter", 3)
    tau1 = o.get("tau1", 1e-5); tau2 = o.get("tau2", 1e-4); line_search = o.get("line_search", 1)
    rng = rng or np.random.default_rng(0)
    b = _as_dense_vec(b)
    c = _as_dense_vec(c)
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {n}, m = {b.size}")
    prob = _UnitTraceProblem(At, b, c, n, p0)
    A, Atc = prob.A, prob.At
    p = p0
    sigma = sigma0
    y = np.zeros(b.size)
    normb = 1.0 + np.linalg.norm(b)
    Y = o.get("Y0", None)                                  # :36-40
    U = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None

    def co(Yv):                                            # :135-140
        x = (Yv @ Yv.T).ravel(order="F")
        Axb = A @ x - b - y / sigma
        return float(c @ x) + sigma / 2.0 * float(Axb @ Axb)

    def do_line_search(Yv, Uv):                            # :142-154
        a = 1.0
        cost0 = co(Yv)
        i = 1
        nY = Yv + a * Uv
        nY = nY / np.linalg.norm(nY)
        while i <= 15 and co(nY) - cost0 > -1e-3:
            a = 0.8 * a
            nY = Yv + a * Uv
            nY = nY / np.linalg.norm(nY)
            i += 1
        return nY

    obj = gap = pinf = dinf = gradnorm = eta_kkt = None
    S = z = None
    for it in range(1, AL_maxiter + 1):                    # :52
        prob.M = SphereF(n, p)                             # :53
        prob.y, prob.sigma = y, sigma
        prob.cur = prob.prop = None
        if U is not None:
            Y = do_line_search(Y, U)                       # :54-56
        t1 = time.time()
        Y, _, info = trustregions(prob, Y, TR_maxiter, TR_maxinner, tolgradnorm, rng=rng)  # :57
        data["rtr_seconds"] += time.time() - t1
        data["hessvecs"] += info.hessvecs
        data["cost_evals"] += info.cost_evals
        data["rejected"] += info.rejected
        gradnorm = info.gradnorm
        X = Y @ Y.T                                        # :59
        x = X.ravel(order="F")
        obj = float(c @ x)                                 # :61
        Axb = A @ x - b
        pinf = float(np.linalg.norm(Axb)) / normb          # :63
        y = y - sigma * Axb                                # :64
        eS = (c - Atc @ y).reshape((n, n), order="F")      # :65
        z = float(np.sum(eS * X))                          # :66
        S = eS - z * np.eye(n)                             # :67
        dS, vS = np.linalg.eigh(S)                         # :68
        dinf = max(0.0, -dS[0]) / (1.0 + dS[-1])           # :69
        by = float(b @ y) + z                              # :70
        gap = abs(obj - by) / (abs(by) + abs(obj) + 1.0)   # :71
        V, e, r = _thin_svd_rank(Y, theta)                 # :72-78
        _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs"
             % (it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
        eta_kkt = max(pinf, gap, dinf)                     # :81
        data["iters"] = it
        if eta_kkt < tol:
            _say(verbose, "Optimality is reached!")
            break
        if it % 20 == 0:                                   # :86-96
            if it > 50 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                data["status"] = 2
                _say(verbose, "Slow progress!")
                break
            else:
                gap0, pinf0, dinf0 = gap, pinf, dinf
        if r <= p - 1:                                     # :97-100
            Y = V[:, :r] * e[:r]
            p = r
        nne = min(int(np.sum(dS < 0)), delta)              # :101 (no max(.,1) here)
        if line_search == 1:
            U = np.hstack([np.zeros((n, p)), vS[:, :nne]])  # :103
        p = p + nne
        if line_search == 1:
            Y = np.hstack([Y, np.zeros((n, nne))])         # :107
        else:
            Y = np.hstack([Y, alpha * vS[:, :nne]])        # :109-110
            Y = Y / np.linalg.norm(Y)
        if pinf < tau1 * gradnorm:                         # :113-117
            sigma = max(sigma / gama, sigma_min)
        elif pinf > tau2 * gradnorm:
            sigma = min(sigma * gama, sigma_max)
    data.update({"Y": Y, "y": y, "S": S, "z": z, "gap": gap, "pinf": pinf, "dinf": dinf,
                 "gradnorm": gradnorm, "time": time.time() - t0, "X": X, "sigma": sigma})     # X of the evaluated point (:59), as the reference returns it
    if data["status"] == 0 and eta_kkt > tol:
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y, obj, data


# --------------------------------------------------------------------------- generic ManiSDP.m (Euclidean manifold)
class EuclidNP:
    """``euclideanfactory(n, p)`` (manopt/manifolds/euclidean/euclideanfactory.m:49-82): flat n x p matrices."""

    def __init__(self, n, p):
        self.n, self.p = n, p

    def dim(self):
        return self.n * self.p                             # :49

    def inner(self, x, d1, d2):
        return float(d1.ravel() @ d2.ravel())              # :51

    def norm(self, x, d):
        return float(np.linalg.norm(d))                    # :53

    def typicaldist(self):
        return math.sqrt(self.n * self.p)                  # :57

    def proj(self, x, d):
        return d                                           # :59

    tangent = proj                                         # :65

    def retr(self, x, d):
        return x + d                                       # :67-76 (retr = exp)

    def zerovec(self, x):
        return np.zeros((self.n, self.p))

    def rand(self, rng):
        return rng.standard_normal((self.n, self.p))       # :82


class _GenericProblem:
    """cost/grad/hess closures of src/primal/ManiSDP.m:149-164.  ``S`` is set by ``grad`` (Manopt evaluates the
    gradient at accepted points only), ``Axb`` by ``cost``; kept per point here."""

    def __init__(self, At, b, c, n, p):
        self.At = At.tocsc()
        self.A = self.At.T.tocsr()
        self.b = b
        self.c = c
        self.n = n
        self.M = EuclidNP(n, p)
        self.y = np.zeros(b.size)
        self.sigma = 1.0
        self.cur = None
        self.prop = None
        self.nhess = 0

    def cost(self, Y):
        n = self.n
        x = (Y @ Y.T).ravel(order="F")                     # :150-151
        Axb = self.A @ x - self.b - self.y / self.sigma    # :152
        f = float(self.c @ x) + self.sigma / 2.0 * float(Axb @ Axb)   # :153
        S = (self.c + self.sigma * (self.At @ Axb)).reshape((n, n), order="F")   # :157
        self.prop = {"S": S, "G": 2.0 * (S @ Y)}           # :158
        if self.cur is None:
            self.cur = self.prop
        return f

    def on_accept(self):
        self.cur = self.prop

    def grad(self, Y):
        return self.cur["G"]

    def hess(self, Y, U):
        self.nhess += 1
        n = self.n
        YU = U @ Y.T                                       # :162
        AyU = (self.At @ (self.A @ YU.ravel(order="F"))).reshape((n, n), order="F")   # :163
        return 2.0 * (self.cur["S"] @ U) + 4.0 * self.sigma * (AyU @ Y)                # :164


def ManiSDP(At, b, c, K, options=None, rng=None, verbose=False):
    """``[X, obj, data] = ManiSDP(At, b, c, K, options)`` (src/primal/ManiSDP.m:6; defaults :9-25); returns
    (Y, obj, data)."""
    o = dict(options or {})
    n = int(K["s"])
    p0 = o.get("p0", 1); AL_maxiter = o.get("AL_maxiter", 1000); gama = o.get("gama", 2)
    sigma0 = o.get("sigma0", 1e-2); sigma_min = o.get("sigma_min", 1e-1); sigma_max = o.get("sigma_max", 1e7)
    tol = o.get("tol", 1e-8); theta = o.get("theta", 1e-2); delta = o.get("delta", 8)
    alpha = o.get("alpha", 0.1); tolgradnorm = o.get("tolgradnorm", 1e-8)
    TR_maxinner = o.get("TR_maxinner", 20); TR_maxiter = o.get("TR_maxiter", 4)
    tau1 = o.get("tau1", 1e-2); tau2 = o.get("tau2", 1e-1); line_search = o.get("line_search", 1)
    rng = rng or np.random.default_rng(0)
    b = _as_dense_vec(b)
    c = _as_dense_vec(c)
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {n}, m = {b.size}")
    prob = _GenericProblem(At, b, c, n, p0)
    A, Atc = prob.A, prob.At
    p = p0
    sigma = sigma0
    y = np.zeros(b.size)
    normb = 1.0 + np.linalg.norm(b)
    Y = o.get("Y0", None)                                  # :35-39
    U = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None

    def co(Yv):                                            # :142-147
        x = (Yv @ Yv.T).ravel(order="F")
        Axb = A @ x - b - y / sigma
        return float(c @ x) + sigma / 2.0 * float(Axb @ Axb)

    def do_line_search(Yv, Uv):                            # :130-140
        a = 1.0
        cost0 = co(Yv)
        i = 1
        nY = Yv + a * Uv
        while i <= 15 and co(nY) - cost0 > -1e-3:
            a = 0.8 * a
            nY = Yv + a * Uv
            i += 1
        return nY

    obj = gap = pinf = dinf = gradnorm = eta_kkt = None
    S = None
    for it in range(1, AL_maxiter + 1):                    # :51
        prob.M = EuclidNP(n, p)                            # :52
        prob.y, prob.sigma = y, sigma
        prob.cur = prob.prop = None
        if U is not None:
            Y = do_line_search(Y, U)                       # :53-55
        t1 = time.time()
        Y, _, info = trustregions(prob, Y, TR_maxiter, TR_maxinner, tolgradnorm, rng=rng)  # :56
        data["rtr_seconds"] += time.time() - t1
        data["hessvecs"] += info.hessvecs
        data["cost_evals"] += info.cost_evals
        data["rejected"] += info.rejected
        gradnorm = info.gradnorm
        X = Y @ Y.T                                        # :58
        x = X.ravel(order="F")
        Axb = A @ x - b                                    # :60
        pinf = float(np.linalg.norm(Axb)) / normb          # :61
        y = y - sigma * Axb                                # :62
        obj = float(c @ x)                                 # :63
        S = (c - Atc @ y).reshape((n, n), order="F")       # :64
        dS, vS = np.linalg.eigh(S)                         # :65
        dinf = max(0.0, -dS[0]) / (1.0 + dS[-1])           # :66
        by = float(b @ y)                                  # :67
        gap = abs(obj - by) / (abs(by) + abs(obj) + 1.0)   # :68
        V, e, r = _thin_svd_rank(Y, theta)                 # :69-75
        _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs"
             % (it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
        eta_kkt = max(pinf, gap, dinf)                     # :79
        data["iters"] = it
        if eta_kkt < tol:
            _say(verbose, "Optimality is reached!")
            break
        if it % 20 == 0:                                   # :84-94
            if it > 50 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                data["status"] = 2
                _say(verbose, "Slow progress!")
                break
            else:
                gap0, pinf0, dinf0 = gap, pinf, dinf
        if r <= p - 1:                                     # :95-98
            Y = V[:, :r] * e[:r]
            p = r
        nne = min(int(np.sum(dS < 0)), delta)              # :99
        if line_search == 1:
            U = np.hstack([np.zeros((n, p)), vS[:, :nne]])  # :101
        p = p + nne
        if line_search == 1:
            Y = np.hstack([Y, np.zeros((n, nne))])         # :105
        else:
            Y = np.hstack([Y, alpha * vS[:, :nne]])        # :107
        if pinf < tau1 * gradnorm:                         # :109-113
            sigma = max(sigma / gama, sigma_min)
        elif pinf > tau2 * gradnorm:
            sigma = min(sigma * gama, sigma_max)
    data.update({"Y": Y, "y": y, "S": S, "gap": gap, "pinf": pinf, "dinf": dinf,
                 "gradnorm": gradnorm, "time": time.time() - t0, "X": X, "sigma": sigma})     # X of the evaluated point (:59), as the reference returns it
    if data["status"] == 0 and eta_kkt > tol:
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y, obj, data


# ------------------------------------------------------------------------ multiblock (src/primal/ManiSDP_multiblock.m)
class BlockVec:
    """A point / tangent vector of the product manifold: the cell array ``Y{1..nb}`` of the reference, block i an
    (n_i, p_i) array (the bytes of MATLAB's p_i x n_i).  Supports exactly the arithmetic Manopt's tCG performs through
    ``M.lincomb`` (src/C-files/lincombc.cpp:3-67: a1*u1 [+ a2*u2] block by block)."""

    def __init__(self, blocks):
        self.b = list(blocks)

    def __add__(self, o):
        return BlockVec([x + y for x, y in zip(self.b, o.b)])

    def __sub__(self, o):
        return BlockVec([x - y for x, y in zip(self.b, o.b)])

    def __mul__(self, a):
        return BlockVec([x * a for x in self.b])

    __rmul__ = __mul__

    def __neg__(self):
        return BlockVec([-x for x in self.b])

    def copy(self):
        return BlockVec([x.copy() for x in self.b])


class MultiBlockManifold:
    """src/basicfunction/multiblockmanifold.m:1-42 over the MEX helpers of src/C-files: the first ``nob`` blocks are
    oblique (unit rows here = unit columns of MATLAB's p x n), the others Euclidean.
    Quirk Q5: the shipped projc.cpp source subtracts ONE Frobenius inner product per block
    (``:36-43``), which is the projection of a sphere, not of the oblique manifold whose retraction retrc.cpp applies
    column by column; the sources are stale against the (Windows-only) binaries.  This restatement uses the per-column
    projection that is consistent with the retraction and with the single-block code (ManiSDP_unitdiag.m:180)."""

    def __init__(self, pset, nset, nob):
        self.pset, self.nset, self.nob = list(pset), list(nset), int(nob)

    def inner(self, x, u, v):
        return float(sum(float(a.ravel() @ b.ravel()) for a, b in zip(u.b, v.b)))    # innerc.cpp:3-33

    def norm(self, x, d):
        return math.sqrt(self.inner(x, d, d))                                          # :10

    def typicaldist(self):                                                             # :11-15
        t = math.pi * sum(self.nset[:self.nob])
        t += sum(p * n for p, n in zip(self.pset[self.nob:], self.nset[self.nob:]))
        return math.sqrt(t)

    def proj(self, x, u):                                                              # :17-21
        out = []
        for i, (xi, ui) in enumerate(zip(x.b, u.b)):
            out.append(ui - xi * np.sum(xi * ui, axis=1, keepdims=True) if i < self.nob else ui.copy())
        return BlockVec(out)

    tangent = proj

    def retr(self, x, u):                                                              # :23-26 (retrc.cpp:4-52)
        out = []
        for i, (xi, ui) in enumerate(zip(x.b, u.b)):
            yi = xi + ui
            out.append(yi / np.sqrt(np.sum(yi * yi, axis=1, keepdims=True)) if i < self.nob else yi)
        return BlockVec(out)

    def zerovec(self, x):                                                              # :38-41
        return BlockVec([np.zeros_like(xi) for xi in x.b])

    def rand(self, rng):                                                               # :33-36 (randc.cpp:30-83)
        out = []
        for i, (p, n) in enumerate(zip(self.pset, self.nset)):
            yi = rng.standard_normal((n, p))
            out.append(yi / np.sqrt(np.sum(yi * yi, axis=1, keepdims=True)) if i < self.nob else yi)
        return BlockVec(out)


class _MultiBlockProblem:
    """cost / grad / hess closures of ManiSDP_multiblock.m:208-249 (shared variables ``Axb``, ``S``; ``eG`` in the store)."""

    def __init__(self, At, b, c, nset, nob):
        self.At = At.tocsc()
        self.A = self.At.T.tocsr()
        self.b, self.c = b, c
        self.nset, self.nob = list(nset), int(nob)
        self.off = np.concatenate([[0], np.cumsum([n * n for n in self.nset])])
        self.M = None
        self.y = np.zeros(b.size)
        self.sigma = 1.0
        self.Axb = None
        self.S = None
        self.eG = None
        self.nhess = 0

    def _x(self, Y):
        return np.concatenate([(yi @ yi.T).ravel(order="F") for yi in Y.b])           # :209-214

    def cost(self, Y):
        x = self._x(Y)
        self.Axb = self.A @ x - self.b - self.y / self.sigma                           # :215
        return float(self.c @ x) + 0.5 * self.sigma * float(self.Axb @ self.Axb)       # :216

    def grad(self, Y):
        tt = self.c + self.sigma * (self.At @ self.Axb)                                # :220
        self.S, self.eG, G = [], [], []
        for i, (yi, n) in enumerate(zip(Y.b, self.nset)):
            Si = tt[self.off[i]:self.off[i + 1]].reshape((n, n), order="F")           # :223
            Gi = 2.0 * (Si.T @ yi)                                                     # :224  G{i} = 2*Y{i}*S{i}
            if i < self.nob:
                eGi = np.sum(yi * Gi, axis=1, keepdims=True)                           # :226
                Gi = Gi - yi * eGi                                                     # :227
            else:
                eGi = None
            self.S.append(Si); self.eG.append(eGi); G.append(Gi)
        return BlockVec(G)

    def hess(self, Y, U):
        self.nhess += 1
        YU = np.concatenate([(yi @ ui.T).ravel(order="F") for yi, ui in zip(Y.b, U.b)])    # :235-237 T = Y{i}'*U{i}
        AyU = self.At @ (self.A @ YU)                                                  # :240
        H = []
        for i, (yi, ui, n) in enumerate(zip(Y.b, U.b, self.nset)):
            Hi = 2.0 * (self.S[i].T @ ui)                                              # :238
            Ai = AyU[self.off[i]:self.off[i + 1]].reshape((n, n), order="F")
            Hi = Hi + 4.0 * self.sigma * (Ai.T @ yi)                                   # :243
            if i < self.nob:
                Hi = Hi - yi * np.sum(yi * Hi, axis=1, keepdims=True) - ui * self.eG[i]    # :245
            H.append(Hi)
        return BlockVec(H)


def ManiSDP_multiblock(At, b, c, K, options=None, rng=None, verbose=False):
    """``[X, obj, data] = ManiSDP_multiblock(At, b, c, K, options)`` (src/primal/ManiSDP_multiblock.m:7; defaults
    :10-27): ``K['s']`` = block orders, ``K['nob']`` = number of leading unit-diagonal blocks.  Returns (Y, obj, data)
    with ``Y`` the list of factors (n_i, p_i); ``options['Y0']`` optionally fixes the start point."""
    o = dict(options or {})
    nset = [int(v) for v in np.atleast_1d(K["s"])]
    nob = int(K.get("nob", 0))
    nb = len(nset)
    min_facsize = o.get("min_facsize", 2); p0 = list(np.atleast_1d(o.get("p0", np.ones(nb, int))).astype(int))
    AL_maxiter = o.get("AL_maxiter", 1000); gama = o.get("gama", 2)
    sigma0 = o.get("sigma0", 1e-1); sigma_min = o.get("sigma_min", 1e-2); sigma_max = o.get("sigma_max", 1e7)
    tol = o.get("tol", 1e-8); theta = o.get("theta", 1e-2); delta = o.get("delta", 8)
    alpha = o.get("alpha", 0.1); tolgradnorm = o.get("tolgradnorm", 1e-8)
    TR_maxinner = o.get("TR_maxinner", 20); TR_maxiter = o.get("TR_maxiter", 4)
    tau1 = o.get("tau1", 1e1); tau2 = o.get("tau2", 1e1); line_search = o.get("line_search", 0)
    rng = rng or np.random.default_rng(0)
    b = _as_dense_vec(b)
    c = _as_dense_vec(c)
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {max(nset)}, m = {b.size}")
    prob = _MultiBlockProblem(At, b, c, nset, nob)
    A, Atc, off = prob.A, prob.At, prob.off
    p = [p0[i] if nset[i] >= min_facsize else nset[i] for i in range(nb)]              # :34-39
    sigma = sigma0
    y = np.zeros(b.size)
    normb = 1.0 + np.linalg.norm(b)
    Y = o.get("Y0", None)
    U = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None
    obj = gap = pinf = dinf = gradnorm = eta_kkt = None
    X = S = None

    def co(Yv):                                                                        # :160-169
        x = prob._x(Yv)
        Axb = A @ x - b - y / sigma
        return float(c @ x) + 0.5 * sigma * float(Axb @ Axb)

    def normalise(Yb, i):
        return Yb / np.sqrt(np.sum(Yb * Yb, axis=1, keepdims=True)) if i < nob else Yb

    def do_line_search(Yv, Uv):                                                        # :171-193
        a = 1.0
        cost0 = co(Yv)
        nY = BlockVec([normalise(yi + a * ui, i) for i, (yi, ui) in enumerate(zip(Yv.b, Uv.b))])
        k = 1
        while k <= 15 and co(nY) - cost0 > -1e-3:
            a = 0.8 * a
            nY = BlockVec([normalise(yi + a * ui, i) for i, (yi, ui) in enumerate(zip(Yv.b, Uv.b))])
            k += 1
        return nY

    for it in range(1, AL_maxiter + 1):                                                # :57
        prob.M = MultiBlockManifold(p, nset, nob)                                      # :58
        prob.y, prob.sigma = y, sigma
        if U is not None:
            Y = do_line_search(Y, U)                                                   # :59-61
        t1 = time.time()
        Y, _, info = trustregions(prob, Y, TR_maxiter, TR_maxinner, tolgradnorm, rng=rng)   # :62
        data["rtr_seconds"] += time.time() - t1
        data["hessvecs"] += info.hessvecs
        data["cost_evals"] += info.cost_evals
        data["rejected"] += info.rejected
        gradnorm = info.gradnorm                                                       # :63
        X = [yi @ yi.T for yi in Y.b]                                                  # :65-69
        x = np.concatenate([Xi.ravel(order="F") for Xi in X])
        obj = float(c @ x)                                                             # :70
        Axb = A @ x - b                                                                # :71
        pinf = float(np.linalg.norm(Axb)) / normb                                      # :72
        y = y - sigma * Axb                                                            # :73
        cy = c - Atc @ y                                                               # :74
        by = float(b @ y)                                                              # :75
        S, vS, dS, dinfs = [], [], [], []
        for i, n in enumerate(nset):                                                   # :78-88
            Si = cy[off[i]:off[i + 1]].reshape((n, n), order="F")
            if i < nob:
                z = np.sum(X[i] * Si, axis=0)                                          # :81
                by += float(np.sum(z))
                Si = Si - np.diag(z)
            w, V = np.linalg.eigh(0.5 * (Si + Si.T))                                   # :86
            S.append(Si); dS.append(w); vS.append(V)
            dinfs.append(max(0.0, -w[0]) / (1.0 + abs(w[-1])))                         # :87
        dinf = max(dinfs)                                                              # :89
        gap = abs(obj - by) / (abs(by) + abs(obj) + 1.0)                               # :90
        _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, p_max:%d, sigma:%0.3f, time:%0.2fs"
             % (it, obj, gap, pinf, dinf, gradnorm, max(p), sigma, time.time() - t0))
        eta_kkt = max(gap, pinf, dinf)                                                 # :93
        data["iters"] = it
        data.setdefault("log", []).append((it, obj, gap, pinf, dinf, gradnorm, max(p), sigma))
        Y_eval = Y
        if eta_kkt < tol:
            _say(verbose, "Optimality is reached!")
            break
        if it % 50 == 0:                                                               # :98-108
            if it > 100 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                data["status"] = 2
                _say(verbose, "Slow progress!")
                break
            gap0, pinf0, dinf0 = gap, pinf, dinf
        newY, newU = [], []
        for i, n in enumerate(nset):                                                   # :109-147
            yi = Y.b[i]
            ui = None
            if n >= min_facsize:
                if p[i] > 1:
                    V, e, _ = np.linalg.svd(yi, full_matrices=False)                   # :112-118
                    r = int(np.sum(e >= theta * e[0]))
                    if r == 0:
                        r = 1
                    if r < p[i]:
                        yi = V[:, :r] * e[:r]                                          # :125
                        p[i] = r
                nneg = int(np.sum(dS[i] < 0))
                nne = max(min(nneg, delta), 1) if i < nob else min(nneg, delta)        # :129-133
                if p[i] + nne > n:
                    nne = 0                                                            # :134-136
                if line_search == 1:
                    ui = np.hstack([np.zeros((n, p[i])), vS[i][:, :nne]])              # :137-139
                    yi = np.hstack([yi, np.zeros((n, nne))])                           # :141-142
                else:
                    yi = normalise(np.hstack([yi, alpha * vS[i][:, :nne]]), i)         # :143-147
                p[i] = p[i] + nne                                                      # :140
            newY.append(yi)
            newU.append(ui if ui is not None else np.zeros_like(yi))
        Y = BlockVec(newY)
        U = BlockVec(newU) if line_search == 1 else None
        if pinf < tau1 * gradnorm:                                                     # :150-154
            sigma = max(sigma / gama, sigma_min)
        elif pinf > tau2 * gradnorm:
            sigma = min(sigma * gama, sigma_max)
    data.update({"Y": Y_eval, "X": X, "y": y, "S": S, "gap": gap, "pinf": pinf, "dinf": dinf, "gradnorm": gradnorm,
                 "time": time.time() - t0, "sigma": sigma, "p": list(p)})
    if data["status"] == 0 and eta_kkt > tol:
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y_eval, obj, data


# ------------------------------------------------------------------ dual, unit diagonal
class _DualUnitDiagProblem:
    """cost/grad/hess closures of src/dual/ManiDSDP_unitdiag.m:174-194.  ``A`` is the
    m x n^2 PSD part, ``B`` the m x K.f free part, ``iA = (diag(dAAt)\\A)'`` (:38).  The
    closures share ``As, Af, X, eG`` through the parent workspace; ``X`` and ``eG`` are set
    by ``grad`` and used by ``hess`` (at the last accepted point)."""

    def __init__(self, A, B, b, c, cf, dAAt, n, p):
        self.A = sp.csr_matrix(A)
        self.At = self.A.T.tocsr()
        self.B = sp.csr_matrix(B)
        self.iAt = sp.diags(1.0 / np.asarray(dAAt, dtype=np.float64)) @ self.A      # iA' = D^-1 A   (:38)
        self.bA = self.iAt.T @ b                                                 # :39
        self.b, self.c, self.cf, self.n = b, c, cf, n
        self.M = ObliqueNT(p, n, inner_all=False)
        self.x = np.zeros(n * n)
        self.w = np.zeros(cf.size)
        self.sigma = 1.0
        self.As = self.Af = self.X = self.YeG = None
        self.nhess = 0

    def parts(self, Y):
        S = Y @ Y.T                                        # :175  S = Y'*Y
        sc = S.ravel(order="F") - self.c                   # :176
        y = self.iAt @ sc                                  # :177
        return S, sc, y

    def cost(self, Y):
        _, sc, y = self.parts(Y)
        self.As = self.At @ y - sc - self.x / self.sigma   # :178
        self.Af = self.B.T @ y - self.cf - self.w / self.sigma      # :179
        return float(self.b @ y) + 0.5 * self.sigma * (float(self.As @ self.As) + float(self.Af @ self.Af))   # :180

    def grad(self, Y):
        n = self.n
        self.X = (self.bA - self.sigma * self.As).reshape((n, n), order="F")      # :184
        eG = 2.0 * (self.X.T @ Y)                          # :185  eG = 2*Y*X
        self.YeG = np.sum(Y * eG, axis=1, keepdims=True)
        return eG - Y * self.YeG                           # :186

    def hess(self, Y, U):
        self.nhess += 1
        n = self.n
        YU = Y @ U.T                                       # :190  YU = Y'*U
        yAU = (self.At @ (self.iAt @ YU.ravel(order="F"))).reshape((n, n), order="F")   # :191
        eH = (2.0 * (self.X.T @ U) - 4.0 * self.sigma * (yAU.T @ Y)
              + 2.0 * self.sigma * (Y @ (U.T @ Y) + U @ (Y.T @ Y)))               # :192
        return eH - Y * np.sum(Y * eH, axis=1, keepdims=True) - U * self.YeG     # :193


def ManiDSDP_unitdiag(A, b, c, K, options=None, rng=None, verbose=False):
    """``[X, obj, data] = ManiDSDP_unitdiag(A, b, c, K, options)`` (src/dual/ManiDSDP_unitdiag.m:8-172):
    ``A`` is m x (K.f + K.s^2), ``c`` has K.f + K.s^2 entries.  Returns (X, obj, data); the factor
    of ``S`` is ``data['Y']`` (n x p)."""
    o = dict(options or {})
    n = int(K["s"]); nf = int(K.get("f", 0))
    b = _as_dense_vec(b)
    call = _as_dense_vec(c)
    m = b.size
    p0 = o.get("p0", int(math.ceil(math.log(m))))          # :11
    maxiter = o.get("ADMM_maxiter", 300); gama = o.get("gama", 2)
    sigma0 = o.get("sigma0", 1e-3); sigma_min = o.get("sigma_min", 1e-3); sigma_max = o.get("sigma_max", 1e7)
    tol = o.get("tol", 1e-8); theta = o.get("theta", 1e-3); delta = o.get("delta", 8)
    alpha = o.get("alpha", 0.1); tolgradnorm = o.get("tolgradnorm", 1e-8)
    TR_maxinner = o.get("TR_maxinner", 20); TR_maxiter = o.get("TR_maxiter", 4)
    tau1 = o.get("tau1", 1e1); tau2 = o.get("tau2", 1e2); line_search = o.get("line_search", 0)
    rng = rng or np.random.default_rng(0)
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {n}, m = {m}")
    normc = 1.0 + np.linalg.norm(call)                     # :32
    Aall = sp.csc_matrix(A)
    B = Aall[:, :nf]; Apsd = Aall[:, nf:]                  # :33-34
    cf = call[:nf]; c = call[nf:]                          # :35-36
    dAAt = o.get("dAAt", None)
    if dAAt is None:
        dAAt = np.asarray(Apsd.multiply(Apsd).sum(axis=1)).ravel()     # :37  diag(A*A')
    p = p0
    prob = _DualUnitDiagProblem(Apsd, B, b, c, cf, dAAt, n, p)
    sigma = sigma0
    Y = o.get("Y0", None)
    U = None
    fac_size = []; seta = []
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0, "log": []}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None

    def rownorm(Z):
        return Z / np.sqrt(np.sum(Z ** 2, axis=1, keepdims=True))

    def co(Yv):                                            # :131-138
        _, sc, yv = prob.parts(Yv)
        As = prob.At @ yv - sc - prob.x / sigma
        Af = prob.B.T @ yv - cf - prob.w / sigma
        return float(b @ yv) + 0.5 * sigma * (float(As @ As) + float(Af @ Af))

    def do_line_search(Yv, Uv):                            # :140-152
        a = 1.0
        cost0 = co(Yv)
        i = 1
        nY = rownorm(Yv + a * Uv)
        while i <= 15 and co(nY) - cost0 > -1e-3:
            a = 0.8 * a
            nY = rownorm(Yv + a * Uv)
            i += 1
        return nY

    obj = gap = pinf = dinf = gradnorm = eta = None
    X = S = y = None
    for it in range(1, maxiter + 1):                       # :62
        fac_size.append(p)
        prob.M = ObliqueNT(p, n, inner_all=False)          # :64
        prob.sigma = sigma
        if U is not None:
            Y = do_line_search(Y, U)                       # :65-67
        t1 = time.time()
        Y, _, info = trustregions(prob, Y, TR_maxiter, TR_maxinner, tolgradnorm, rng=rng)   # :68
        data["rtr_seconds"] += time.time() - t1
        data["hessvecs"] += info.hessvecs; data["cost_evals"] += info.cost_evals; data["rejected"] += info.rejected
        gradnorm = info.gradnorm
        Yeval = Y
        S, sc, y = prob.parts(Y)                           # :70-72
        As = prob.At @ y - sc                              # :73
        Af = prob.B.T @ y - cf                             # :74
        pinf = (np.linalg.norm(As) + np.linalg.norm(Af)) / normc     # :75
        by = float(b @ y)                                  # :76
        prob.x = prob.x - sigma * As                       # :77
        prob.w = prob.w - sigma * Af                       # :78
        eX = (prob.x + prob.bA).reshape((n, n), order="F")  # :79
        z = np.sum(S * eX, axis=0)                         # :80
        X = eX - np.diag(z)                                # :81
        dX, vX = np.linalg.eigh(X)                         # :82
        obj = float(c @ eX.ravel(order="F")) + float(cf @ prob.w) + float(np.sum(z))   # :85
        dinf = max(0.0, -dX[0]) / (1.0 + abs(dX[-1]))      # :86
        gap = abs(obj - by) / (1.0 + abs(obj) + abs(by))   # :87
        V, e, r = _thin_svd_rank_strict(Y, theta)          # :88-90  (strict >)
        _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs"
             % (it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
        data["log"].append((obj, gap, pinf, dinf, gradnorm, r, p, sigma))
        eta = max(gap, pinf, dinf)                         # :93
        seta.append(eta)
        data["iters"] = it
        if eta < tol:
            _say(verbose, "Optimality is reached!")
            break
        if it % 50 == 0:                                   # :99-109
            if it > 100 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                data["status"] = 2
                _say(verbose, "Slow progress!")
                break
            else:
                gap0, pinf0, dinf0 = gap, pinf, dinf
        if r <= p - 1:                                     # :110-113
            Y = V[:, :r] * e[:r]
            p = r
        nne = max(min(int(np.sum(dX < 0)), delta), 1)      # :114
        if line_search == 1:
            U = np.hstack([np.zeros((n, p)), vX[:, :nne]])  # :116
        p = p + nne
        if line_search == 1:
            Y = np.hstack([Y, np.zeros((n, nne))])         # :120
        else:
            Y = rownorm(np.hstack([Y, alpha * vX[:, :nne]]))    # :122-123
        if pinf < tau1 * gradnorm:                         # :125-129
            sigma = max(sigma / gama, sigma_min)
        elif pinf > tau2 * gradnorm:
            sigma = min(sigma * gama, sigma_max)
    data.update({"X": X, "y": y, "S": S, "w": prob.w.copy(), "gap": gap, "pinf": pinf, "dinf": dinf, "gradnorm": gradnorm,
                 "time": time.time() - t0, "fac_size": fac_size, "seta": seta, "Y": Yeval, "sigma": sigma})
    if data["status"] == 0 and eta > tol:
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiDSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return X, obj, data


def _thin_svd_rank_strict(Y, theta):
    """``[~, D, V] = svd(Y)`` + ``r = sum(e > theta*e(1))`` (ManiDSDP_unitdiag.m:88-90: strict, unlike the primal files)."""
    V, e, _ = np.linalg.svd(Y, full_matrices=False)
    return V, e, int(np.sum(e > theta * e[0]))
