"""ORACLE (test infrastructure, not product code) -- CPU restatement of the three
ManiSDP primal entry points, following the reference ``.m`` files line by line.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path never does.

Follows (paths relative to the reference tree):
  * src/primal/ManiSDP_onlyunitdiag.m:6-156
  * src/primal/ManiSDP_unitdiag.m:7-198
  * src/primal/ManiSDP_unittrace.m:7-177
  * manopt7.0/manopt/manifolds/sphere/spherefactory.m:83-153,220-232,249-254
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
