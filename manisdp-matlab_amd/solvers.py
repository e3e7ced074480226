"""Host-side mirrors of the reference's three primal entry points.

Same names, option names, defaults, printed protocol and ``data`` fields as
``src/primal/ManiSDP_onlyunitdiag.m``, ``ManiSDP_unitdiag.m`` and
``ManiSDP_unittrace.m``.  The augmented-Lagrangian loop and the bookkeeping
(multiplier / sigma updates, KKT residues, rank cut, escape set-up) stay on the
host exactly as in the reference; every ``trustregions(problem, Y, opts)`` call
(``ManiSDP_onlyunitdiag.m:43``, ``ManiSDP_unitdiag.m:57``, ``ManiSDP_unittrace.m:57``)
is replaced by ``msdp_rtr`` on the HIP library, with the factor resident in HBM.

Factors are NumPy ``(n, p)`` arrays (for the oblique kinds the bytes of MATLAB's
``p x n`` column-major ``Y``).  Returned ``X`` is the factor ``Y`` (``X = Y Y'``);
``data['X']`` holds the dense matrix only when ``n <= options['dense_X_max']``.
"""
from __future__ import annotations

import math
import time

import numpy as np
import scipy.sparse as sp

from . import _lib


# Several replicated host loops in ONE process (the in-process ranks of tests/test_gpu_local_ranks.py: N threads, one BLAS
# pool) must not race for BLAS threads: how many each call gets would decide the summation order of the p x p Gram / thin-SVD
# arithmetic, and replicated loops must produce the same bits.  The harness sets this to 1; one process per GPU never needs it.
FORCED_HOST_THREADS = None


def _host_threads():
    """Context manager capping the BLAS/LAPACK threads of the host-side AL bookkeeping (thin SVD, residues) at the
    CPUs this process may actually use.  On a 256-core box inside a 16-CPU cgroup quota the default (256 threads on
    20000 x 40 matrices) burns the quota and gets the thread that feeds the GPU throttled."""
    import contextlib
    import os
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                      # pragma: no cover
        return contextlib.nullcontext()
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:                                      # pragma: no cover
        ncpu = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = min(ncpu, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    if FORCED_HOST_THREADS:
        return threadpool_limits(limits=int(FORCED_HOST_THREADS))
    return threadpool_limits(limits=max(1, min(ncpu, 16)))

__all__ = ["ManiSDP_onlyunitdiag", "ManiSDP_unitdiag", "ManiSDP_unittrace", "ManiSDP", "ManiSDP_multiblock",
           "ManiDSDP_unitdiag", "DEFAULTS", "DATA_FIELDS"]

# Option defaults of the reference's entry points: ManiSDP_onlyunitdiag.m:8-17,
# ManiSDP_unitdiag.m:10-26, ManiSDP_unittrace.m:10-25, ManiSDP.m:9-25.
DEFAULTS = {
    "onlyunitdiag": dict(p0=2, AL_maxiter=20, tol=1e-8, theta=1e-1, delta=8, alpha=0.5, tolgradnorm=1e-8,
                         TR_maxinner=100, TR_maxiter=40, line_search=0),
    "unitdiag": dict(p0=2, AL_maxiter=300, gama=2, sigma0=1e-3, sigma_min=1e-2, sigma_max=1e7, tol=1e-8,
                     theta=1e-3, delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20, TR_maxiter=4,
                     tau1=1, tau2=1, line_search=0),
    "unittrace": dict(p0=1, AL_maxiter=1000, gama=2, sigma0=1e1, sigma_min=1e2, sigma_max=1e7, tol=1e-8,
                      theta=1e-2, delta=8, alpha=0.05, tolgradnorm=1e-8, TR_maxinner=40, TR_maxiter=3,
                      tau1=1e-5, tau2=1e-4, line_search=1),
    "generic": dict(p0=1, AL_maxiter=1000, gama=2, sigma0=1e-2, sigma_min=1e-1, sigma_max=1e7, tol=1e-8,
                    theta=1e-2, delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20, TR_maxiter=4,
                    tau1=1e-2, tau2=1e-1, line_search=1, solver=0),
}
# Fields of the reference's `data` output (ManiSDP_onlyunitdiag.m:86-95, ManiSDP_unitdiag.m:114-127,
# ManiSDP_unittrace.m:119-131); the Python mirrors add counters (hessvecs, rtr_seconds, ...) next to them.
DATA_FIELDS = {
    "onlyunitdiag": ("X", "S", "z", "dinf", "gradnorm", "time", "status"),
    "unitdiag": ("X", "y", "S", "z", "gap", "pinf", "dinf", "gradnorm", "time", "fac_size", "status"),
    "unittrace": ("X", "y", "S", "z", "gap", "pinf", "dinf", "gradnorm", "time", "status"),
    "generic": ("X", "y", "S", "gap", "pinf", "dinf", "gradnorm", "time", "status"),
}


def _join_comm(h, comm):
    """options['comm']: (nranks, rank, rccl_unique_id) -- one process per GPU -- or ("local", nranks, rank, group) -- the
    in-process stand-in of msdp_comm_init_local: N handles of one process on one GPU, one host thread each."""
    if comm[0] == "local":
        h.comm_init_local(int(comm[1]), int(comm[2]), int(comm[3]))
    else:
        h.comm_init(int(comm[0]), int(comm[1]), comm[2])


def _say(verbose, msg):
    if verbose:
        print(msg, flush=True)


def _rtr_opts(o):
    return _lib.default_opts(maxiter=int(o["TR_maxiter"]), maxinner=int(o["TR_maxinner"]),
                             tolgradnorm=float(o["tolgradnorm"]))


_RANK_CUT_SVD = False


def _thin_svd_rank(Y, theta, gram=None):
    """svd(Y) and r = sum(e >= theta*e(1)) (ManiSDP_onlyunitdiag.m:52-54) through the
    p x p Gram matrix: never forms the n x n V of the reference.  ``gram``: the Gram matrix
    as the device computed it (msdp_factor_gram), Y is not needed then."""
    if _RANK_CUT_SVD and gram is None:
        _, e, Qt = np.linalg.svd(Y, full_matrices=False)
        return Qt.T, e, int(np.sum(e >= theta * e[0]))
    G = gram if gram is not None else Y.T @ Y
    w, Q = np.linalg.eigh(G)
    order = np.argsort(w)[::-1]
    w = np.maximum(w[order], 0.0)
    Q = Q[:, order]
    e = np.sqrt(w)
    r = int(np.sum(e >= theta * e[0]))
    return Q, e, r


def _rank_cut(Y, Q, e, r):
    """Y = V(:,1:r)'.*e(1:r): with Y = V diag(e) Q', V(:,k) e_k = Y Q(:,k)."""
    return Y @ Q[:, :r]


def _extreme_eigs_host(S, k, dense_max):
    """lambda_min side (k smallest eigenpairs) and lambda_max of S on the host.
    n <= dense_max: LAPACK like the reference's eig(full(S)) (ManiSDP_onlyunitdiag.m:50)."""
    n = S.shape[0]
    if n <= dense_max:
        Sd = S.toarray() if sp.issparse(S) else np.asarray(S)
        dS, vS = np.linalg.eigh(Sd)
        return dS, vS, int(np.sum(dS < 0))
    raise RuntimeError("host eigensolver limit exceeded; use the device escape (options['eig'] = 'device')")


def _device_escape(h, run, o, data, default_tol, default_maxit):
    """One device escape call (``run(tol, maxit)`` -> lam, V, lam_max, steps) with the convergence contract of
    ``msdp_escape_info``: a Lanczos run that hit ``maxit`` only bounds lambda_min from above, so it is repeated
    once with four times the step budget; ``certified`` tells the AL loop whether dinf may end the solve."""
    tol = float(o.get("eig_tol", default_tol))
    maxit = int(o.get("eig_maxit", default_maxit))
    lam, vS, lam_max, _ = run(tol, maxit)
    nvalid, conv, res = h.escape_info()
    if not conv:
        data["eig_retries"] = data.get("eig_retries", 0) + 1
        lam, vS, lam_max, _ = run(tol, 4 * maxit)
        nvalid, conv, res = h.escape_info()
    if not conv:
        data["eig_unconverged"] = data.get("eig_unconverged", 0) + 1
    return lam, vS, lam_max, nvalid, conv


def _verify_lambda_min(h, run1, o, data, default_tol, default_maxit, dense_n=0):
    """The independent check behind "Optimality is reached!".  The regular escape call deflates span(Y) and starts
    from what the previous call found: fast, but its lambda_min is only as good as S*Y is small and it can stay above
    the true one.  Before dinf may end the solve -- and once at the end of a solve that did not converge, so that the
    reported dinf is the true one -- lambda_min and lambda_max are recomputed without those shortcuts:
      * affine kinds with a dense S of moderate order (dense_n <= options.verify_dense_max, default 4000): the
        reference's own eig(S) (ManiSDP_unitdiag.m:68) on the host, on the S the device holds;
      * sparse C (block eigen-solver, msdp_blockeig.hip): the subspace iteration once more from a block of hashed noise --
        no column of Y, nothing carried over from earlier calls, four times tighter tolerance;
      * otherwise plain Lanczos runs on S itself: no deflation, nothing carried over from earlier calls; the start
        vector is a hashed random combination of the columns of Y plus 5 % hashed noise (span(Y), the near-kernel of S
        at a near-stationary point, is where a lambda_min the deflated estimate missed would live).
    Returns (lambda_min, its eigenvector as an n x 1 array, lambda_max, converged)."""
    t1 = time.time()
    data["eig_verifications"] = data.get("eig_verifications", 0) + 1
    if 0 < dense_n <= int(o.get("verify_dense_max", 4000)):
        S = h.get_dual_slack()
        w, V = np.linalg.eigh(0.5 * (S + S.T))
        data["eig_seconds"] += time.time() - t1
        return float(w[0]), V[:, :1], float(w[-1]), True
    h.set_option("escape_deflate", 0)
    h.set_option("escape_warm", 0)
    # block eigen-solver (sparse C): hashed noise only, no column of Y -- the cold run costs ~2400 filter steps on G81 and is
    # independent of everything the regular calls used; Lanczos path: span(Y) + 5 % noise (33 000 -> 22 000 steps on G81)
    h.set_option("escape_start_y", int(o.get("verify_start_in_span_y", 0 if h.escape_method() == 1 else 1)))
    try:
        lam, vS, lam_max, _ = run1(float(o.get("eig_tol", default_tol)), int(o.get("eig_maxit", default_maxit)))
        _, conv, _ = h.escape_info()
        if not conv:                                      # once more with four times the step budget
            lam, vS, lam_max, _ = run1(float(o.get("eig_tol", default_tol)), 4 * int(o.get("eig_maxit", default_maxit)))
            _, conv, _ = h.escape_info()
    finally:
        h.set_option("escape_deflate", 1)
        h.set_option("escape_warm", 1)
        h.set_option("escape_start_y", 0)
    data["eig_seconds"] += time.time() - t1
    return float(lam[0]), vS[:, :1], float(lam_max), conv


# =============================================================== onlyunitdiag
def ManiSDP_onlyunitdiag(C, options=None, verbose=True, rng=None):
    with _host_threads():
        return _onlyunitdiag_impl(C, options, verbose, rng)


def _onlyunitdiag_impl(C, options=None, verbose=True, rng=None):
    """``[X, obj, data] = ManiSDP_onlyunitdiag(C, options)`` (reference
    src/primal/ManiSDP_onlyunitdiag.m:6).  Extra, optional option fields that the
    reference does not have: ``Y0`` (start point instead of ``randn``), ``eig``
    (``'host'`` dense LAPACK | ``'device'`` few-eigenvector escape; default by size),
    ``dense_X_max``."""
    o = dict(options or {})
    for k, v in DEFAULTS["onlyunitdiag"].items():
        o.setdefault(k, v)
    dense_max = int(o.get("dense_eig_max", 3000))          # the host eig(S) is allowed up to this order (options['eig'] = 'host') ...
    dense_default = int(o.get("dense_eig_default", 600))   # ... and the default up to this one (round 6: G1, n = 800, 0.124 s with it, 0.047 s
                                                           # with the device escape + its independent check; the affine kinds switch at 400 / 600)
    dense_X_max = int(o.get("dense_X_max", 4000))
    rng = rng or np.random.default_rng(0)

    _say(verbose, "ManiSDP is starting...")
    n = C.shape[0]
    _say(verbose, f"SDP size: n = {n}, m = {n}")
    from .problems import SyntheticDenseC
    comm = o.get("comm")
    pcap = max(32, int(o["p0"]) + 2 * int(o["delta"]))
    if isinstance(C, SyntheticDenseC):
        # BASELINE config 5: every rank generates its rows of the dense C on the device; the matrix never exists on the host
        # (only the host eigen-solver of small test problems asks for it)
        eig_mode = o.get("eig", "host" if (n <= dense_default and comm is None) else "device")
        Csp = C.toarray() if eig_mode == "host" else None
        nr, rk = (1, 0) if comm is None else ((int(comm[1]), int(comm[2])) if comm[0] == "local" else (int(comm[0]), int(comm[1])))
        h = _lib.Handle.dense_synthetic(n, C.seed, nranks=nr, rank=rk, pcap=pcap)
    else:
        Csp = C.tocsr() if sp.issparse(C) else np.asarray(C, dtype=np.float64)
        eig_mode = o.get("eig", "host" if n <= dense_default else "device")
        h = _lib.Handle.onlyunitdiag(Csp, pcap=pcap)
    # options['comm'] = (nranks, rank, unique_id): rows of the factor and of C sharded over the ranks (msdp_comm_init); this
    # host loop then runs replicated -- same start point (pass Y0 or seed rng identically), same data, same decisions on
    # every rank; the escape runs replicated on a full copy of the sparse C (device eigen-solver only)
    if comm is not None:
        if not (sp.issparse(Csp) or isinstance(C, SyntheticDenseC)):
            h.close()
            raise ValueError("row-sharded solves need a sparse C or a problems.SyntheticDenseC")
        eig_mode = "device"
        _join_comm(h, comm)
        if o.get("halo_exchange"):                         # only the rows this rank's rows of C reference travel before S*U
            h.set_option("halo_exchange", 1)
    if "escape_method" in o:                               # 0 auto, 1 Lanczos, 2 block eigen-solver also below its size threshold
        h.set_option("escape_method", int(o["escape_method"]))
    for name, value in (o.get("device_options") or {}).items():     # run-time switches of the handle (msdp_set_option): A/B runs, tests
        h.set_option(name, int(value))
    topts = _rtr_opts(o)
    p = int(o["p0"])
    Y = o.get("Y0", None)
    if Y is None:                                          # trustregions.m:390-392 -> M.rand()
        Y = rng.standard_normal((n, p))
        Y /= np.sqrt(np.sum(Y * Y, axis=1, keepdims=True))
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    U = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0,
            "eig_seconds": 0.0, "log": []}
    t0 = time.time()
    dinf0 = None
    obj = dinf = gradnorm = None
    z = S = None
    certified = True
    last_verified = True
    # Rank decision, rank cut and widening of the factor on the device (msdp_factor_gram / _rotate / _append): the factor
    # stays resident between the trustregions() calls and comes to the host once, when the solve ends.  Default with the
    # device escape (large n); the line-search variant keeps the host form (its direction U is a host array anyway).
    device_factor = bool(o.get("device_factor", eig_mode == "device" and o["line_search"] != 1))
    resident = False
    try:
        for it in range(1, int(o["AL_maxiter"]) + 1):      # :38
            if not resident:
                h.set_point(Y)
            if U is not None:                              # :40-42 line_search
                _line_search(h, U)
            st = h.rtr(topts)                              # :43
            data["rtr_seconds"] += st.seconds
            data["hessvecs"] += st.hessvecs
            data["cost_evals"] += st.cost_evals
            data["rejected"] += st.rejected
            gradnorm = st.gradnorm                         # :44
            if not device_factor:
                Y = (h.get_point_all() if comm is not None else h.get_point())
                Y_eval = Y                                 # X = Y'*Y of :45 -- what the reference returns (:86)
            z = h.get_z_all() if comm is not None else h.get_z()    # :46-47  z = sum((Y*C).*Y)
            obj = float(np.sum(z))                         # :48
            t1 = time.time()
            certified = True
            if eig_mode == "host":
                S = (Csp - sp.diags(z)) if sp.issparse(Csp) else (Csp - np.diag(z))   # :49
                dS, vS, nneg = _extreme_eigs_host(S, int(o["delta"]), dense_max)     # :50
                lam_min, lam_max = dS[0], dS[-1]
            elif eig_mode == "host_sparse":
                # diagnostic: ARPACK shift-invert on the host (sparse LU of S - sigma I), k bottom eigenpairs
                import scipy.sparse.linalg as spla
                S = (Csp - sp.diags(z)).tocsc()
                k = int(o["delta"])
                lam_max = float(spla.eigsh(S, k=1, which="LA", return_eigenvectors=False, tol=1e-6)[0])
                lam, vS = spla.eigsh(S, k=k, sigma=-1e-3 - 0.0 * lam_max, which="LM", tol=1e-12)
                order = np.argsort(lam); lam = lam[order]; vS = vS[:, order]
                lam_min = lam[0]
                nneg = int(np.sum(lam < 0))
            else:
                k = int(o["delta"])
                lam, vS, lam_max, _, certified = _device_escape(
                    h, lambda tol, maxit: h.escape_eigs(k, tol=tol, maxit=maxit), o, data, 1e-9, 60000)
                lam_min = lam[0]
                nneg = int(np.sum(lam < 0))            # missing pairs come back as +inf
                S = None
                data["escape_method"] = h.escape_method()      # 1: block eigen-solver, 0: Lanczos
            data["eig_seconds"] += time.time() - t1
            dinf = max(0.0, -lam_min) / (1.0 + lam_max)    # :51
            last_verified = eig_mode != "device"
            # the regular escape call is warm-started (columns of Y and the previous call's vectors in its start block, or
            # span(Y) deflated): before dinf may end the solve -- always, the estimate of the same call is never taken as a
            # certificate -- lambda_min is recomputed by a cold-started, undeflated run
            if eig_mode == "device" and certified and (dinf < o["tol"] or it == int(o["AL_maxiter"])):
                last_verified = True
                lam_v, v_v, lmax_v, certified = _verify_lambda_min(
                    h, lambda tol, maxit: h.escape_eigs(1, tol=tol, maxit=maxit), o, data, 1e-9, 60000)
                dinf_v = max(0.0, -lam_v) / (1.0 + lmax_v)
                if dinf_v >= o["tol"] > dinf:              # the deflated run had missed the bottom of the spectrum
                    vS = np.hstack([v_v, vS[:, :max(int(o["delta"]) - 1, 0)]])
                    nneg = max(nneg, 1)
                dinf = dinf_v
            if device_factor:
                Q, e, r = _thin_svd_rank(None, float(o["theta"]), gram=h.factor_gram())   # :52-54 from the device's Gram matrix
            else:
                Q, e, r = _thin_svd_rank(Y, float(o["theta"]))  # :52-54
            _say(verbose, "Iter %d, obj:%0.8f, dinf:%0.1e, r:%d, p:%d, time:%0.2fs"
                 % (it, obj, dinf, r, p, time.time() - t0))
            data["log"].append((it, obj, dinf, r, p, time.time() - t0, st.hessvecs))
            data["iters"] = it
            if dinf < o["tol"] and certified:              # :57-60 (an unconverged Lanczos run certifies nothing)
                _say(verbose, "Optimality is reached!")
                if device_factor:
                    Y_eval = (h.get_point_all() if comm is not None else h.get_point())
                break
            if it % 20 == 0:                               # :61-69
                if it > 50 and dinf > dinf0:
                    data["status"] = 2
                    _say(verbose, "Slow progress!")
                    if device_factor:
                        Y_eval = (h.get_point_all() if comm is not None else h.get_point())
                    break
                dinf0 = dinf
            nne = max(min(nneg, int(o["delta"])), 1)       # :74
            if device_factor:
                if it == int(o["AL_maxiter"]):
                    Y_eval = (h.get_point_all() if comm is not None else h.get_point())                 # last pass: the evaluated point, before it is re-shaped
                try:
                    if r <= p - 1:                         # :70-73
                        h.factor_rotate(Q[:, :r])
                        p = r
                    h.factor_append(vS[:, :nne], float(o["alpha"]), normalize=True)   # :78-83
                    p = p + nne
                    resident = True
                    continue
                except _lib.MsdpError as err:              # wider than the handle's buffers: re-enter through set_point
                    if "allocated capacity" not in str(err):
                        raise
                    Y = (h.get_point_all() if comm is not None else h.get_point())                      # the factor as the device holds it (already cut)
                    Y = np.hstack([Y, o["alpha"] * vS[:, :nne]])
                    Y = np.ascontiguousarray(Y / np.sqrt(np.sum(Y * Y, axis=1, keepdims=True)))
                    p = Y.shape[1]
                    resident = False
                    continue
            if r <= p - 1:                                 # :70-73
                Y = _rank_cut(Y, Q, e, r)
                p = r
            if o["line_search"] == 1:                      # :75-77
                U = np.hstack([np.zeros((n, p)), vS[:, :nne]])
            p = p + nne                                    # :78
            if o["line_search"] == 1:
                Y = np.hstack([Y, np.zeros((n, nne))])     # :80
            else:
                Y = np.hstack([Y, o["alpha"] * vS[:, :nne]])   # :82
                Y = Y / np.sqrt(np.sum(Y * Y, axis=1, keepdims=True))   # :83
            Y = np.ascontiguousarray(Y)
        if obj is not None and not last_verified and certified:   # a solve that stopped on "Slow progress": report the true dinf
            lam_v, _, lmax_v, certified = _verify_lambda_min(
                h, lambda tol, maxit: h.escape_eigs(1, tol=tol, maxit=maxit), o, data, 1e-9, 60000)
            dinf = max(0.0, -lam_v) / (1.0 + lmax_v)
    finally:
        h.close()
    if obj is not None:
        Y = Y_eval          # the point the residues belong to (the loop's last pass has already widened its own copy)
    if S is None and Csp is not None and sp.issparse(Csp) and z is not None:
        S = Csp - sp.diags(z)                              # :49 (kept sparse; the reference returns full(S))
    data.update({"Y": Y, "S": S, "z": z, "dinf": dinf, "gradnorm": gradnorm,
                 "time": time.time() - t0, "p": Y.shape[1],
                 "X": (Y @ Y.T if n <= dense_X_max else None)})   # :45,86
    if data["status"] == 0 and (dinf > o["tol"] or not certified):   # :92-95
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y, obj, data


def _line_search(h, U):
    """line_search(Y, U) (ManiSDP_onlyunitdiag.m:103-115, ManiSDP_unitdiag.m:138-150,
    ManiSDP_unittrace.m:142-154) with every co() evaluated on the device; leaves the
    accepted trial point resident."""
    alpha = 1.0
    cost0 = h.linesearch_cost(None, 0.0)
    i = 1
    val = h.linesearch_cost(U, alpha)
    while i <= 15 and val - cost0 > -1e-3:
        alpha = 0.8 * alpha
        val = h.linesearch_cost(U, alpha)
        i += 1
    h.linesearch_accept()


# =================================================================== unitdiag
def _dense_vec(v):
    if sp.issparse(v):
        return np.asarray(v.todense()).ravel()
    return np.asarray(v, dtype=np.float64).ravel()


def _affine_common(kind, At, b, c, K, options, verbose, rng, defaults):
    with _host_threads():
        return _affine_impl(kind, At, b, c, K, options, verbose, rng, defaults)


def _affine_impl(kind, At, b, c, K, options, verbose, rng, defaults):
    o = dict(options or {})
    for k, v in defaults.items():
        o.setdefault(k, v)
    n = int(K["s"])
    rng = rng or np.random.default_rng(0)
    b = _dense_vec(b)
    c = _dense_vec(c)
    Atc = sp.csc_matrix(At)
    A = Atc.T.tocsr()
    sphere = kind == _lib.KIND_UNITTRACE
    generic = kind == _lib.KIND_GENERIC                # src/primal/ManiSDP.m: Euclidean manifold, no z term
    eig_mode = o.get("eig", "host" if n <= int(o.get("dense_eig_max", 400)) else "device")
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {n}, m = {b.size}")
    h = _lib.Handle.affine(kind, Atc, b, c, n, pcap=max(32, int(o["p0"]) + 2 * int(o["delta"])))
    # options['comm'] = (nranks, rank, unique_id): one process per GPU, rows of the factor sharded over the ranks
    # (msdp_comm_init).  Every rank runs this same host loop on identical data -- the operator state is replicated on
    # the device, the start point must be the same on all ranks (pass Y0 or seed rng identically) -- and reaches the same
    # decisions; collectives happen inside the library calls.
    comm = o.get("comm")
    if comm is not None:
        _join_comm(h, comm)
    if "escape_method" in o:                               # 0 auto (Lanczos on the explicit S of these kinds), 1 Lanczos, 2 block eigen-solver
        h.set_option("escape_method", int(o["escape_method"]))
    for name, value in (o.get("device_options") or {}).items():     # run-time switches of the handle (msdp_set_option): A/B runs, tests
        h.set_option(name, int(value))
    topts = _rtr_opts(o)
    p = int(o["p0"])
    sigma = float(o["sigma0"])
    gama = float(o["gama"])
    y = np.zeros(b.size)
    normb = 1.0 + np.linalg.norm(b)
    Y = o.get("Y0", None)
    if Y is None:
        Y = rng.standard_normal((n, p))
        if generic:
            pass                                           # euclideanfactory.m:82 (M.rand = randn)
        elif sphere:
            Y /= np.linalg.norm(Y)                         # spherefactory.m:249-254
        else:
            Y /= np.sqrt(np.sum(Y * Y, axis=1, keepdims=True))
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    U = None
    fac_size = []
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0,
            "eig_seconds": 0.0, "log": []}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None
    obj = gap = pinf = dinf = gradnorm = eta_kkt = None
    S = z = None
    slow_every, slow_after = (20, 50) if (sphere or generic) else (50, 100)
    certified = True
    last_verified = True
    try:
        for it in range(1, int(o["AL_maxiter"]) + 1):
            fac_size.append(p)
            h.set_multipliers(y, sigma)
            h.set_point(Y)
            if U is not None:
                _line_search(h, U)
            st = h.rtr(topts)
            data["rtr_seconds"] += st.seconds
            data["hessvecs"] += st.hessvecs
            data["cost_evals"] += st.cost_evals
            data["rejected"] += st.rejected
            gradnorm = st.gradnorm
            Y = h.get_point_all() if comm is not None else h.get_point()
            Y_eval = Y                                     # X of :59 -- what the reference returns (:114)
            dev_al = (eig_mode == "device") and bool(o.get("device_al", True))
            certified = True
            if dev_al:
                # obj, A x, eS, z and S from the device kernels (no n x n work on the host)
                obj, Ax = h.al_primal(b.size)              # :59-61
                Axb = Ax - b                               # :62
                pinf = float(np.linalg.norm(Axb)) / normb  # :63
                y = y - sigma * Axb                        # :64
                t1 = time.time()
                zz = h.al_dual(y)                          # :65-67 (S stays on the device)
                if generic:
                    by = float(b @ y)
                elif sphere:
                    z = zz
                    by = float(b @ y) + z
                else:
                    z = zz
                    by = float(b @ y) + float(np.sum(z))
                lam, vS, lam_max, _, certified = _device_escape(
                    h, lambda tol, maxit: h.escape_eigs_dual(int(o["delta"]), tol=tol, maxit=maxit), o, data,
                    1e-10, 20000)                          # :68
                dS = np.concatenate([lam, [lam_max]])
                S = None
                data["eig_seconds"] += time.time() - t1
            else:
                X = Y @ Y.T                                # unitdiag :59 / unittrace :59
                x = X.ravel(order="F")
                obj = float(c @ x)                         # :61
                Axb = A @ x - b                            # :62
                pinf = float(np.linalg.norm(Axb)) / normb  # :63
                y = y - sigma * Axb                        # :64
                eS = (c - Atc @ y).reshape((n, n), order="F")  # :65
                t1 = time.time()
            if dev_al:
                pass
            elif generic:
                S = eS                                     # ManiSDP.m:64
                by = float(b @ y)                          # :67
            elif sphere:
                z = float(np.sum(eS * X))                  # unittrace :66
                S = eS - z * np.eye(n)                     # :67
                by = float(b @ y) + z                      # :70
            else:
                z = np.sum(X * eS, axis=0)                 # unitdiag :66
                S = eS - np.diag(z)                        # :67
                by = float(b @ y) + float(np.sum(z))       # :70
            if dev_al:
                pass
            elif eig_mode == "device":
                # few-eigenvector escape on the device instead of the O(n^3) eig(S) of :68
                lam, vS, lam_max, _, certified = _device_escape(
                    h, lambda tol, maxit: h.escape_eigs_matrix(S, int(o["delta"]), tol=tol, maxit=maxit), o, data,
                    1e-10, 20000)
                dS = np.concatenate([lam, [lam_max]])      # dS[0] = lambda_min ... dS[-1] = lambda_max
                data["eig_seconds"] += time.time() - t1
            else:
                dS, vS = np.linalg.eigh(S)                 # :68
                data["eig_seconds"] += time.time() - t1
            dinf = max(0.0, -dS[0]) / (1.0 + dS[-1])       # :69
            gap = abs(obj - by) / (abs(by) + abs(obj) + 1.0)   # :71
            last_verified = not dev_al
            if dev_al and certified and ((max(gap, pinf, dinf) < o["tol"]) or it == int(o["AL_maxiter"])):
                last_verified = True
                lam_v, v_v, lmax_v, certified = _verify_lambda_min(
                    h, lambda tol, maxit: h.escape_eigs_dual(1, tol=tol, maxit=maxit), o, data, 1e-10, 20000, dense_n=n)
                dinf_v = max(0.0, -lam_v) / (1.0 + lmax_v)
                if dinf_v >= o["tol"] > dinf:
                    vS = np.hstack([v_v, vS[:, :max(int(o["delta"]) - 1, 0)]])
                    dS = np.concatenate([[lam_v], dS[:-2], [lmax_v]])
                dinf = dinf_v
            Q, e, r = _thin_svd_rank(Y, float(o["theta"]))     # :72-74
            _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs"
                 % (it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
            data["log"].append((it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
            eta_kkt = max(gap, pinf, dinf)                 # :77
            data["iters"] = it
            if "iter_hook" in o:                           # diagnostics (tools/): sees the loop's local state
                o["iter_hook"](locals())
            if eta_kkt < o["tol"] and certified:
                _say(verbose, "Optimality is reached!")
                break
            if it % slow_every == 0:                       # unitdiag :82-92 / unittrace :86-96
                if it > slow_after and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                    data["status"] = 2
                    _say(verbose, "Slow progress!")
                    break
                gap0, pinf0, dinf0 = gap, pinf, dinf
            if r <= p - 1:                                 # :93-96
                Y = _rank_cut(Y, Q, e, r)
                p = r
            nneg = int(np.sum(dS[:-1] < 0)) if eig_mode == "device" else int(np.sum(dS < 0))
            if sphere or generic:
                nne = min(nneg, int(o["delta"]))           # unittrace :101 / ManiSDP.m:99
            else:
                nne = max(min(nneg, int(o["delta"])), 1)   # unitdiag :97
            if o["line_search"] == 1:
                U = np.hstack([np.zeros((n, p)), vS[:, :nne]])
            p = p + nne
            if o["line_search"] == 1:
                Y = np.hstack([Y, np.zeros((n, nne))])
            else:
                Y = np.hstack([Y, o["alpha"] * vS[:, :nne]])
                if generic:
                    pass                                   # ManiSDP.m:107
                elif sphere:
                    Y = Y / np.linalg.norm(Y)              # unittrace :110
                else:
                    Y = Y / np.sqrt(np.sum(Y * Y, axis=1, keepdims=True))   # unitdiag :106
            Y = np.ascontiguousarray(Y)
            if pinf < o["tau1"] * gradnorm:                # :108-112
                sigma = max(sigma / gama, o["sigma_min"])
            elif pinf > o["tau2"] * gradnorm:
                sigma = min(sigma * gama, o["sigma_max"])
        if obj is not None and not last_verified and certified:   # stopped on "Slow progress": report the true dinf
            lam_v, _, lmax_v, certified = _verify_lambda_min(
                h, lambda tol, maxit: h.escape_eigs_dual(1, tol=tol, maxit=maxit), o, data, 1e-10, 20000, dense_n=n)
            dinf = max(0.0, -lam_v) / (1.0 + lmax_v)
        if S is None and obj is not None and n <= int(o.get("dense_X_max", 6000)):
            S = h.get_dual_slack()                         # data.S of the reference (:116), from the device
    finally:
        h.close()
    if obj is not None:
        Y = Y_eval          # the point the residues belong to (the loop's last pass has already widened its own copy)
    data.update({"Y": Y, "X": (Y @ Y.T if n <= int(o.get("dense_X_max", 6000)) else None), "y": y, "S": S, "z": z, "gap": gap, "pinf": pinf, "dinf": dinf,
                 "gradnorm": gradnorm, "time": time.time() - t0, "sigma": sigma})
    if not sphere and not generic:
        data["fac_size"] = fac_size
    if data["status"] == 0 and (eta_kkt > o["tol"] or not certified):
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y, obj, data


def ManiSDP_unitdiag(At, b, c, K, options=None, verbose=True, rng=None):
    """``[X, obj, data] = ManiSDP_unitdiag(At, b, c, K, options)`` (reference
    src/primal/ManiSDP_unitdiag.m:7; defaults :10-26)."""
    return _affine_common(_lib.KIND_UNITDIAG, At, b, c, K, options, verbose, rng, DEFAULTS["unitdiag"])


def ManiSDP_unittrace(At, b, c, K, options=None, verbose=True, rng=None):
    """``[X, obj, data] = ManiSDP_unittrace(At, b, c, K, options)`` (reference
    src/primal/ManiSDP_unittrace.m:7; defaults :10-25)."""
    return _affine_common(_lib.KIND_UNITTRACE, At, b, c, K, options, verbose, rng, DEFAULTS["unittrace"])


def ManiSDP(At, b, c, K, options=None, verbose=True, rng=None):
    """``[X, obj, data] = ManiSDP(At, b, c, K, options)`` -- the generic entry point on the Euclidean manifold
    (reference src/primal/ManiSDP.m:6; defaults :9-25).  Same device kernels as the two structured affine entry
    points with the projection / retraction terms switched off."""
    return _affine_common(_lib.KIND_GENERIC, At, b, c, K, options, verbose, rng, DEFAULTS["generic"])


# ================================================================= multiblock
DEFAULTS["multiblock"] = dict(min_facsize=2, AL_maxiter=1000, gama=2, sigma0=1e-1, sigma_min=1e-2, sigma_max=1e7,
                              tol=1e-8, theta=1e-2, delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20,
                              TR_maxiter=4, tau1=1e1, tau2=1e1, line_search=0)       # ManiSDP_multiblock.m:10-27 (+ p0 = ones)
DATA_FIELDS["multiblock"] = ("X", "y", "S", "gap", "pinf", "dinf", "gradnorm", "time", "status")   # :156-163


def _pack_blocks(blocks, r0, N, pmax):
    """Cell array of factors (n_i, p_i) -> one (N, pmax) array, zero beyond each block's own width."""
    Y = np.zeros((N, pmax))
    for i, Yi in enumerate(blocks):
        Y[r0[i]:r0[i + 1], :Yi.shape[1]] = Yi
    return Y


def ManiSDP_multiblock(At, b, c, K, options=None, verbose=True, rng=None):
    """``[X, obj, data] = ManiSDP_multiblock(At, b, c, K, options)`` (reference src/primal/ManiSDP_multiblock.m:7):
    block-diagonal X with unit diagonal on the first ``K['nob']`` blocks of orders ``K['s']``.  The product manifold
    of ``multiblockmanifold.m`` lives on the device as ONE factor of N = sum n_i rows whose blocks are zero-padded to a
    common width; the per-block bookkeeping of the outer loop (eig(S{i}), svd(Y{i}), escape directions; :78-147) stays
    on the host -- the blocks are small by construction.  Returns (list of factors, obj, data)."""
    with _host_threads():
        return _multiblock_impl(At, b, c, K, options, verbose, rng)


def _multiblock_impl(At, b, c, K, options, verbose, rng):
    o = dict(options or {})
    for k, v in DEFAULTS["multiblock"].items():
        o.setdefault(k, v)
    nset = [int(v) for v in np.atleast_1d(K["s"])]
    nob = int(K.get("nob", 0))
    nb = len(nset)
    p0 = [int(v) for v in np.atleast_1d(o.get("p0", np.ones(nb, int)))]
    rng = rng or np.random.default_rng(0)
    b = _dense_vec(b)
    c = _dense_vec(c)
    Atc = sp.csc_matrix(At)
    r0 = np.concatenate([[0], np.cumsum(nset)]).astype(int)
    N = int(r0[-1])
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {max(nset)}, m = {b.size}")
    p = [p0[i] if nset[i] >= o["min_facsize"] else nset[i] for i in range(nb)]        # :34-39
    h = _lib.Handle.multiblock(Atc, b, c, nset, nob, pcap=max(32, max(p) + 2 * int(o["delta"])))
    # options["block_eig"]: "host" = the reference's loop of eig(S{i}) on the host (LAPACK through NumPy), "device" = all blocks in
    # one launch on the GPU (msdp_block_eigs: one workgroup per block; Householder tridiagonalisation, bisection, inverse iteration
    # for the `delta` <= 8 vectors the loop uses), "auto" (default) = device from 16 blocks of order <= 256 on, host below -- where the
    # oracle-parity tests compare iterate by iterate: the eigenvectors of two eigen-solvers differ by signs / rotations inside
    # eigenspaces
    be = o.get("block_eig", "auto")
    block_eig_device = be == "device" or (be == "auto" and nb >= 16 and max(nset) <= 256)
    sigma = float(o["sigma0"]); gama = float(o["gama"])
    y = np.zeros(b.size)
    normb = 1.0 + np.linalg.norm(b)

    def normalise(Yi, i):
        return Yi / np.sqrt(np.sum(Yi * Yi, axis=1, keepdims=True)) if i < nob else Yi

    Yb = o.get("Y0", None)
    if Yb is None:                                          # trustregions.m:390-392 -> M.rand() (randc.cpp)
        Yb = [normalise(rng.standard_normal((nset[i], p[i])), i) for i in range(nb)]
    Yb = [np.ascontiguousarray(Yi, dtype=np.float64) for Yi in Yb]
    Ub = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0, "eig_seconds": 0.0, "log": []}
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None
    obj = gap = pinf = dinf = gradnorm = eta_kkt = None
    X = S = Y_eval = None
    try:
        for it in range(1, int(o["AL_maxiter"]) + 1):       # :57
            pmax = max(p)
            # M.typicaldist of multiblockmanifold.m:11-15
            tdist = math.sqrt(math.pi * sum(nset[:nob]) + sum(pi * ni for pi, ni in zip(p[nob:], nset[nob:])))
            topts = _lib.default_opts(maxiter=int(o["TR_maxiter"]), maxinner=int(o["TR_maxinner"]),
                                      tolgradnorm=float(o["tolgradnorm"]), Delta_bar=tdist)
            h.set_multipliers(y, sigma)
            h.set_point(_pack_blocks(Yb, r0, N, pmax))
            if Ub is not None:
                _line_search(h, _pack_blocks(Ub, r0, N, pmax))   # :59-61, 171-193
            st = h.rtr(topts)                               # :62
            data["rtr_seconds"] += st.seconds
            data["hessvecs"] += st.hessvecs
            data["cost_evals"] += st.cost_evals
            data["rejected"] += st.rejected
            gradnorm = st.gradnorm                          # :63
            Yfull = h.get_point()
            Yb = [np.ascontiguousarray(Yfull[r0[i]:r0[i + 1], :p[i]]) for i in range(nb)]
            Y_eval = Yb
            obj, Ax = h.al_primal(b.size)                   # :65-70
            Axb = Ax - b                                    # :71
            pinf = float(np.linalg.norm(Axb)) / normb       # :72
            y = y - sigma * Axb                             # :73
            t1 = time.time()
            z = h.al_dual(y)                                # :74-84 on the device: S = cy blocks - diag(z), z = 0 on free rows
            by = float(b @ y) + float(np.sum(z))            # :75,82
            S, vS, dS, dinfs = [], [], [], []
            if block_eig_device:
                # eig(S{i}) of every block in one launch (msdp_block_eigs: cyclic Jacobi, one workgroup per block); what the loop
                # below uses of it -- all eigenvalues, the eigenvectors of the `delta` smallest -- is what comes back
                try:
                    wall, Vall = h.block_eigs(r0[:-1], nset, int(o["delta"]))
                except _lib.MsdpError:
                    if o.get("block_eig", "auto") == "device":
                        raise
                    block_eig_device = False
            if block_eig_device:
                for i in range(nb):
                    w = wall[r0[i]:r0[i + 1]]
                    dS.append(w); vS.append(Vall[r0[i]:r0[i + 1], :])
                    dinfs.append(max(0.0, -w[0]) / (1.0 + abs(w[-1])))   # :87
            else:
                for i in range(nb):                         # :78-88 (only the diagonal blocks come to the host)
                    Si = h.get_dual_slack_block(r0[i], nset[i])
                    w, V = np.linalg.eigh(0.5 * (Si + Si.T))    # :86
                    S.append(Si); dS.append(w); vS.append(V)
                    dinfs.append(max(0.0, -w[0]) / (1.0 + abs(w[-1])))   # :87
            data["eig_seconds"] += time.time() - t1
            dinf = max(dinfs)                               # :89
            gap = abs(obj - by) / (abs(by) + abs(obj) + 1.0)   # :90
            _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, p_max:%d, sigma:%0.3f, time:%0.2fs"
                 % (it, obj, gap, pinf, dinf, gradnorm, max(p), sigma, time.time() - t0))
            data["log"].append((it, obj, gap, pinf, dinf, gradnorm, max(p), sigma, time.time() - t0))
            eta_kkt = max(gap, pinf, dinf)                  # :93
            data["iters"] = it
            if eta_kkt < o["tol"]:
                _say(verbose, "Optimality is reached!")
                break
            if it % 50 == 0:                                # :98-108
                if it > 100 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                    data["status"] = 2
                    _say(verbose, "Slow progress!")
                    break
                gap0, pinf0, dinf0 = gap, pinf, dinf
            newY, newU = [], []
            for i, n in enumerate(nset):                    # :109-147
                Yi = Yb[i]
                Ui = None
                if n >= o["min_facsize"]:
                    if p[i] > 1:
                        Q, e, r = _thin_svd_rank(Yi, float(o["theta"]))   # :112-121
                        r = max(r, 1)
                        if r < p[i]:
                            Yi = _rank_cut(Yi, Q, e, r)     # :125
                            p[i] = r
                    nneg = int(np.sum(dS[i] < 0))
                    nne = max(min(nneg, int(o["delta"])), 1) if i < nob else min(nneg, int(o["delta"]))   # :129-133
                    if p[i] + nne > n:
                        nne = 0                             # :134-136
                    if o["line_search"] == 1:
                        Ui = np.hstack([np.zeros((n, p[i])), vS[i][:, :nne]])    # :137-139
                        Yi = np.hstack([Yi, np.zeros((n, nne))])                 # :141-142
                    else:
                        Yi = normalise(np.hstack([Yi, o["alpha"] * vS[i][:, :nne]]), i)   # :143-147
                    p[i] = p[i] + nne                       # :140
                newY.append(np.ascontiguousarray(Yi))
                newU.append(Ui if Ui is not None else np.zeros_like(Yi))
            Yb = newY
            Ub = newU if o["line_search"] == 1 else None
            if pinf < o["tau1"] * gradnorm:                 # :150-154
                sigma = max(sigma / gama, o["sigma_min"])
            elif pinf > o["tau2"] * gradnorm:
                sigma = min(sigma * gama, o["sigma_max"])
        if block_eig_device and Y_eval is not None:         # data.S (:158): the blocks of the last iterate, fetched once
            S = [h.get_dual_slack_block(r0[i], nset[i]) for i in range(nb)]
    finally:
        h.close()
    if Y_eval is not None:
        X = [Yi @ Yi.T for Yi in Y_eval]                    # :65-69, 156
    data.update({"Y": Y_eval, "X": X, "y": y, "S": S, "gap": gap, "pinf": pinf, "dinf": dinf, "gradnorm": gradnorm,
                 "time": time.time() - t0, "sigma": sigma, "p": [Yi.shape[1] for Yi in (Y_eval or [])]})
    if data["status"] == 0 and eta_kkt > o["tol"]:
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return Y_eval, obj, data


# ================================================================= dual approach, unit diagonal
DEFAULTS["dual_unitdiag"] = dict(ADMM_maxiter=300, gama=2, sigma0=1e-3, sigma_min=1e-3, sigma_max=1e7, tol=1e-8, theta=1e-3,
                                 delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20, TR_maxiter=4, tau1=1e1, tau2=1e2,
                                 line_search=0)      # ManiDSDP_unitdiag.m:10-26 (+ p0 = ceil(log(m)), :11)
DATA_FIELDS["dual_unitdiag"] = ("X", "y", "S", "w", "gap", "pinf", "dinf", "gradnorm", "time", "fac_size", "seta", "status")   # :132-144


def ManiDSDP_unitdiag(A, b, c, K, options=None, verbose=True, rng=None):
    """``[X, obj, data] = ManiDSDP_unitdiag(A, b, c, K, options)`` (reference src/dual/ManiDSDP_unitdiag.m:8): the dual
    approach for SDPs whose dual slack has a unit diagonal.  ``A`` is m x (K['f'] + K['s']^2) -- free columns first --
    and ``c`` has K['f'] + K['s']^2 entries.  The Riemannian subproblem (cost/grad/hess :174-194 on the oblique factor
    of S), the line search and the outer-step bookkeeping (:70-81) run on the device; the multiplier matrix x never
    leaves it.  Returns (X, obj, data); ``data['Y']`` is the factor of S (n x p)."""
    with _host_threads():
        return _dual_unitdiag_impl(A, b, c, K, options, verbose, rng)


def _dual_unitdiag_impl(A, b, c, K, options, verbose, rng):
    o = dict(options or {})
    for k, v in DEFAULTS["dual_unitdiag"].items():
        o.setdefault(k, v)
    n = int(K["s"]); nf = int(K.get("f", 0))
    b = _dense_vec(b)
    call = _dense_vec(c)
    m = b.size
    o.setdefault("p0", int(math.ceil(math.log(m))))        # :11
    rng = rng or np.random.default_rng(0)
    _say(verbose, "ManiSDP is starting...")
    _say(verbose, f"SDP size: n = {n}, m = {m}")
    normc = 1.0 + np.linalg.norm(call)                     # :32
    Aall = sp.csc_matrix(A)
    B = Aall[:, :nf]; Apsd = sp.csr_matrix(Aall[:, nf:])   # :33-34
    cf = call[:nf]; cpsd = call[nf:]                       # :35-36
    dAAt = o.get("dAAt", None)
    if dAAt is None:
        dAAt = np.asarray(Apsd.multiply(Apsd).sum(axis=1)).ravel()      # :37
    # eig(X) (:82): the reference's dense eig on the host up to n = 600 (0.5 ms there), beyond that the device escape on
    # the resident X with the independent lambda_min check before the solve may stop (d = 60, n = 1831: 1.6 s instead of 5.0 s)
    dense_max = int(o.get("dense_eig_max", 600))
    eig_mode = o.get("eig", "host" if n <= dense_max else "device")
    p = int(o["p0"])
    delta = int(o["delta"])
    h = _lib.Handle.dual_unitdiag(Apsd, b, cpsd, dAAt, B if nf else None, cf, pcap=max(32, p + 2 * delta))
    topts = _rtr_opts(o)
    sigma = float(o["sigma0"]); gama = float(o["gama"])
    w = np.zeros(nf)
    Y = o.get("Y0", None)
    if Y is None:                                          # trustregions.m:390-392 -> M.rand()
        Y = rng.standard_normal((n, p))
        Y /= np.sqrt(np.sum(Y * Y, axis=1, keepdims=True))
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    U = None
    data = {"status": 0, "hessvecs": 0, "cost_evals": 0, "rejected": 0, "rtr_seconds": 0.0, "eig_seconds": 0.0, "log": []}
    fac_size, seta = [], []
    t0 = time.time()
    gap0 = pinf0 = dinf0 = None
    obj = gap = pinf = dinf = gradnorm = eta = None
    Y_eval = None
    certified = True
    try:
        for it in range(1, int(o["ADMM_maxiter"]) + 1):    # :62
            fac_size.append(p)
            h.dual_set_penalty(sigma, w)
            h.set_point(Y)
            if U is not None:
                _line_search(h, U)                         # :65-67, 164-172
            st = h.rtr(topts)                              # :68
            data["rtr_seconds"] += st.seconds
            data["hessvecs"] += st.hessvecs; data["cost_evals"] += st.cost_evals; data["rejected"] += st.rejected
            gradnorm = st.gradnorm                         # :69
            Y = h.get_point()
            Y_eval = Y
            by, cex, as2, Af, z = h.dual_outer_step()      # :70-81 (x <- x - sigma*As on the device)
            pinf = (math.sqrt(as2) + float(np.linalg.norm(Af))) / normc      # :75
            w = w - sigma * Af                             # :78
            obj = cex + float(cf @ w) + float(np.sum(z))   # :85
            t1 = time.time()
            certified = True
            if eig_mode == "host":
                Xd = h.get_dual_slack()
                dX, vX = np.linalg.eigh(0.5 * (Xd + Xd.T)) # :82
                lam_min, lam_max = float(dX[0]), float(dX[-1])
                nneg = int(np.sum(dX < 0))
            else:
                lam, vX, lam_max, _, certified = _device_escape(
                    h, lambda tol, maxit: h.escape_eigs_dual(delta, tol=tol, maxit=maxit), o, data, 1e-10, 20000)
                lam_min = float(lam[0])
                nneg = int(np.sum(lam < 0))
            dinf = max(0.0, -lam_min) / (1.0 + abs(lam_max))     # :86
            if eig_mode != "host" and certified and (dinf < o["tol"] or it == int(o["ADMM_maxiter"])):
                lam_v, v_v, lmax_v, certified = _verify_lambda_min(
                    h, lambda tol, maxit: h.escape_eigs_dual(1, tol=tol, maxit=maxit), o, data, 1e-10, 20000, dense_n=n)
                dinf_v = max(0.0, -lam_v) / (1.0 + abs(lmax_v))
                if dinf_v >= o["tol"] > dinf:
                    vX = np.hstack([v_v, vX[:, :max(delta - 1, 0)]])
                    nneg = max(nneg, 1)
                dinf = dinf_v
            data["eig_seconds"] += time.time() - t1
            gap = abs(obj - by) / (1.0 + abs(obj) + abs(by))     # :87
            if _RANK_CUT_SVD:
                _, e, Qt = np.linalg.svd(Y, full_matrices=False); Q = Qt.T
            else:
                Q, e, _ = _thin_svd_rank(Y, float(o["theta"]))
            r = int(np.sum(e > float(o["theta"]) * e[0]))  # :88-90 (strict, unlike the primal entry points)
            _say(verbose, "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs"
                 % (it, obj, gap, pinf, dinf, gradnorm, r, p, sigma, time.time() - t0))
            data["log"].append((obj, gap, pinf, dinf, gradnorm, r, p, sigma))
            eta = max(gap, pinf, dinf)                     # :93
            seta.append(eta)
            data["iters"] = it
            if eta < o["tol"] and certified:
                _say(verbose, "Optimality is reached!")
                break
            if it % 50 == 0:                               # :99-109
                if it > 100 and gap > gap0 and pinf > pinf0 and dinf > dinf0:
                    data["status"] = 2
                    _say(verbose, "Slow progress!")
                    break
                gap0, pinf0, dinf0 = gap, pinf, dinf
            if r <= p - 1:                                 # :110-113
                Y = _rank_cut(Y, Q, e, r)
                p = r
            nne = max(min(nneg, delta), 1)                 # :114
            if o["line_search"] == 1:
                U = np.hstack([np.zeros((n, p)), vX[:, :nne]])   # :116
            p = p + nne
            if o["line_search"] == 1:
                Y = np.hstack([Y, np.zeros((n, nne))])     # :120
            else:
                Y = np.hstack([Y, o["alpha"] * vX[:, :nne]])     # :122-123
                Y = Y / np.sqrt(np.sum(Y * Y, axis=1, keepdims=True))
            Y = np.ascontiguousarray(Y)
            if pinf < o["tau1"] * gradnorm:                # :125-129
                sigma = max(sigma / gama, float(o["sigma_min"]))
            elif pinf > o["tau2"] * gradnorm:
                sigma = min(sigma * gama, float(o["sigma_max"]))
        X = h.get_dual_slack() if obj is not None else None
        y = h.dual_get_y() if obj is not None else None
    finally:
        h.close()
    data.update({"X": X, "y": y, "S": (Y_eval @ Y_eval.T if Y_eval is not None and n <= int(o.get("dense_X_max", 4000)) else None),
                 "w": w, "gap": gap, "pinf": pinf, "dinf": dinf, "gradnorm": gradnorm, "time": time.time() - t0,
                 "fac_size": fac_size, "seta": seta, "Y": Y_eval, "sigma": sigma})
    if data["status"] == 0 and (eta is None or eta > o["tol"] or not certified):
        data["status"] = 1
        _say(verbose, "Iteration maximum is reached!")
    _say(verbose, "ManiDSDP: optimum = %0.8f, time = %0.2fs" % (obj, time.time() - t0))
    return X, obj, data
