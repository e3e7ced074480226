"""The algebra behind the one-reduction tCG trip of the persistent kernel (manisdp-matlab_amd/csrc/msdp_pipe.h), restated in NumPy and
checked against the oracle's tCG (tCG.m:95-292) -- no GPU: what is tested here is that the REFORMULATION takes the decisions of tCG.m.

A trip of the kernel forms H*mdelta, publishes its rows, and reduces EIGHT inner products in one grid reduction; everything the second
reduction of tCG.m's trip carries (model value of the trial step, <r', r'>, tCG.m:227-241) is expanded in the step length, and the
products C*tangent(r), C*mdelta the next trip needs follow by linearity from the gathered rows of H*mdelta (refreshed from direct
products every `refresh` trips).  `tcg_one_reduction` below is that trip, statement by statement, on whole arrays."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import manisdp_ref as R, manopt_rtr


def tcg_one_reduction(C, Y, eG, grad, Delta, maxinner, kappa=0.1, theta=1.0, mininner=1, refresh=16):
    """msdp_pipe.h, tcg_pipe_body: returns (eta, Heta, trips, stop code) like oracle.manopt_rtr.tCG."""
    rowdot = lambda A, B: np.sum(A * B, axis=1, keepdims=True)
    tangent = lambda U: U - Y * rowdot(Y, U)
    dot = lambda A, B: float(np.sum(A * B))
    g = grad
    eta = np.zeros_like(Y)
    r = g.copy()
    md = g.copy()
    hmd = np.zeros_like(Y)
    gg = dot(g, g)
    z_r, d_Pd, e_Pd, e_Pe, model_value, alpha, beta = gg, gg, 0.0, 0.0, 0.0, 0.0, 0.0
    norm_r0 = math.sqrt(gg)
    j, stop = 0, 5
    first, direct = True, False
    ctr = cmd = None
    pub_tr = pub_md = None
    while True:
        # ---- the products (top of the trip): by linearity from the gathered rows of last trip's Hmd, or afresh
        if first:
            cmd = C @ md                                  # the first direction = the gradient
            ctr = cmd.copy()
        else:
            if direct:
                ctr = C @ pub_tr                          # published next to Hmd by the trip before
                cmd = C @ pub_md
            chq = C @ hmd                                 # the neighbours' rows of last trip's Hmd
            ctr = ctr - alpha * chq                       # C tangent(r') = C tangent(r) - alpha C Hmd
            cmd = ctr + beta * cmd                        # C md'         = C tangent(r') + beta C md
        pub = (not first) and refresh > 0 and ((j + 1) % refresh) == 0
        if pub:
            pub_tr, pub_md = tangent(r), md.copy()
        # ---- Hmd = proj(C*md) - md.*eG (tCG.m:163, ManiSDP_onlyunitdiag.m:127-130) and the eight sums of THE reduction
        hmd = cmd - Y * rowdot(Y, cmd) - md * eG
        rg = r - g
        v = [dot(md, hmd), dot(r, hmd), dot(hmd, hmd), dot(md, g), dot(eta, hmd), dot(md, rg), dot(r, r), dot(eta, g) + 0.5 * dot(eta, rg)]
        d_Hd, z_r, model_value = v[0], v[6], v[7]
        alpha = z_r / d_Hd if d_Hd != 0.0 else (math.copysign(math.inf, z_r) if z_r != 0 else math.nan)
        e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd                  # :173
        if d_Hd <= 0.0 or e_Pe_new >= Delta * Delta:                                # :183
            tau = (-e_Pd + math.sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd
            eta = eta - tau * md
            r = r - tau * hmd
            stop = 1 if d_Hd <= 0.0 else 2
            j += 1
            break
        new_model = model_value - alpha * v[3] - 0.5 * alpha * (v[4] + v[5]) + 0.5 * alpha * alpha * d_Hd     # :227, expanded
        r_r = max(z_r - 2.0 * alpha * v[1] + alpha * alpha * v[2], 0.0)                                       # :241, expanded
        e_Pe = e_Pe_new
        if new_model >= model_value:                                                # :228
            stop = 6
            j += 1
            break
        eta = eta - alpha * md
        r = r - alpha * hmd
        model_value = new_model
        j += 1
        norm_r = math.sqrt(r_r)
        if j >= mininner and norm_r <= norm_r0 * min(norm_r0 ** theta, kappa):      # :249
            stop = 3 if kappa < norm_r0 ** theta else 4
            break
        if j >= maxinner:
            break
        beta = r_r / z_r                                                            # :272
        e_Pd = beta * (e_Pd + alpha * d_Pd)
        d_Pd = r_r + beta * beta * d_Pd
        z_r = r_r
        md = tangent(r + beta * md)                                                 # :273, 283
        direct, first = pub, False
    return eta, r - g, j, stop


def _grid(nx, ny, seed):
    rng = np.random.default_rng(seed)
    n = nx * ny
    idx = np.arange(n).reshape(nx, ny)
    rows, cols, vals = [], [], []
    for a, b in ((idx, np.roll(idx, -1, axis=1)), (idx, np.roll(idx, -1, axis=0))):
        w = rng.choice([-1.0, 1.0], size=n)
        rows += [a.ravel(), b.ravel()]; cols += [b.ravel(), a.ravel()]; vals += [w, w]
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    return (sp.diags(np.asarray(abs(A).sum(axis=1)).ravel()) - A).tocsr() * 0.25


@pytest.mark.parametrize("shape,p,seed", [((12, 15), 6, 0), ((20, 30), 16, 1), ((9, 41), 32, 2), ((30, 30), 12, 3)])
def test_one_reduction_trip_takes_the_decisions_of_tcg(shape, p, seed):
    """Same trip count and stop code as tCG.m for every inner cap and trust-region radius (negative curvature, boundary, model and
    residual stops all occur), the step and Heta to rounding; with and without the refresh of the two products."""
    C = _grid(shape[0], shape[1], seed)
    n = C.shape[0]
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    prob = R._OnlyUnitDiagProblem(C, n, p, q1="correct")
    seen = set()
    for it in range(6):                                   # a few points along a solve: far from / close to a stationary point
        prob.cost(Y)
        g = prob.grad(Y)
        for Delta in (1e-2, 1.0, 1e3):
            for maxinner in (1, 3, 40):
                eta0, Heta0, j0, stop0 = manopt_rtr.tCG(prob, Y, g, Delta, maxinner)
                for refresh in (0, 4, 16):
                    eta1, Heta1, j1, stop1 = tcg_one_reduction(C, Y, prob.eG, g, Delta, maxinner, refresh=refresh)
                    assert (j1, stop1) == (j0, stop0), (it, Delta, maxinner, refresh)
                    scale = max(1.0, float(np.linalg.norm(eta0)))
                    assert np.linalg.norm(eta1 - eta0) <= 1e-9 * scale
                    assert np.linalg.norm(Heta1 - Heta0) <= 1e-9 * max(1.0, float(np.linalg.norm(Heta0)))
                seen.add(stop0)
        eta, _, _, _ = manopt_rtr.tCG(prob, Y, g, 1.0, 40)
        Y = Y + eta; Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    assert {1, 2}.issubset(seen) or {2, 5}.issubset(seen)  # the radius / curvature exits were exercised


def test_products_by_linearity_equal_direct_products():
    """C*tangent(r') = C*tangent(r) - alpha*C*Hmd needs Hmd tangent: it is (a projection minus a multiple of the tangent mdelta)."""
    C = _grid(10, 14, 5)
    n, p = C.shape[0], 8
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((n, p)); Y /= np.linalg.norm(Y, axis=1, keepdims=True)
    rowdot = lambda A, B: np.sum(A * B, axis=1, keepdims=True)
    tangent = lambda U: U - Y * rowdot(Y, U)
    eG = rowdot(C @ Y, Y)
    md = tangent(rng.standard_normal((n, p)))
    r = tangent(rng.standard_normal((n, p)))
    hmd = C @ md - Y * rowdot(Y, C @ md) - md * eG
    assert np.abs(rowdot(Y, hmd)).max() <= 1e-13 * np.abs(hmd).max()
    alpha, beta = 0.37, 1.9
    r2 = r - alpha * hmd
    md2 = tangent(r2 + beta * md)
    assert np.allclose(C @ tangent(r2), C @ r - alpha * (C @ hmd), atol=1e-12)
    assert np.allclose(C @ md2, (C @ r - alpha * (C @ hmd)) + beta * (C @ md), atol=1e-12)
