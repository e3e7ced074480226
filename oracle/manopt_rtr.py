"""ORACLE (test infrastructure, not product code) -- CPU restatement of Manopt 7.0's
Riemannian trust-region solver exactly as ManiSDP drives it.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path never does.

Follows (paths relative to the reference tree):
  * manopt7.0/manopt/solvers/trustregions/trustregions.m:340-372 (option defaults),
    :387-416 (initialisation), :441-767 (TR loop)
  * manopt7.0/manopt/solvers/trustregions/tCG.m:95-292
  * manopt7.0/manopt/core/getCostGrad.m:44-93 (cost then grad at the same point),
    getHessian.m:40-58, getPrecon.m:65 (identity), stoppingcriterion.m:51-72
  * manopt7.0/manopt/tools/matrixlincomb.m:23-29

Branches ManiSDP never takes (useRand, preconditioner, hooks, statsfun, debug,
verbosity>0) are omitted.  Parity status: the reference cannot be executed here
(no MATLAB/Octave); this restatement is pinned by the SDPLIB / Gset known optimal
values shipped with the reference (see tests/test_oracle_known_answers.py).

A ``problem`` is any object with
  cost(x) -> float               (called first at every new point)
  grad(x) -> array               (called after cost(x) at the same point)
  hess(x, u) -> array
  M.inner(x,a,b), M.norm(x,a), M.tangent(x,u), M.retr(x,u), M.zerovec(x),
  M.typicaldist(), M.rand(rng)
and the optional hooks ``on_accept()`` / ``on_reject()`` used by the
"correct per-point state" variant.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

EPS = np.finfo(np.float64).eps


@dataclass
class RTRInfo:
    gradnorm: float = 0.0
    cost: float = 0.0
    iters: int = 0
    hessvecs: int = 0
    accepted: int = 0
    rejected: int = 0
    cost_evals: int = 0
    stop_inner: list = field(default_factory=list)
    numinner: list = field(default_factory=list)
    Delta: float = 0.0


def tCG(problem, x, grad, Delta, maxinner, kappa=0.1, theta=1.0, mininner=1):
    """Steihaug-Toint truncated CG, tCG.m:95-292 with useRand=false and no
    preconditioner.  Works with ``mdelta = -(search direction)`` like the reference.
    Returns (eta, Heta, inner_it, stop_tCG)."""
    M = problem.M
    inner = lambda a, b: M.inner(x, a, b)
    eta = M.zerovec(x)
    Heta = M.zerovec(x)                      # tCG.m:103
    r = grad                                 # :104
    e_Pe = 0.0
    r_r = inner(r, r)                        # :114
    norm_r = math.sqrt(r_r)
    norm_r0 = norm_r
    z = r                                    # getPrecon.m:65 -> identity
    z_r = inner(z, r)                        # :126
    d_Pd = z_r
    mdelta = z                               # :130
    e_Pd = 0.0
    model_value = 0.0                        # :148
    stop_tCG = 5                             # :157
    j = 0
    for j in range(1, maxinner + 1):         # :160
        Hmdelta = problem.hess(x, mdelta)    # :163
        d_Hd = inner(mdelta, Hmdelta)        # :166
        # MATLAB yields +-Inf/NaN on division by zero; mirror that instead of raising
        alpha = z_r / d_Hd if d_Hd != 0.0 else math.copysign(math.inf, z_r) if z_r != 0 else math.nan
        e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd   # :173
        if d_Hd <= 0 or e_Pe_new >= Delta ** 2:                         # :183
            tau = (-e_Pd + math.sqrt(e_Pd * e_Pd + d_Pd * (Delta ** 2 - e_Pe))) / d_Pd  # :188
            eta = eta - tau * mdelta         # :192
            Heta = Heta - tau * Hmdelta      # :198
            stop_tCG = 1 if d_Hd <= 0 else 2
            break
        e_Pe = e_Pe_new                      # :214
        new_eta = eta - alpha * mdelta       # :215
        new_Heta = Heta - alpha * Hmdelta    # :220
        new_model_value = inner(new_eta, grad) + 0.5 * inner(new_eta, new_Heta)  # :227
        if new_model_value >= model_value:   # :228
            stop_tCG = 6
            break
        eta = new_eta
        Heta = new_Heta
        model_value = new_model_value        # :235
        r = r - alpha * Hmdelta              # :238
        r_r = inner(r, r)                    # :241
        norm_r = math.sqrt(r_r)
        if j >= mininner and norm_r <= norm_r0 * min(norm_r0 ** theta, kappa):  # :249
            stop_tCG = 3 if kappa < norm_r0 ** theta else 4
            break
        z = r                                # :261
        zold_rold = z_r
        z_r = inner(z, r)                    # :270
        beta = z_r / zold_rold               # :272
        mdelta = z + beta * mdelta           # :273
        mdelta = M.tangent(x, mdelta)        # :283
        e_Pd = beta * (e_Pd + alpha * d_Pd)  # :286
        d_Pd = z_r + beta * beta * d_Pd      # :287
    return eta, Heta, j, stop_tCG


def trustregions(problem, x, maxiter, maxinner, tolgradnorm, rng=None,
                 kappa=0.1, theta=1.0, rho_prime=0.1, rho_regularization=1e3,
                 mininner=1, Delta_bar=None, Delta0=None):
    """trustregions.m restated for the option set ManiSDP passes
    (ManiSDP_unitdiag.m:44-47): returns (x, fx, RTRInfo)."""
    M = problem.M
    if Delta_bar is None:
        Delta_bar = M.typicaldist()                 # trustregions.m:363-369
    if Delta0 is None:
        Delta0 = Delta_bar / 8.0                    # :370-372
    if x is None:
        x = M.rand(rng)                             # :390-392
    info = RTRInfo()
    fx = problem.cost(x)                            # :405 getCostGrad -> cost, then grad
    info.cost_evals += 1
    fgradx = problem.grad(x)
    norm_grad = M.norm(x, fgradx)                   # :406
    Delta = Delta0                                  # :409
    k = 0
    while True:                                     # :441
        if norm_grad < tolgradnorm:                 # stoppingcriterion.m:51-56 (strict <)
            break
        if k >= maxiter:                            # stoppingcriterion.m:67-72
            break
        eta, Heta, numit, stop_inner = tCG(problem, x, fgradx, Delta, maxinner,
                                           kappa=kappa, theta=theta, mininner=mininner)  # :495
        info.hessvecs += numit
        info.numinner.append(numit)
        info.stop_inner.append(stop_inner)
        x_prop = M.retr(x, eta)                     # :540
        fx_prop = problem.cost(x_prop)              # :544
        info.cost_evals += 1
        rhonum = fx - fx_prop                       # :548
        vecrho = fgradx + 0.5 * Heta                # :549
        rhoden = -M.inner(x, eta, vecrho)           # :550
        rho_reg = max(1.0, abs(fx)) * EPS * rho_regularization   # :579
        rhonum = rhonum + rho_reg
        rhoden = rhoden + rho_reg
        model_decreased = rhoden >= 0               # :614
        rho = rhonum / rhoden if rhoden != 0 else (math.nan if rhonum == 0 else math.copysign(math.inf, rhonum))
        if rho < 0.25 or (not model_decreased) or math.isnan(rho):      # :653
            Delta = Delta / 4.0
        elif rho > 0.75 and stop_inner in (1, 2):                        # :669
            Delta = min(2.0 * Delta, Delta_bar)
        if model_decreased and rho > rho_prime:                          # :688
            x = x_prop
            fx = fx_prop
            if hasattr(problem, "on_accept"):
                problem.on_accept()
            fgradx = problem.grad(x)                # :709 (cost is cached for x_prop)
            norm_grad = M.norm(x, fgradx)
            info.accepted += 1
        else:
            if hasattr(problem, "on_reject"):
                problem.on_reject()
            info.rejected += 1
        k += 1                                      # :729
    info.gradnorm = norm_grad
    info.cost = fx
    info.iters = k
    info.Delta = Delta
    return x, fx, info
