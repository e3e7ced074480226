"""a14: the option defaults and `data` fields of the host-side mirrors are the
reference's -- ManiSDP_onlyunitdiag.m:8-17,86-95; ManiSDP_unitdiag.m:10-26,114-127; ManiSDP_unittrace.m:10-25,119-131;
ManiSDP.m:9-25.  Pure CPU: importing the mirrors needs no GPU."""
import inspect

import pytest

APPENDIX_A = {
    #               onlyunitdiag  unitdiag  unittrace
    "p0":          (2,            2,        1),
    "AL_maxiter":  (20,           300,      1000),
    "gama":        (None,         2,        2),
    "sigma0":      (None,         1e-3,     1e1),
    "sigma_min":   (None,         1e-2,     1e2),
    "sigma_max":   (None,         1e7,      1e7),
    "tol":         (1e-8,         1e-8,     1e-8),
    "theta":       (1e-1,         1e-3,     1e-2),
    "delta":       (8,            8,        8),
    "alpha":       (0.5,          0.1,      0.05),
    "tolgradnorm": (1e-8,         1e-8,     1e-8),
    "TR_maxinner": (100,          20,       40),
    "TR_maxiter":  (40,           4,        3),
    "tau1":        (None,         1,        1e-5),
    "tau2":        (None,         1,        1e-4),
    "line_search": (0,            0,        1),
}
KINDS = ("onlyunitdiag", "unitdiag", "unittrace")


@pytest.mark.parametrize("col,kind", list(enumerate(KINDS)))
def test_defaults_are_appendix_a(col, kind):
    from manisdp_matlab_amd import solvers
    want = {k: v[col] for k, v in APPENDIX_A.items() if v[col] is not None}
    assert solvers.DEFAULTS[kind] == want


def test_generic_defaults():
    from manisdp_matlab_amd import solvers
    assert solvers.DEFAULTS["generic"] == dict(p0=1, AL_maxiter=1000, gama=2, sigma0=1e-2, sigma_min=1e-1, sigma_max=1e7,
                                               tol=1e-8, theta=1e-2, delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20,
                                               TR_maxiter=4, tau1=1e-2, tau2=1e-1, line_search=1, solver=0)   # ManiSDP.m:9-25


def test_data_fields_and_protocol_in_source():
    """Every field of the reference's `data` struct is produced, and the printed protocol lines are the reference's."""
    from manisdp_matlab_amd import solvers
    src = inspect.getsource(solvers)
    for kind, fields in solvers.DATA_FIELDS.items():
        for f in fields:
            assert '"%s"' % f in src, (kind, f)
    for line in ("ManiSDP is starting...", "SDP size: n = ", "Optimality is reached!", "Slow progress!",
                 "Iteration maximum is reached!", "ManiSDP: optimum = %0.8f, time = %0.2fs",
                 "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs",
                 "Iter %d, obj:%0.8f, dinf:%0.1e, r:%d, p:%d, time:%0.2fs"):
        assert line in src, line
    assert solvers.DATA_FIELDS["onlyunitdiag"] == ("X", "S", "z", "dinf", "gradnorm", "time", "status")
    assert set(solvers.DATA_FIELDS["unitdiag"]) - set(solvers.DATA_FIELDS["unittrace"]) == {"fac_size"}   # ManiSDP_unitdiag.m:123


def test_multiblock_defaults():
    from manisdp_matlab_amd import solvers
    assert solvers.DEFAULTS["multiblock"] == dict(min_facsize=2, AL_maxiter=1000, gama=2, sigma0=1e-1, sigma_min=1e-2,
                                                  sigma_max=1e7, tol=1e-8, theta=1e-2, delta=8, alpha=0.1,
                                                  tolgradnorm=1e-8, TR_maxinner=20, TR_maxiter=4, tau1=1e1, tau2=1e1,
                                                  line_search=0)                      # ManiSDP_multiblock.m:10-27


def test_dual_unitdiag_defaults():
    from manisdp_matlab_amd import solvers
    assert solvers.DEFAULTS["dual_unitdiag"] == dict(ADMM_maxiter=300, gama=2, sigma0=1e-3, sigma_min=1e-3, sigma_max=1e7,
                                                     tol=1e-8, theta=1e-3, delta=8, alpha=0.1, tolgradnorm=1e-8,
                                                     TR_maxinner=20, TR_maxiter=4, tau1=1e1, tau2=1e2,
                                                     line_search=0)                   # src/dual/ManiDSDP_unitdiag.m:12-26
    assert solvers.DATA_FIELDS["dual_unitdiag"] == ("X", "y", "S", "w", "gap", "pinf", "dinf", "gradnorm", "time", "fac_size",
                                                    "seta", "status")                 # :132-144
    src = inspect.getsource(solvers)
    assert "ManiDSDP: optimum = %0.8f, time = %0.2fs" in src and "math.ceil(math.log(m))" in src   # :148, :11
