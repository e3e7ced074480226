"""The MATLAB-side binding (manisdp-matlab_amd/matlab/): the MEX gateway compiled against the stand-in mex.h of
tests/mex_stub/ and driven, command by command, the way msdp_al_engine.m drives it.

CPU: the gateway compiles, its argument / handle / command errors surface as mexErrMsgIdAndTxt with the documented
ids (convention of the reference's own MEX files, src/C-files/innerc.cpp:5-10), and the option defaults written in
the .m entry points are the reference's (ManiSDP_onlyunitdiag.m:8-17, ManiSDP_unitdiag.m:10-26,
ManiSDP_unittrace.m:10-25).
GPU: the same command sequences run against the real library and must give bit-identical results to the ctypes
binding on identical inputs (same library, deterministic kernels)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_path

STUB = os.path.join(ROOT, "tests", "mex_stub")
EXE = os.path.join(STUB, "_build", "mex_selftest")
MATLAB = os.path.join(ROOT, "manisdp-matlab_amd", "matlab")


def _build():
    subprocess.run(["make", "-C", STUB], check=True, capture_output=True)
    assert os.path.exists(EXE)


def test_gateway_compiles_and_reports_errors():
    _build()
    out = subprocess.run([EXE, "errors"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "errors: ok" in out.stdout
    for ident in ("ManiSDP:hip:nrhs", "ManiSDP:hip:handle", "ManiSDP:hip:arg"):
        assert ident in out.stdout


def _m_defaults(fname):
    src = open(os.path.join(MATLAB, fname)).read()
    body = re.search(r"defaults\s*=\s*\{(.*?)\};", src, re.S).group(1).replace("...", " ")
    out = {}
    for name, val in re.findall(r"'(\w+)'\s*,\s*([-+0-9.eE]+)", body):
        out[name] = float(val)
    return out


APPENDIX_A = {      # SURVEY.md appendix A (reference file:line in the module docstring)
    "ManiSDP_onlyunitdiag.m": dict(p0=2, AL_maxiter=20, tol=1e-8, theta=1e-1, delta=8, alpha=0.5, tolgradnorm=1e-8,
                                   TR_maxinner=100, TR_maxiter=40, line_search=0),
    "ManiSDP_unitdiag.m": dict(p0=2, AL_maxiter=300, gama=2, sigma0=1e-3, sigma_min=1e-2, sigma_max=1e7, tol=1e-8,
                               theta=1e-3, delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20, TR_maxiter=4,
                               tau1=1, tau2=1, line_search=0),
    "ManiSDP_multiblock.m": dict(min_facsize=2, AL_maxiter=1000, gama=2, sigma0=1e-1, sigma_min=1e-2, sigma_max=1e7,
                                 tol=1e-8, theta=1e-2, delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20,
                                 TR_maxiter=4, tau1=1e1, tau2=1e1, line_search=0),    # ManiSDP_multiblock.m:10-27
    "ManiDSDP_unitdiag.m": dict(ADMM_maxiter=300, gama=2, sigma0=1e-3, sigma_min=1e-3, sigma_max=1e7, tol=1e-8, theta=1e-3,
                                delta=8, alpha=0.1, tolgradnorm=1e-8, TR_maxinner=20, TR_maxiter=4, tau1=1e1, tau2=1e2,
                                line_search=0),      # src/dual/ManiDSDP_unitdiag.m:12-26 (p0 = ceil(log(m)), :11)
    "ManiSDP_unittrace.m": dict(p0=1, AL_maxiter=1000, gama=2, sigma0=1e1, sigma_min=1e2, sigma_max=1e7, tol=1e-8,
                                theta=1e-2, delta=8, alpha=0.05, tolgradnorm=1e-8, TR_maxinner=40, TR_maxiter=3,
                                tau1=1e-5, tau2=1e-4, line_search=1),
}


@pytest.mark.parametrize("fname", sorted(APPENDIX_A))
def test_m_entry_point_defaults(fname):
    got = _m_defaults(fname)
    assert got == {k: float(v) for k, v in APPENDIX_A[fname].items()}


def test_m_engine_keeps_protocol_and_fields():
    src = open(os.path.join(MATLAB, "msdp_al_engine.m")).read()
    # printed protocol (ManiSDP_unitdiag.m:28-29,75-76,79,85,126,129; ManiSDP_onlyunitdiag.m:55-56)
    for line in ("ManiSDP is starting...", "SDP size: n = %i, m = %i", "Optimality is reached!", "Slow progress!",
                 "Iteration maximum is reached!", "ManiSDP: optimum = %0.8f, time = %0.2fs",
                 "Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs",
                 "Iter %d, obj:%0.8f, dinf:%0.1e, r:%d, p:%d, time:%0.2fs"):
        assert line in src, line
    # data fields (ManiSDP_onlyunitdiag.m:86-95, ManiSDP_unitdiag.m:114-127, ManiSDP_unittrace.m:119-131)
    for field in ("X", "S", "z", "dinf", "gradnorm", "time", "status", "y", "gap", "pinf", "fac_size"):
        assert re.search(r"data\.%s\b" % field, src), field
    # no reference-style n x n host work inside the loop: everything goes through the gateway
    code = "\n".join(ln.split("%", 1)[0] for ln in src.splitlines())
    for banned in ("eig(full(S", "svd(", "trustregions(", "Y'*Y;\n    x =", "reshape("):
        assert banned not in code, banned


# ----------------------------------------------------------------------------------------------------- GPU
def _run(mode, blob, tmp_path):
    _build()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(blob)
    out = subprocess.run([EXE, mode, str(fin), str(fout)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    meta = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    return meta, np.fromfile(fout, dtype=np.float64), out.stdout


def _i64(*v):
    return np.asarray(v, dtype=np.int64).tobytes()


@pytest.mark.gpu
def test_gateway_onlyunitdiag_matches_ctypes(tmp_path):
    from manisdp_matlab_amd import _lib, problems
    C = problems.toroidal_grid_maxcut(20, 30, seed=5).tocsc()
    C.sort_indices()
    n, p, k = C.shape[0], 7, 3
    rng = np.random.default_rng(1)
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    blob = (_i64(n, C.nnz, p, 30, 60, k) + C.indptr.astype(np.int64).tobytes() + C.indices.astype(np.int64).tobytes()
            + C.data.astype(np.float64).tobytes() + np.ascontiguousarray(Y0).tobytes())
    meta, arr, stdout = _run("onlyunitdiag", blob, tmp_path)
    assert meta["kind"] == _lib.KIND_ONLYUNITDIAG and (meta["rows"], meta["cols"]) == (p, n)     # p x n at the boundary
    assert "use after destroy" in stdout and "ManiSDP:hip:handle" in stdout
    h = _lib.Handle.onlyunitdiag(C)
    h.set_point(Y0)
    st = h.rtr(_lib.default_opts(maxiter=30, maxinner=60, tolgradnorm=1e-8))
    Y = h.get_point(); z = h.get_z()
    lam, V, lmax, _ = h.escape_eigs(k, tol=1e-10, maxit=2000)
    _, conv, _ = h.escape_info()
    G = h.factor_gram()
    Q = np.eye(p)[:, :p - 2]
    h.factor_rotate(Q)
    h.factor_append(V[:, :1], 0.5, normalize=True)
    Y2 = h.get_point()
    assert Y2.shape == (n, p - 1)
    h.close()
    o = 0
    for ref in (Y.ravel(), z, lam, V.ravel(order="F"), G.ravel(), Y2.ravel()):
        got = arr[o:o + ref.size]; o += ref.size
        assert np.array_equal(got, ref)
    assert o == arr.size
    assert meta["cost"] == st.cost and meta["gradnorm"] == st.gradnorm and meta["hessvecs"] == st.hessvecs
    assert meta["lmax"] == lmax and meta["ok"] == int(conv)


@pytest.mark.gpu
@pytest.mark.parametrize("kind_name,sparse_bc", [("unitdiag", 1), ("unittrace", 0), ("generic", 0)])
def test_gateway_affine_matches_ctypes(tmp_path, kind_name, sparse_bc):
    from manisdp_matlab_amd import _lib, problems
    if kind_name == "unitdiag":
        kind = _lib.KIND_UNITDIAG
        Q = np.loadtxt(golden_path("bqp_Q_10_1.txt.gz"), delimiter=",")
        e = np.loadtxt(golden_path("bqp_e_10_1.txt.gz"), delimiter=",")
        At, b, c, K = problems.bqpmom(10, Q, e)
    else:
        kind = _lib.KIND_UNITTRACE if kind_name == "unittrace" else _lib.KIND_GENERIC
        At, b, c, K = problems.from_sdpa(golden_path("theta1.dat-s.gz"))
    import scipy.sparse as sp
    c = np.asarray(c.todense()).ravel() if sp.issparse(c) else np.asarray(c, float).ravel()
    b = np.asarray(b.todense()).ravel() if sp.issparse(b) else np.asarray(b, float).ravel()
    At = sp.csc_matrix(At); At.sort_indices()
    n, m, p, k = K["s"], b.size, 5, 4
    rng = np.random.default_rng(2)
    Y0 = rng.standard_normal((n, p))
    if kind == _lib.KIND_UNITDIAG:
        Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    elif kind == _lib.KIND_UNITTRACE:
        Y0 /= np.linalg.norm(Y0)
    U = 0.3 * rng.standard_normal((n, p))
    y = 0.1 * rng.standard_normal(m)
    sigma, alpha = 0.7, 0.5
    lay = (lambda M: np.ascontiguousarray(M)) if kind == _lib.KIND_UNITDIAG else (lambda M: np.asfortranarray(M))
    blob = (_i64(kind, n, m, At.nnz, p, 3, 15, k, sparse_bc) + np.asarray([sigma, alpha]).tobytes()
            + At.indptr.astype(np.int64).tobytes() + At.indices.astype(np.int64).tobytes() + At.data.astype(np.float64).tobytes()
            + b.tobytes() + c.tobytes() + y.tobytes() + lay(Y0).tobytes(order="A") + lay(U).tobytes(order="A"))
    meta, arr, stdout = _run("affine", blob, tmp_path)
    want_shape = (p, n) if kind == _lib.KIND_UNITDIAG else (n, p)
    assert (meta["rows"], meta["cols"]) == want_shape
    assert "factor in the wrong layout" in stdout and "ManiSDP:hip:layout" in stdout
    assert "use after the exit hook" in stdout
    h = _lib.Handle.affine(kind, At, b, c, n)
    h.set_multipliers(y, sigma)
    h.set_point(Y0)
    co0 = h.linesearch_cost(None, 0.0)
    co1 = h.linesearch_cost(U, alpha)
    st = h.rtr(_lib.default_opts(maxiter=3, maxinner=15, tolgradnorm=1e-8))
    Y = h.get_point()
    obj, Ax = h.al_primal(m)
    z = h.al_dual(y)
    lam, V, lmax, _ = h.escape_eigs_dual(k, tol=1e-10, maxit=4000)
    S = h.get_dual_slack()
    h.close()
    zarr = np.zeros(0) if z is None else np.atleast_1d(np.asarray(z, float))
    assert meta["zlen"] == zarr.size
    o = 0
    Yb = Y.ravel() if kind == _lib.KIND_UNITDIAG else Y.ravel(order="F")
    Sb = S[2:n - 2, 2:n - 2]                             # get_dual_slack_block with first row 3 (1-based), order n - 4
    for ref in (Yb, Ax, zarr, lam, V.ravel(order="F"), S.ravel(), Sb.ravel()):
        got = arr[o:o + ref.size]; o += ref.size
        assert np.array_equal(got, ref)
    assert o == arr.size
    assert (meta["co0"], meta["co1"], meta["cost"], meta["obj"], meta["lmax"]) == (co0, co1, st.cost, obj, lmax)
    assert meta["hessvecs"] == st.hessvecs


@pytest.mark.gpu
def test_gateway_dual_matches_ctypes(tmp_path):
    """create_dual_unitdiag / dual_set_penalty / dual_outer_step / dual_get_y through the gateway, the way
    ManiDSDP_unitdiag.m drives them, against the ctypes binding."""
    import scipy.sparse as sp
    from manisdp_matlab_amd import _lib, problems
    d = 7
    rng = np.random.default_rng(3)
    Q = rng.standard_normal((d, d)); Q = (Q + Q.T) / 2
    e = rng.standard_normal(d)
    A, b, c, K, dAAt, _ = problems.bqpsos_dual_problem(Q, e, d)
    n, m, nf = K["s"], b.size, K["f"]
    Ac = sp.csc_matrix(A)
    Apsd, B = sp.csr_matrix(Ac[:, nf:]), sp.csc_matrix(Ac[:, :nf])
    At = sp.csc_matrix(Apsd.T); At.sort_indices(); B.sort_indices()
    p = 6
    Y0 = rng.standard_normal((n, p)); Y0 /= np.linalg.norm(Y0, axis=1, keepdims=True)
    U = 0.3 * rng.standard_normal((n, p))
    sigma, alpha, w0 = 0.05, 0.5, 0.1
    blob = (_i64(n, m, At.nnz, B.nnz, p, 3, 15) + np.asarray([sigma, alpha, w0, c[0]]).tobytes()
            + At.indptr.astype(np.int64).tobytes() + At.indices.astype(np.int64).tobytes() + At.data.astype(np.float64).tobytes()
            + B.indptr.astype(np.int64).tobytes() + B.indices.astype(np.int64).tobytes() + B.data.astype(np.float64).tobytes()
            + dAAt.astype(np.float64).tobytes() + b.tobytes() + c[nf:].tobytes()
            + np.ascontiguousarray(Y0).tobytes() + np.ascontiguousarray(U).tobytes())
    meta, arr, stdout = _run("dual", blob, tmp_path)
    assert meta["kind"] == _lib.KIND_DUAL_UNITDIAG and (meta["rows"], meta["cols"]) == (p, n)
    assert "rtr after dual_outer_step without dual_set_penalty" in stdout and "ManiSDP:hip:call" in stdout
    h = _lib.Handle.dual_unitdiag(Apsd, b, c[nf:], dAAt, B, c[:nf])
    h.dual_set_penalty(sigma, np.array([w0]))
    h.set_point(Y0)
    co0 = h.linesearch_cost(None, 0.0)
    co1 = h.linesearch_cost(U, alpha)
    st = h.rtr(_lib.default_opts(maxiter=3, maxinner=15, tolgradnorm=1e-8))
    Y = h.get_point()
    by, cex, as2, Af, z = h.dual_outer_step()
    y = h.dual_get_y()
    X = h.get_dual_slack()
    h.close()
    o = 0
    for ref in (Y.ravel(), Af, z, y, X.ravel()):
        got = arr[o:o + ref.size]; o += ref.size
        assert np.array_equal(got, ref)
    assert o == arr.size
    assert (meta["co0"], meta["co1"], meta["cost"], meta["by"], meta["cex"], meta["as2"]) == (co0, co1, st.cost, by, cex, as2)
    assert meta["hessvecs"] == st.hessvecs


def _m_code(line):
    """One line of MATLAB without its comment and with string literals emptied (a quote after an identifier, a closing
    bracket or another quote is the transpose operator)."""
    out, inq, i = "", False, 0
    while i < len(line):
        ch = line[i]
        if inq:
            if ch == "'":
                if i + 1 < len(line) and line[i + 1] == "'":
                    i += 2
                    continue
                inq = False
            i += 1
            continue
        if ch == "'":
            prev = line[i - 1] if i > 0 else " "
            if not (prev.isalnum() or prev in ")]}'._"):
                inq = True
                i += 1
                continue
        if ch == "%":
            break
        out += ch
        i += 1
    return out


def test_m_files_are_structurally_balanced():
    """MATLAB is not in the image: at least every statement of the shipped .m files closes its brackets and every
    function / for / while / if / switch / try has its end."""
    import glob
    for fn in sorted(glob.glob(os.path.join(MATLAB, "*.m"))):
        lines = [_m_code(l) for l in open(fn).read().splitlines()]
        stmt, start = "", 0
        for k, c in enumerate(lines, 1):
            if not stmt:
                start = k
            if c.rstrip().endswith("..."):
                stmt += c.rstrip()[:-3]
                continue
            stmt += c
            for o, cl in ("()", "[]", "{}"):
                assert stmt.count(o) == stmt.count(cl), (os.path.basename(fn), start, stmt.strip()[:120])
            stmt = ""
        text = "\n".join(lines).replace("...\n", " ")
        prev = None
        while prev != text:                      # `end` inside an index expression is not a block end
            prev = text
            text = re.sub(r"\([^()\[\]{}]*\)|\[[^()\[\]{}]*\]|\{[^()\[\]{}]*\}", "", text)
        depth = 0
        for tok in re.findall(r"\b(function|for|while|if|switch|try|end)\b", text):
            depth += -1 if tok == "end" else 1
            assert depth >= 0, os.path.basename(fn)
        assert depth == 0, os.path.basename(fn)
