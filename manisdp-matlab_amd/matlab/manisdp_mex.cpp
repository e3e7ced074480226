// manisdp_mex.cpp -- thin MEX gateway over the C ABI of libmanisdp_hip.so (include/manisdp_hip.h).
//
// Follows the only native-call convention the reference has (src/C-files/innerc.cpp:3-33: plain
// mexFunction entry, double data via mxGetPr, argument-count checks and errors through
// mexErrMsgIdAndTxt("<toolbox>:<fn>:<what>", ...)).  One gateway, string-dispatched:
//
//   h = manisdp_mex('create_onlyunitdiag', C)            C sparse or dense n x n double
//   h = manisdp_mex('create_unitdiag',  At, b, c, n)     At sparse n^2 x m; b, c sparse or dense
//   h = manisdp_mex('create_unittrace', At, b, c, n)
//   h = manisdp_mex('create_generic',   At, b, c, n)
//   h = manisdp_mex('create_multiblock', At, b, c, nset, nob)   block orders nset (vector), first nob blocks unit-diagonal;
//                                                        the factor is one p x sum(nset) matrix (blocks side by side, zero rows
//                                                        below a block's own width)
//   h = manisdp_mex('create_dual_unitdiag', At, dAAt, b, c, n, B, cf)   dual approach (ManiDSDP_unitdiag.m): At = A(:,K.f+1:end)'
//                                                        sparse n^2 x m, dAAt = diag(A*A'), c the PSD part of the cost,
//                                                        B = A(:,1:K.f) sparse m x nf (or []), cf its costs
//       manisdp_mex('dual_set_penalty', h, sigma, w)     before every rtr of a dual handle (w: nf x 1)
//   [by, cex, as2, Af, z] = manisdp_mex('dual_outer_step', h)   ManiDSDP_unitdiag.m:70-81 on the device
//   y = manisdp_mex('dual_get_y', h)
//       manisdp_mex('set_multipliers', h, y, sigma)
//       manisdp_mex('set_point', h, Y)                   Y in the reference layout of the handle's kind
//   Y = manisdp_mex('get_point', h)
//   G = manisdp_mex('factor_gram', h)                    p x p Gram matrix of the resident factor (computed on the GPU)
//       manisdp_mex('factor_rotate', h, Qt)              resident factor <- its rank cut; Qt = Q(:,1:r)' is r x p
//       manisdp_mex('factor_append', h, D, alpha, normalize)   resident factor <- [factor; alpha*D'] (D: n x k), columns renormalised
//   info = manisdp_mex('rtr', h, opts)                   trustregions() on the resident point; opts.maxiter/maxinner/tolgradnorm
//                                                        [/Delta_bar: M.typicaldist() of a product manifold]
//   v = manisdp_mex('linesearch_cost', h, U, alpha)      co(retr(Y + alpha*U)); alpha = 0 (U may be []) gives co(Y)
//       manisdp_mex('linesearch_accept', h)
//   z = manisdp_mex('get_z', h)                          onlyunitdiag: 1 x n
//   [lam, V, lmax, ok, lower] = manisdp_mex('escape_eigs', h, k, tol, maxit)   lower: Weyl bound of lambda_min
//   [obj, Ax] = manisdp_mex('al_primal', h)
//   z = manisdp_mex('al_dual', h, y)                     n x 1 (unitdiag), scalar (unittrace), [] (generic)
//   [lam, V, lmax, ok, lower] = manisdp_mex('escape_eigs_dual', h, k, tol, maxit)
//   S = manisdp_mex('get_dual_slack', h)
//   k = manisdp_mex('kind', h)
//       manisdp_mex('set_option', h, name, value)        run-time switch of the handle (msdp_set_option)
//       manisdp_mex('comm_init_ipc', h, nranks, rank, name)   this MATLAB worker is member `rank` of a group of processes (row sharding)
//       manisdp_mex('destroy', h)
//
// Handles travel as uint64 scalars and are remembered here together with (kind, n, m), so the factor layout is
// taken from the handle, never guessed from array shapes; handles still alive when MATLAB clears the MEX file are
// destroyed by the mexAtExit hook.  The library returns codes (no exceptions cross the C ABI); this shim turns a
// non-zero code into mexErrMsgIdAndTxt after releasing its temporaries.  MATLAB owns every input (read-only) and
// every output (mxCreate*).
// Build:  mex -R2018a manisdp_mex.cpp -I<repo>/include -L<repo>/manisdp-matlab_amd/lib -lmanisdp_hip
// (tests/mex_stub/ holds a stand-in mex.h that lets the repository's own tests compile and drive this file
//  without MATLAB; see tests/test_mex_shim.py).
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "mex.h"
#include "matrix.h"
#include "manisdp_hip.h"

namespace {

struct Meta { int kind; int64_t n; int64_t m; int64_t nf; };
std::map<uint64_t, Meta> g_live;
bool g_exit_hooked = false;

void destroy_all() {
    for (auto& kv : g_live) msdp_destroy((msdp_handle)(uintptr_t)kv.first);
    g_live.clear();
    msdp_release_cache();                      // the escape workspace the library parks between handles
}

void fail(const char* what, int rc) {
    mexErrMsgIdAndTxt("ManiSDP:hip:call", "%s failed (%d): %s", what, rc, msdp_last_error());
}

void need(bool ok, const char* usage) {
    if (!ok) mexErrMsgIdAndTxt("ManiSDP:hip:nrhs", "usage: %s", usage);
}

uint64_t handle_key(const mxArray* a) {
    if (!mxIsUint64(a) || mxGetNumberOfElements(a) != 1)
        mexErrMsgIdAndTxt("ManiSDP:hip:handle", "handle must be a uint64 scalar");
    const uint64_t key = *(const uint64_t*)mxGetData(a);
    if (!g_live.count(key)) mexErrMsgIdAndTxt("ManiSDP:hip:handle", "unknown or already destroyed handle");
    return key;
}

mxArray* wrap_handle(msdp_handle h, int kind, int64_t n, int64_t m, int64_t nf = 0) {
    const uint64_t key = (uint64_t)(uintptr_t)h;
    g_live[key] = Meta{kind, n, m, nf};
    if (!g_exit_hooked) { mexAtExit(destroy_all); g_exit_hooked = true; }
    mxArray* o = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)mxGetData(o) = key;
    return o;
}

// bqpmom.m:37-38,114-115 hand over sparse b and c, example_theta.m:9-12 dense ones
std::vector<double> as_dense(const mxArray* a, size_t len) {
    std::vector<double> v(len, 0.0);
    if (mxIsSparse(a)) {
        const mwIndex* ir = mxGetIr(a);
        const mwIndex* jc = mxGetJc(a);
        const double* pr = mxGetPr(a);
        const mwSize ncol = mxGetN(a), nrow = mxGetM(a);
        for (mwSize j = 0; j < ncol; ++j)
            for (mwIndex t = jc[j]; t < jc[j + 1]; ++t) {
                const size_t pos = (size_t)ir[t] + (size_t)j * nrow;
                if (pos < len) v[pos] = pr[t];
            }
    } else {
        const size_t have = mxGetNumberOfElements(a);
        memcpy(v.data(), mxGetPr(a), (have < len ? have : len) * sizeof(double));
    }
    return v;
}

double field_or(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = (s && mxIsStruct(s)) ? mxGetField(s, 0, name) : nullptr;
    return f ? mxGetScalar(f) : dflt;
}

bool factor_is_n_by_p(int kind) { return kind == MSDP_KIND_UNITTRACE || kind == MSDP_KIND_GENERIC; }

// p of a factor handed over in the layout of `kind`; checks the other dimension against n
int32_t width_of(const mxArray* Y, const Meta& me) {
    const mwSize r = mxGetM(Y), c = mxGetN(Y);
    if (mxIsSparse(Y)) mexErrMsgIdAndTxt("ManiSDP:hip:layout", "the factor must be a full double matrix");
    if (factor_is_n_by_p(me.kind)) {
        if ((int64_t)r != me.n) mexErrMsgIdAndTxt("ManiSDP:hip:layout", "expected an n x p factor with n = %d rows", (int)me.n);
        return (int32_t)c;
    }
    if ((int64_t)c != me.n) mexErrMsgIdAndTxt("ManiSDP:hip:layout", "expected a p x n factor with n = %d columns", (int)me.n);
    return (int32_t)r;
}

mxArray* new_factor(const Meta& me, int32_t p) {
    return factor_is_n_by_p(me.kind) ? mxCreateDoubleMatrix((mwSize)me.n, (mwSize)p, mxREAL)
                                     : mxCreateDoubleMatrix((mwSize)p, (mwSize)me.n, mxREAL);
}

typedef int (*escape_fn)(msdp_handle, int32_t, double, int32_t, double*, double*, double*, int32_t*);

void run_escape(escape_fn fn, const char* name, msdp_handle h, const Meta& me, int nlhs, mxArray* plhs[],
                int nrhs, const mxArray* prhs[]) {
    need(nrhs == 5, "[lam, V, lmax, ok] = manisdp_mex('escape_eigs[_dual]', h, k, tol, maxit)");
    const int32_t k = (int32_t)mxGetScalar(prhs[2]);
    if (k < 1) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "k must be >= 1");
    mxArray* lam = mxCreateDoubleMatrix((mwSize)k, 1, mxREAL);
    mxArray* V = mxCreateDoubleMatrix((mwSize)me.n, (mwSize)k, mxREAL);
    double lmax = 0.0;
    int32_t steps = 0;
    const int rc = fn(h, k, mxGetScalar(prhs[3]), (int32_t)mxGetScalar(prhs[4]), mxGetPr(lam), mxGetPr(V), &lmax, &steps);
    if (rc) { mxDestroyArray(lam); mxDestroyArray(V); fail(name, rc); }
    int32_t nvalid = 0, conv = 0;
    double res = 0.0;
    (void)msdp_escape_info(h, &nvalid, &conv, &res);
    plhs[0] = lam;
    if (nlhs > 1) plhs[1] = V; else mxDestroyArray(V);
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(lmax);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(conv ? 1.0 : 0.0);
    if (nlhs > 4) {
        double lower = 0.0;
        (void)msdp_escape_lower_bound(h, &lower);
        plhs[4] = mxCreateDoubleScalar(lower);
    }
}

}  // namespace

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("ManiSDP:hip:nrhs", "first argument must be a command string");
    char cmdbuf[64];
    if (mxGetString(prhs[0], cmdbuf, sizeof(cmdbuf))) mexErrMsgIdAndTxt("ManiSDP:hip:cmd", "command string too long");
    const std::string cmd(cmdbuf);

    if (cmd == "release_cache") {              // device memory the library keeps between handles (the parked escape workspace)
        msdp_release_cache();
        return;
    }
    // ---------------------------------------------------------------- construction
    if (cmd == "create_onlyunitdiag") {
        need(nrhs == 2, "h = manisdp_mex('create_onlyunitdiag', C)");
        const mxArray* C = prhs[1];
        const int64_t n = (int64_t)mxGetM(C);
        if ((int64_t)mxGetN(C) != n) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "C must be square");
        msdp_handle h = nullptr;
        // MATLAB sparse = compressed columns with 64-bit mwIndex: the library's own input format
        const int rc = mxIsSparse(C)
            ? msdp_create_onlyunitdiag_csc(n, (const int64_t*)mxGetJc(C), (const int64_t*)mxGetIr(C), mxGetPr(C), 32, &h)
            : msdp_create_onlyunitdiag_dense(n, mxGetPr(C), 32, &h);
        if (rc) fail("create_onlyunitdiag", rc);
        plhs[0] = wrap_handle(h, MSDP_KIND_ONLYUNITDIAG, n, n);
        return;
    }
    if (cmd == "create_unitdiag" || cmd == "create_unittrace" || cmd == "create_generic") {
        need(nrhs == 5, "h = manisdp_mex('create_unitdiag' | 'create_unittrace' | 'create_generic', At, b, c, n)");
        const mxArray* At = prhs[1];
        if (!mxIsSparse(At)) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "At must be sparse (n^2 x m)");
        const int64_t n = (int64_t)mxGetScalar(prhs[4]);
        const int64_t m = (int64_t)mxGetN(At);
        if ((int64_t)mxGetM(At) != n * n) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "At must have n^2 rows");
        const std::vector<double> b = as_dense(prhs[2], (size_t)m);
        const std::vector<double> c = as_dense(prhs[3], (size_t)n * n);
        const int kind = cmd == "create_unitdiag" ? MSDP_KIND_UNITDIAG
                         : (cmd == "create_unittrace" ? MSDP_KIND_UNITTRACE : MSDP_KIND_GENERIC);
        msdp_handle h = nullptr;
        const int rc = msdp_create_affine(kind, n, m, (const int64_t*)mxGetJc(At), (const int64_t*)mxGetIr(At), mxGetPr(At),
                                          b.data(), c.data(), 32, &h);
        if (rc) fail(cmd.c_str(), rc);
        plhs[0] = wrap_handle(h, kind, n, m);
        return;
    }

    if (cmd == "create_multiblock") {
        need(nrhs == 6, "h = manisdp_mex('create_multiblock', At, b, c, nset, nob)");
        const mxArray* At = prhs[1];
        if (!mxIsSparse(At)) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "At must be sparse (sum(n_i^2) x m)");
        const size_t nb = mxGetNumberOfElements(prhs[4]);
        if (nb < 1) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "nset must list at least one block");
        std::vector<int64_t> nset(nb);
        int64_t N = 0, E = 0;
        for (size_t i = 0; i < nb; ++i) { nset[i] = (int64_t)mxGetPr(prhs[4])[i]; N += nset[i]; E += nset[i] * nset[i]; }
        if ((int64_t)mxGetM(At) != E) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "At must have sum(n_i^2) rows");
        const int64_t m = (int64_t)mxGetN(At);
        const std::vector<double> b = as_dense(prhs[2], (size_t)m);
        const std::vector<double> c = as_dense(prhs[3], (size_t)E);
        msdp_handle h = nullptr;
        const int rc = msdp_create_multiblock((int32_t)nb, nset.data(), (int32_t)mxGetScalar(prhs[5]), m, (const int64_t*)mxGetJc(At),
                                              (const int64_t*)mxGetIr(At), mxGetPr(At), b.data(), c.data(), 32, &h);
        if (rc) fail("create_multiblock", rc);
        plhs[0] = wrap_handle(h, MSDP_KIND_MULTIBLOCK, N, m);
        return;
    }

    if (cmd == "create_dual_unitdiag") {
        need(nrhs == 8, "h = manisdp_mex('create_dual_unitdiag', At, dAAt, b, c, n, B, cf)");
        const mxArray* At = prhs[1];
        if (!mxIsSparse(At)) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "At must be sparse (n^2 x m)");
        const int64_t n = (int64_t)mxGetScalar(prhs[5]);
        const int64_t m = (int64_t)mxGetN(At);
        if ((int64_t)mxGetM(At) != n * n) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "At must have n^2 rows");
        if ((int64_t)mxGetNumberOfElements(prhs[2]) != m) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "dAAt must have m entries");
        const std::vector<double> dAAt = as_dense(prhs[2], (size_t)m);
        const std::vector<double> b = as_dense(prhs[3], (size_t)m);
        const std::vector<double> c = as_dense(prhs[4], (size_t)(n * n));
        const mxArray* B = prhs[6];
        const int64_t nf = mxIsEmpty(B) ? 0 : (int64_t)mxGetN(B);
        if (nf > 0 && (!mxIsSparse(B) || (int64_t)mxGetM(B) != m)) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "B must be sparse m x nf");
        const std::vector<double> cf = as_dense(prhs[7], (size_t)(nf > 0 ? nf : 1));
        msdp_handle h = nullptr;
        const int rc = msdp_create_dual_unitdiag(n, m, (const int64_t*)mxGetJc(At), (const int64_t*)mxGetIr(At), mxGetPr(At), dAAt.data(),
                                                 b.data(), c.data(), (int32_t)nf, nf ? (const int64_t*)mxGetJc(B) : nullptr,
                                                 nf ? (const int64_t*)mxGetIr(B) : nullptr, nf ? mxGetPr(B) : nullptr, cf.data(), 32, &h);
        if (rc) fail("create_dual_unitdiag", rc);
        plhs[0] = wrap_handle(h, MSDP_KIND_DUAL_UNITDIAG, n, m, nf);
        return;
    }

    // ---------------------------------------------------------------- everything else takes a handle
    need(nrhs >= 2, "manisdp_mex(command, h, ...)");
    const uint64_t key = handle_key(prhs[1]);
    const Meta me = g_live[key];
    msdp_handle h = (msdp_handle)(uintptr_t)key;

    if (cmd == "destroy") {
        g_live.erase(key);
        msdp_destroy(h);
    } else if (cmd == "kind") {
        int32_t k = 0;
        const int rc = msdp_get_kind(h, &k);
        if (rc) fail("get_kind", rc);
        plhs[0] = mxCreateDoubleScalar((double)k);
    } else if (cmd == "escape_method") {
        int32_t mth = 0;
        const int rc = msdp_escape_method(h, &mth);
        if (rc) fail("escape_method", rc);
        plhs[0] = mxCreateDoubleScalar((double)mth);
    } else if (cmd == "set_option") {
        need(nrhs == 4 && mxIsChar(prhs[2]), "manisdp_mex('set_option', h, name, value)");
        char name[64];
        if (mxGetString(prhs[2], name, sizeof(name))) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "option name too long");
        const int rc = msdp_set_option(h, name, (int32_t)mxGetScalar(prhs[3]));
        if (rc) fail("set_option", rc);
    } else if (cmd == "comm_init_ipc") {
        // row sharding over MATLAB workers (one per GPU, or several on one GPU): member `rank` of `nranks` processes; `name` is a POSIX
        // shared-memory name, the same on every worker (msdp_comm_init_ipc); call right after create, before any point is set
        need(nrhs == 5 && mxIsChar(prhs[4]), "manisdp_mex('comm_init_ipc', h, nranks, rank, name)");
        char name[128];
        if (mxGetString(prhs[4], name, sizeof(name))) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "group name too long");
        const int rc = msdp_comm_init_ipc(h, (int32_t)mxGetScalar(prhs[2]), (int32_t)mxGetScalar(prhs[3]), name);
        if (rc) fail("comm_init_ipc", rc);
    } else if (cmd == "set_multipliers") {
        need(nrhs == 4, "manisdp_mex('set_multipliers', h, y, sigma)");
        if ((int64_t)mxGetNumberOfElements(prhs[2]) != me.m) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "y must have m entries");
        const int rc = msdp_set_multipliers(h, mxGetPr(prhs[2]), mxGetScalar(prhs[3]));
        if (rc) fail("set_multipliers", rc);
    } else if (cmd == "set_point") {
        need(nrhs == 3, "manisdp_mex('set_point', h, Y)");
        const int32_t p = width_of(prhs[2], me);
        const int rc = msdp_set_point(h, p, mxGetPr(prhs[2]));
        if (rc) fail("set_point", rc);
    } else if (cmd == "get_point") {
        int32_t p = 0;
        int rc = msdp_get_p(h, &p);
        if (rc) fail("get_p", rc);
        mxArray* Y = new_factor(me, p);
        rc = msdp_get_point(h, mxGetPr(Y));
        if (rc) { mxDestroyArray(Y); fail("get_point", rc); }
        plhs[0] = Y;
    } else if (cmd == "factor_gram") {
        int32_t p = 0;
        int rc = msdp_get_p(h, &p);
        if (rc) fail("get_p", rc);
        mxArray* G = mxCreateDoubleMatrix((mwSize)p, (mwSize)p, mxREAL);
        rc = msdp_factor_gram(h, mxGetPr(G));
        if (rc) { mxDestroyArray(G); fail("factor_gram", rc); }
        plhs[0] = G;
    } else if (cmd == "factor_rotate") {
        need(nrhs == 3, "manisdp_mex('factor_rotate', h, Qt)   % Qt = Q(:,1:r)' (r x p)");
        int32_t p = 0;
        int rc = msdp_get_p(h, &p);
        if (rc) fail("get_p", rc);
        if (mxIsSparse(prhs[2]) || (int32_t)mxGetN(prhs[2]) != p) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "Qt must be a full r x p matrix (p = %d)", (int)p);
        // r x p column-major == p x r row-major, the layout msdp_factor_rotate reads
        rc = msdp_factor_rotate(h, (int32_t)mxGetM(prhs[2]), mxGetPr(prhs[2]));
        if (rc) fail("factor_rotate", rc);
    } else if (cmd == "factor_append") {
        need(nrhs == 5, "manisdp_mex('factor_append', h, D, alpha, normalize)");
        if (mxIsSparse(prhs[2]) || (int64_t)mxGetM(prhs[2]) != me.n) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "D must be a full n x k matrix");
        const int rc = msdp_factor_append(h, (int32_t)mxGetN(prhs[2]), mxGetPr(prhs[2]), mxGetScalar(prhs[3]), (int32_t)mxGetScalar(prhs[4]));
        if (rc) fail("factor_append", rc);
    } else if (cmd == "rtr") {
        need(nrhs == 3, "info = manisdp_mex('rtr', h, opts)");
        msdp_rtr_opts o;
        msdp_rtr_default_opts(&o);
        o.maxiter = (int32_t)field_or(prhs[2], "maxiter", o.maxiter);          // opts of ManiSDP_unitdiag.m:44-47
        o.maxinner = (int32_t)field_or(prhs[2], "maxinner", o.maxinner);
        o.tolgradnorm = field_or(prhs[2], "tolgradnorm", o.tolgradnorm);
        o.Delta_bar = field_or(prhs[2], "Delta_bar", o.Delta_bar);
        msdp_rtr_stats st;
        const int rc = msdp_rtr(h, &o, &st);
        if (rc) fail("rtr", rc);
        const char* names[] = {"gradnorm", "cost", "iters", "hessvecs", "accepted", "rejected", "seconds"};
        const double vals[] = {st.gradnorm, st.cost, (double)st.iters, (double)st.hessvecs, (double)st.accepted,
                               (double)st.rejected, st.seconds};
        plhs[0] = mxCreateStructMatrix(1, 1, 7, names);
        for (int i = 0; i < 7; ++i) mxSetField(plhs[0], 0, names[i], mxCreateDoubleScalar(vals[i]));
    } else if (cmd == "linesearch_cost") {
        need(nrhs == 4, "v = manisdp_mex('linesearch_cost', h, U, alpha)");
        const double alpha = mxGetScalar(prhs[3]);
        const double* U = nullptr;
        if (alpha != 0.0) {
            int32_t p = 0;
            (void)msdp_get_p(h, &p);
            if (width_of(prhs[2], me) != p) mexErrMsgIdAndTxt("ManiSDP:hip:layout", "U must have the width of the resident point");
            U = mxGetPr(prhs[2]);
        }
        double v = 0.0;
        const int rc = msdp_linesearch_cost(h, U, alpha, &v);
        if (rc) fail("linesearch_cost", rc);
        plhs[0] = mxCreateDoubleScalar(v);
    } else if (cmd == "linesearch_accept") {
        const int rc = msdp_linesearch_accept(h);
        if (rc) fail("linesearch_accept", rc);
    } else if (cmd == "get_z") {
        mxArray* z = mxCreateDoubleMatrix(1, (mwSize)me.n, mxREAL);
        const int rc = msdp_get_z(h, mxGetPr(z));
        if (rc) { mxDestroyArray(z); fail("get_z", rc); }
        plhs[0] = z;
    } else if (cmd == "escape_eigs") {
        run_escape(msdp_escape_eigs, "escape_eigs", h, me, nlhs, plhs, nrhs, prhs);
    } else if (cmd == "escape_eigs_dual") {
        run_escape(msdp_escape_eigs_dual, "escape_eigs_dual", h, me, nlhs, plhs, nrhs, prhs);
    } else if (cmd == "al_primal") {
        double obj = 0.0;
        mxArray* Ax = mxCreateDoubleMatrix((mwSize)me.m, 1, mxREAL);
        const int rc = msdp_al_primal(h, &obj, mxGetPr(Ax));
        if (rc) { mxDestroyArray(Ax); fail("al_primal", rc); }
        plhs[0] = mxCreateDoubleScalar(obj);
        if (nlhs > 1) plhs[1] = Ax; else mxDestroyArray(Ax);
    } else if (cmd == "al_dual") {
        need(nrhs == 3, "z = manisdp_mex('al_dual', h, y)");
        if ((int64_t)mxGetNumberOfElements(prhs[2]) != me.m) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "y must have m entries");
        const mwSize zlen = (me.kind == MSDP_KIND_UNITDIAG || me.kind == MSDP_KIND_MULTIBLOCK) ? (mwSize)me.n
                            : (me.kind == MSDP_KIND_UNITTRACE ? 1 : 0);
        mxArray* z = mxCreateDoubleMatrix(zlen, zlen ? 1 : 0, mxREAL);
        const int rc = msdp_al_dual(h, mxGetPr(prhs[2]), zlen ? mxGetPr(z) : nullptr);
        if (rc) { mxDestroyArray(z); fail("al_dual", rc); }
        plhs[0] = z;
    } else if (cmd == "dual_set_penalty") {
        need(nrhs == 4, "manisdp_mex('dual_set_penalty', h, sigma, w)");
        if ((int64_t)mxGetNumberOfElements(prhs[3]) != me.nf) mexErrMsgIdAndTxt("ManiSDP:hip:arg", "w must have K.f entries");
        const std::vector<double> w = as_dense(prhs[3], (size_t)(me.nf > 0 ? me.nf : 1));
        const int rc = msdp_dual_set_penalty(h, mxGetScalar(prhs[2]), w.data());
        if (rc) fail("dual_set_penalty", rc);
    } else if (cmd == "dual_outer_step") {
        need(nrhs == 2, "[by, cex, as2, Af, z] = manisdp_mex('dual_outer_step', h)");
        double scal[3] = {0.0, 0.0, 0.0};
        mxArray* Af = mxCreateDoubleMatrix((mwSize)me.nf, 1, mxREAL);
        mxArray* z = mxCreateDoubleMatrix(1, (mwSize)me.n, mxREAL);
        std::vector<double> afbuf((size_t)(me.nf > 0 ? me.nf : 1), 0.0);
        const int rc = msdp_dual_outer_step(h, scal, afbuf.data(), mxGetPr(z));
        if (rc) { mxDestroyArray(Af); mxDestroyArray(z); fail("dual_outer_step", rc); }
        if (me.nf > 0) memcpy(mxGetPr(Af), afbuf.data(), (size_t)me.nf * sizeof(double));
        plhs[0] = mxCreateDoubleScalar(scal[0]);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(scal[1]);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(scal[2]);
        if (nlhs > 3) plhs[3] = Af; else mxDestroyArray(Af);
        if (nlhs > 4) plhs[4] = z; else mxDestroyArray(z);
    } else if (cmd == "dual_get_y") {
        mxArray* y = mxCreateDoubleMatrix((mwSize)me.m, 1, mxREAL);
        const int rc = msdp_dual_get_y(h, mxGetPr(y));
        if (rc) { mxDestroyArray(y); fail("dual_get_y", rc); }
        plhs[0] = y;
    } else if (cmd == "get_dual_slack_block") {           // (h, first row (1-based), order): one diagonal block of S
        need(nrhs == 4, "S_i = manisdp_mex('get_dual_slack_block', h, first_row, order)");
        const mwSize nb = (mwSize)mxGetScalar(prhs[3]);
        mxArray* S = mxCreateDoubleMatrix(nb, nb, mxREAL);
        const int rc = msdp_get_dual_slack_block(h, (int64_t)mxGetScalar(prhs[2]) - 1, (int64_t)nb, mxGetPr(S));
        if (rc) { mxDestroyArray(S); fail("get_dual_slack_block", rc); }
        plhs[0] = S;
    } else if (cmd == "get_dual_slack") {
        mxArray* S = mxCreateDoubleMatrix((mwSize)me.n, (mwSize)me.n, mxREAL);
        const int rc = msdp_get_dual_slack(h, mxGetPr(S));
        if (rc) { mxDestroyArray(S); fail("get_dual_slack", rc); }
        plhs[0] = S;
    } else {
        mexErrMsgIdAndTxt("ManiSDP:hip:cmd", "unknown command '%s'", cmd.c_str());
    }
}
