// manisdp_mex.cpp -- thin MEX gateway over the C ABI of libmanisdp_hip.so (include/manisdp_hip.h).
//
// Follows the only native-call convention the reference has (src/C-files/<fn>.cpp: plain
// mexFunction entry, double data via mxGetPr, errors via mexErrMsgIdAndTxt("MyToolbox:<fn>:...")).
// One gateway, string-dispatched:
//
//   h   = manisdp_mex('create_onlyunitdiag', C)              % C sparse or dense n x n double
//   h   = manisdp_mex('create_unitdiag',  At, b, c, n)       % At sparse n^2 x m, b/c sparse or dense
//   h   = manisdp_mex('create_unittrace', At, b, c, n)
//   h   = manisdp_mex('create_generic', At, b, c, n)          (ManiSDP.m, Euclidean manifold; Y is n x p)
//         manisdp_mex('set_multipliers', h, y, sigma)
//   [Y, info] = manisdp_mex('rtr', h, Y, opts)               % opts: struct with maxiter,maxinner,tolgradnorm
//   val = manisdp_mex('linesearch_cost', h, Y, U, alpha)     % co(retr(Y + alpha U))
//   z   = manisdp_mex('get_z', h)
//   [lam, V, lmax] = manisdp_mex('escape_eigs', h, k, tol, maxit)
//   [obj, Ax]      = manisdp_mex('al_primal', h, m)          % affine kinds: c'x and A*x at the resident point
//   z              = manisdp_mex('al_dual', h, y)            % builds S = eS - diag(z) | eS - z*I | eS on the device
//   [lam, V, lmax] = manisdp_mex('escape_eigs_dual', h, k, tol, maxit)
//         manisdp_mex('destroy', h)
//
// Handles travel as uint64 scalars.  The library returns codes (no exceptions cross the C ABI);
// this shim turns a non-zero code into mexErrMsgIdAndTxt after releasing its temporaries.
// MATLAB owns every input (read-only) and every output (mxCreate*).  Not compiled in this
// repository's CI (no MATLAB / mex.h in the build image): build with
//   mex -R2018a manisdp_mex.cpp -I../../include -L../lib -lmanisdp_hip
#include <cstring>
#include <string>
#include <vector>
#include "mex.h"
#include "matrix.h"
#include "manisdp_hip.h"

static void fail(const char* what, int rc) {
    mexErrMsgIdAndTxt("ManiSDP:hip", "%s failed (%d): %s", what, rc, msdp_last_error());
}
static msdp_handle get_handle(const mxArray* a) {
    if (!mxIsUint64(a) || mxGetNumberOfElements(a) != 1) mexErrMsgIdAndTxt("ManiSDP:hip:handle", "handle must be a uint64 scalar");
    return (msdp_handle)(uintptr_t)(*(uint64_t*)mxGetData(a));
}
static mxArray* put_handle(msdp_handle h) {
    mxArray* o = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)mxGetData(o) = (uint64_t)(uintptr_t)h;
    return o;
}
static std::vector<double> densify(const mxArray* a, size_t len) {
    std::vector<double> v(len, 0.0);
    if (mxIsSparse(a)) {
        const mwIndex* ir = mxGetIr(a); const mwIndex* jc = mxGetJc(a); const double* pr = mxGetPr(a);
        const mwSize ncol = mxGetN(a), nrow = mxGetM(a);
        for (mwSize j = 0; j < ncol; ++j) for (mwIndex t = jc[j]; t < jc[j + 1]; ++t) v[ir[t] + j * nrow] = pr[t];
    } else {
        memcpy(v.data(), mxGetPr(a), len * sizeof(double));
    }
    return v;
}
static double opt_field(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, name) : nullptr;
    return f ? mxGetScalar(f) : dflt;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("ManiSDP:hip:nrhs", "first argument must be a command string");
    char cmdbuf[64];
    mxGetString(prhs[0], cmdbuf, sizeof(cmdbuf));
    const std::string cmd(cmdbuf);
    if (cmd == "create_onlyunitdiag") {
        if (nrhs != 2) mexErrMsgIdAndTxt("ManiSDP:hip:nrhs", "create_onlyunitdiag(C)");
        const mxArray* C = prhs[1];
        const int64_t n = (int64_t)mxGetM(C);
        msdp_handle h = nullptr;
        int rc;
        if (mxIsSparse(C)) {
            // MATLAB sparse = CSC with 64-bit mwIndex: exactly the library's input format
            rc = msdp_create_onlyunitdiag_csc(n, (const int64_t*)mxGetJc(C), (const int64_t*)mxGetIr(C), mxGetPr(C), 32, &h);
        } else {
            rc = msdp_create_onlyunitdiag_dense(n, mxGetPr(C), 32, &h);
        }
        if (rc) fail("create_onlyunitdiag", rc);
        plhs[0] = put_handle(h);
    } else if (cmd == "create_unitdiag" || cmd == "create_unittrace" || cmd == "create_generic") {
        if (nrhs != 5) mexErrMsgIdAndTxt("ManiSDP:hip:nrhs", "%s(At, b, c, n)", cmd.c_str());
        const mxArray* At = prhs[1];
        if (!mxIsSparse(At)) mexErrMsgIdAndTxt("ManiSDP:hip:At", "At must be sparse (n^2 x m)");
        const int64_t n = (int64_t)mxGetScalar(prhs[4]);
        const int64_t m = (int64_t)mxGetN(At);
        std::vector<double> b = densify(prhs[2], (size_t)m);          // bqpmom.m:37 gives a sparse b
        std::vector<double> c = densify(prhs[3], (size_t)n * n);      // bqpmom.m:115 gives a sparse c
        msdp_handle h = nullptr;
        const int kind = cmd == "create_unitdiag" ? MSDP_KIND_UNITDIAG
                         : (cmd == "create_generic" ? MSDP_KIND_GENERIC : MSDP_KIND_UNITTRACE);
        int rc = msdp_create_affine(kind, n, m, (const int64_t*)mxGetJc(At), (const int64_t*)mxGetIr(At), mxGetPr(At),
                                    b.data(), c.data(), 32, &h);
        if (rc) fail(cmd.c_str(), rc);
        plhs[0] = put_handle(h);
    } else if (cmd == "set_multipliers") {
        msdp_handle h = get_handle(prhs[1]);
        int rc = msdp_set_multipliers(h, mxGetPr(prhs[2]), mxGetScalar(prhs[3]));
        if (rc) fail("set_multipliers", rc);
    } else if (cmd == "rtr") {
        // [Y, info] = rtr(h, Y, opts): Y in the reference layout (p x n for the oblique kinds, n x p for unittrace)
        msdp_handle h = get_handle(prhs[1]);
        const mxArray* Y = prhs[2];
        int32_t p_old = 0;
        (void)msdp_get_p(h, &p_old);
        msdp_rtr_opts o;
        msdp_rtr_default_opts(&o);
        o.maxiter = (int32_t)opt_field(prhs[3], "maxiter", o.maxiter);
        o.maxinner = (int32_t)opt_field(prhs[3], "maxinner", o.maxinner);
        o.tolgradnorm = opt_field(prhs[3], "tolgradnorm", o.tolgradnorm);
        const bool colmajor_np = opt_field(prhs[3], "unittrace", 0.0) != 0.0;
        const int32_t p = (int32_t)(colmajor_np ? mxGetN(Y) : mxGetM(Y));
        plhs[0] = mxDuplicateArray(Y);
        msdp_rtr_stats st;
        int rc = msdp_rtr_host(h, p, mxGetPr(plhs[0]), &o, &st);
        if (rc) { mxDestroyArray(plhs[0]); fail("rtr", rc); }
        if (nlhs > 1) {
            const char* fields[] = {"gradnorm", "cost", "iters", "hessvecs", "accepted", "rejected", "seconds"};
            plhs[1] = mxCreateStructMatrix(1, 1, 7, fields);
            mxSetField(plhs[1], 0, "gradnorm", mxCreateDoubleScalar(st.gradnorm));
            mxSetField(plhs[1], 0, "cost", mxCreateDoubleScalar(st.cost));
            mxSetField(plhs[1], 0, "iters", mxCreateDoubleScalar(st.iters));
            mxSetField(plhs[1], 0, "hessvecs", mxCreateDoubleScalar(st.hessvecs));
            mxSetField(plhs[1], 0, "accepted", mxCreateDoubleScalar(st.accepted));
            mxSetField(plhs[1], 0, "rejected", mxCreateDoubleScalar(st.rejected));
            mxSetField(plhs[1], 0, "seconds", mxCreateDoubleScalar(st.seconds));
        }
    } else if (cmd == "linesearch_cost") {
        msdp_handle h = get_handle(prhs[1]);
        int32_t p = 0;
        const mxArray* Y = prhs[2];
        const bool np_layout = mxGetM(Y) > mxGetN(Y);      // n x p (unittrace) vs p x n
        p = (int32_t)(np_layout ? mxGetN(Y) : mxGetM(Y));
        int rc = msdp_set_point(h, p, mxGetPr(Y));
        if (rc) fail("set_point", rc);
        double v = 0.0;
        rc = msdp_linesearch_cost(h, mxGetPr(prhs[3]), mxGetScalar(prhs[4]), &v);
        if (rc) fail("linesearch_cost", rc);
        plhs[0] = mxCreateDoubleScalar(v);
    } else if (cmd == "get_z") {
        msdp_handle h = get_handle(prhs[1]);
        int64_t r0 = 0, r1 = 0;
        (void)msdp_local_rows(h, &r0, &r1);
        plhs[0] = mxCreateDoubleMatrix(1, (mwSize)r1, mxREAL);
        int rc = msdp_get_z(h, mxGetPr(plhs[0]));
        if (rc) fail("get_z", rc);
    } else if (cmd == "escape_eigs") {
        msdp_handle h = get_handle(prhs[1]);
        const int32_t k = (int32_t)mxGetScalar(prhs[2]);
        int64_t r0 = 0, r1 = 0;
        (void)msdp_local_rows(h, &r0, &r1);
        plhs[0] = mxCreateDoubleMatrix(k, 1, mxREAL);
        mxArray* V = mxCreateDoubleMatrix((mwSize)r1, k, mxREAL);
        double lmax = 0.0;
        int32_t its = 0;
        int rc = msdp_escape_eigs(h, k, mxGetScalar(prhs[3]), (int32_t)mxGetScalar(prhs[4]), mxGetPr(plhs[0]), mxGetPr(V), &lmax, &its);
        if (rc) { mxDestroyArray(V); fail("escape_eigs", rc); }
        if (nlhs > 1) plhs[1] = V; else mxDestroyArray(V);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(lmax);
    } else if (cmd == "al_primal") {
        msdp_handle h = get_handle(prhs[1]);
        const mwSize m = (mwSize)mxGetScalar(prhs[2]);
        double obj = 0.0;
        mxArray* Ax = mxCreateDoubleMatrix(m, 1, mxREAL);
        int rc = msdp_al_primal(h, &obj, mxGetPr(Ax));
        if (rc) { mxDestroyArray(Ax); fail("al_primal", rc); }
        plhs[0] = mxCreateDoubleScalar(obj);
        if (nlhs > 1) plhs[1] = Ax; else mxDestroyArray(Ax);
    } else if (cmd == "al_dual") {
        msdp_handle h = get_handle(prhs[1]);
        int64_t r0 = 0, r1 = 0;
        (void)msdp_local_rows(h, &r0, &r1);
        plhs[0] = mxCreateDoubleMatrix((mwSize)r1, 1, mxREAL);      // unit trace / generic: only z(1) is meaningful
        int rc = msdp_al_dual(h, mxGetPr(prhs[2]), mxGetPr(plhs[0]));
        if (rc) fail("al_dual", rc);
    } else if (cmd == "escape_eigs_dual") {
        msdp_handle h = get_handle(prhs[1]);
        const int32_t k = (int32_t)mxGetScalar(prhs[2]);
        int64_t r0 = 0, r1 = 0;
        (void)msdp_local_rows(h, &r0, &r1);
        plhs[0] = mxCreateDoubleMatrix(k, 1, mxREAL);
        mxArray* V = mxCreateDoubleMatrix((mwSize)r1, k, mxREAL);
        double lmax = 0.0;
        int32_t its = 0;
        int rc = msdp_escape_eigs_dual(h, k, mxGetScalar(prhs[3]), (int32_t)mxGetScalar(prhs[4]), mxGetPr(plhs[0]), mxGetPr(V), &lmax, &its);
        if (rc) { mxDestroyArray(V); fail("escape_eigs_dual", rc); }
        if (nlhs > 1) plhs[1] = V; else mxDestroyArray(V);
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(lmax);
    } else if (cmd == "destroy") {
        msdp_destroy(get_handle(prhs[1]));
    } else {
        mexErrMsgIdAndTxt("ManiSDP:hip:cmd", "unknown command '%s'", cmd.c_str());
    }
}
