// mex_selftest.cpp -- drives manisdp_mex.cpp's mexFunction the way the MATLAB drop-ins do, through the stand-in
// mex.h of this directory (test infrastructure; see tests/test_mex_shim.py).
//
//   mex_selftest errors                      argument / handle / command errors of the gateway (no GPU needed)
//   mex_selftest onlyunitdiag <in> <out>     create_onlyunitdiag -> set_point -> rtr -> get_point -> get_z -> escape_eigs
//   mex_selftest affine <in> <out>           create_* -> set_multipliers -> set_point -> linesearch_cost -> rtr -> get_point
//                                            -> al_primal -> al_dual -> escape_eigs_dual -> get_dual_slack
// <in> is a little-endian binary file written by the Python test, <out> receives the arrays (doubles, in the order
// they are produced); scalars are printed as one JSON object on stdout.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "mex.h"

namespace {

struct Reader {
    FILE* f;
    explicit Reader(const char* path) : f(fopen(path, "rb")) { if (!f) { perror(path); exit(2); } }
    ~Reader() { if (f) fclose(f); }
    int64_t i64() { int64_t v = 0; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "short read\n"); exit(2); } return v; }
    double f64() { double v = 0; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "short read\n"); exit(2); } return v; }
    void i64s(std::vector<mwIndex>& dst, size_t cnt) {
        dst.resize(cnt ? cnt : 1);
        for (size_t i = 0; i < cnt; ++i) dst[i] = (mwIndex)i64();
    }
    void f64s(std::vector<double>& dst, size_t cnt) {
        dst.resize(cnt ? cnt : 1);
        if (cnt && fread(dst.data(), 8, cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
    }
};

struct Writer {
    FILE* f;
    explicit Writer(const char* path) : f(fopen(path, "wb")) { if (!f) { perror(path); exit(2); } }
    ~Writer() { if (f) fclose(f); }
    void put(const mxArray* a) { const size_t cnt = mxGetNumberOfElements(a); if (cnt) fwrite(mxGetPr(a), 8, cnt, f); }
};

// call the gateway like MATLAB: outputs are owned by the caller afterwards
std::vector<mxArray*> call(int nlhs, std::vector<const mxArray*> in) {
    std::vector<mxArray*> out((size_t)(nlhs > 0 ? nlhs : 1), nullptr);
    mexFunction(nlhs, out.data(), (int)in.size(), in.data());
    return out;
}

mxArray* sparse_from(Reader& r, mwSize rows, mwSize cols, int64_t nnz) {
    mxArray* a = mxCreateSparse(rows, cols, (mwSize)nnz, mxREAL);
    r.i64s(a->jc, cols + 1);
    r.i64s(a->ir, (size_t)nnz);
    r.f64s(a->pr, (size_t)nnz);
    return a;
}

mxArray* full_from(Reader& r, mwSize rows, mwSize cols) {
    mxArray* a = mxCreateDoubleMatrix(rows, cols, mxREAL);
    r.f64s(a->pr, rows * cols);
    return a;
}

mxArray* rtr_opts(int maxiter, int maxinner, double tolgradnorm) {
    const char* names[] = {"maxiter", "maxinner", "tolgradnorm"};
    mxArray* s = mxCreateStructMatrix(1, 1, 3, names);
    mxSetField(s, 0, "maxiter", mxCreateDoubleScalar(maxiter));
    mxSetField(s, 0, "maxinner", mxCreateDoubleScalar(maxinner));
    mxSetField(s, 0, "tolgradnorm", mxCreateDoubleScalar(tolgradnorm));
    return s;
}

double field(const mxArray* s, const char* name) { return mxGetScalar(mxGetField(s, 0, name)); }

int expect_error(const char* what, const char* want_id, std::vector<const mxArray*> in) {
    try {
        call(1, in);
    } catch (const mex_stub_error& e) {
        if (e.id == want_id) { printf("  %-34s -> %s: %s\n", what, e.id.c_str(), e.what()); return 0; }
        printf("  %-34s -> WRONG id %s (wanted %s): %s\n", what, e.id.c_str(), want_id, e.what());
        return 1;
    }
    printf("  %-34s -> no error raised (wanted %s)\n", what, want_id);
    return 1;
}

int run_errors() {
    int bad = 0;
    mxArray* cmd_unknown = mxCreateString("no_such_command");
    mxArray* cmd_rtr = mxCreateString("rtr");
    mxArray* cmd_create = mxCreateString("create_onlyunitdiag");
    mxArray* not_handle = mxCreateDoubleScalar(3.0);
    mxArray* stale = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)mxGetData(stale) = 0xdeadbeefULL;
    mxArray* rect = mxCreateDoubleMatrix(3, 4, mxREAL);
    bad += expect_error("no arguments", "ManiSDP:hip:nrhs", {});
    bad += expect_error("command is not a string", "ManiSDP:hip:nrhs", {not_handle});
    bad += expect_error("command without a handle", "ManiSDP:hip:nrhs", {cmd_rtr});
    bad += expect_error("handle of the wrong class", "ManiSDP:hip:handle", {cmd_rtr, not_handle});
    bad += expect_error("unknown handle", "ManiSDP:hip:handle", {cmd_unknown, stale});
    bad += expect_error("create with a non-square C", "ManiSDP:hip:arg", {cmd_create, rect});
    bad += expect_error("create with too many arguments", "ManiSDP:hip:nrhs", {cmd_create, rect, rect});
    for (mxArray* a : {cmd_unknown, cmd_rtr, cmd_create, not_handle, stale, rect}) mxDestroyArray(a);
    printf(bad ? "errors: %d FAILED\n" : "errors: ok\n", bad);
    return bad ? 1 : 0;
}

int run_onlyunitdiag(const char* in, const char* outp) {
    Reader r(in);
    const int64_t n = r.i64(), nnz = r.i64(), p = r.i64(), maxiter = r.i64(), maxinner = r.i64(), k = r.i64();
    mxArray* C = sparse_from(r, (mwSize)n, (mwSize)n, nnz);
    mxArray* Y0 = full_from(r, (mwSize)p, (mwSize)n);               // p x n column-major (ManiSDP_onlyunitdiag.m:39)
    Writer w(outp);
    mxArray* c_create = mxCreateString("create_onlyunitdiag");
    mxArray* h = call(1, {c_create, C})[0];
    mxArray* c_set = mxCreateString("set_point");
    call(0, {c_set, h, Y0});
    mxArray* c_rtr = mxCreateString("rtr");
    mxArray* opts = rtr_opts((int)maxiter, (int)maxinner, 1e-8);
    mxArray* info = call(1, {c_rtr, h, opts})[0];
    mxArray* c_get = mxCreateString("get_point");
    mxArray* Y = call(1, {c_get, h})[0];
    mxArray* c_z = mxCreateString("get_z");
    mxArray* z = call(1, {c_z, h})[0];
    mxArray* c_esc = mxCreateString("escape_eigs");
    mxArray* kk = mxCreateDoubleScalar((double)k);
    mxArray* tol = mxCreateDoubleScalar(1e-10);
    mxArray* mit = mxCreateDoubleScalar(2000);
    std::vector<mxArray*> e = call(4, {c_esc, h, kk, tol, mit});
    mxArray* c_kind = mxCreateString("kind");
    mxArray* kind = call(1, {c_kind, h})[0];
    w.put(Y); w.put(z); w.put(e[0]); w.put(e[1]);
    // the resident-factor commands the engine uses between two rtr calls: Gram matrix, rank cut to r = p - 2 with the
    // leading columns of the identity (Qt = [I_r 0]), widening by the first escape vector
    mxArray* c_gram = mxCreateString("factor_gram");
    mxArray* G = call(1, {c_gram, h})[0];
    const mwSize rr = (mwSize)p - 2;
    mxArray* Qt = mxCreateDoubleMatrix(rr, (mwSize)p, mxREAL);
    for (mwSize i = 0; i < rr; ++i) mxGetPr(Qt)[i + i * rr] = 1.0;
    mxArray* c_rot = mxCreateString("factor_rotate");
    call(0, {c_rot, h, Qt});
    mxArray* c_app = mxCreateString("factor_append");
    mxArray* D = mxCreateDoubleMatrix((mwSize)n, 1, mxREAL);
    for (int64_t i = 0; i < n; ++i) mxGetPr(D)[i] = mxGetPr(e[1])[i];
    mxArray* alpha = mxCreateDoubleScalar(0.5);
    mxArray* one = mxCreateDoubleScalar(1.0);
    call(0, {c_app, h, D, alpha, one});
    mxArray* Y2 = call(1, {c_get, h})[0];
    w.put(G); w.put(Y2);
    for (mxArray* a : {c_gram, G, Qt, c_rot, c_app, D, alpha, one, Y2}) mxDestroyArray(a);
    printf("{\"kind\": %d, \"rows\": %zu, \"cols\": %zu, \"cost\": %.17g, \"gradnorm\": %.17g, \"hessvecs\": %d, \"iters\": %d, "
           "\"lmax\": %.17g, \"ok\": %d}\n",
           (int)mxGetScalar(kind), mxGetM(Y), mxGetN(Y), field(info, "cost"), field(info, "gradnorm"), (int)field(info, "hessvecs"),
           (int)field(info, "iters"), mxGetScalar(e[2]), (int)mxGetScalar(e[3]));
    mxArray* c_destroy = mxCreateString("destroy");
    call(0, {c_destroy, h});
    // a destroyed handle must be refused afterwards
    int bad = expect_error("use after destroy", "ManiSDP:hip:handle", {c_get, h});
    for (mxArray* a : {C, Y0, c_create, h, c_set, c_rtr, opts, info, c_get, Y, c_z, z, c_esc, kk, tol, mit, e[0], e[1], e[2], e[3],
                       c_kind, kind, c_destroy})
        mxDestroyArray(a);
    return bad;
}

int run_affine(const char* in, const char* outp) {
    Reader r(in);
    const int64_t kind = r.i64(), n = r.i64(), m = r.i64(), nnz = r.i64(), p = r.i64(), maxiter = r.i64(), maxinner = r.i64(),
                  k = r.i64(), sparse_bc = r.i64();
    const double sigma = r.f64(), alpha = r.f64();
    mxArray* At = sparse_from(r, (mwSize)(n * n), (mwSize)m, nnz);
    mxArray* b = full_from(r, (mwSize)m, 1);
    mxArray* c = full_from(r, (mwSize)(n * n), 1);
    if (sparse_bc) {
        // hand b and c over as sparse columns, as bqpmom.m:37-38,114-115 does
        for (mxArray** pa : {&b, &c}) {
            mxArray* d = *pa;
            const mwSize len = mxGetM(d);
            mwSize cnt = 0;
            for (mwSize i = 0; i < len; ++i) cnt += d->pr[i] != 0.0;
            mxArray* s = mxCreateSparse(len, 1, cnt, mxREAL);
            mwSize q = 0;
            for (mwSize i = 0; i < len; ++i) if (d->pr[i] != 0.0) { s->ir[q] = i; s->pr[q] = d->pr[i]; ++q; }
            s->jc[0] = 0; s->jc[1] = cnt;
            mxDestroyArray(d);
            *pa = s;
        }
    }
    mxArray* y = full_from(r, (mwSize)m, 1);
    const bool n_by_p = kind != 2;                                   // MSDP_KIND_UNITDIAG = 2: p x n
    mxArray* Y0 = n_by_p ? full_from(r, (mwSize)n, (mwSize)p) : full_from(r, (mwSize)p, (mwSize)n);
    mxArray* U = n_by_p ? full_from(r, (mwSize)n, (mwSize)p) : full_from(r, (mwSize)p, (mwSize)n);
    Writer w(outp);
    mxArray* c_create = mxCreateString(kind == 2 ? "create_unitdiag" : (kind == 3 ? "create_unittrace" : "create_generic"));
    mxArray* nn = mxCreateDoubleScalar((double)n);
    mxArray* h = call(1, {c_create, At, b, c, nn})[0];
    mxArray* c_mult = mxCreateString("set_multipliers");
    mxArray* sg = mxCreateDoubleScalar(sigma);
    call(0, {c_mult, h, y, sg});
    mxArray* c_set = mxCreateString("set_point");
    call(0, {c_set, h, Y0});
    mxArray* c_ls = mxCreateString("linesearch_cost");
    mxArray* zero = mxCreateDoubleScalar(0.0);
    mxArray* al = mxCreateDoubleScalar(alpha);
    mxArray* co0 = call(1, {c_ls, h, U, zero})[0];
    mxArray* co1 = call(1, {c_ls, h, U, al})[0];
    mxArray* c_rtr = mxCreateString("rtr");
    mxArray* opts = rtr_opts((int)maxiter, (int)maxinner, 1e-8);
    mxArray* info = call(1, {c_rtr, h, opts})[0];
    mxArray* c_get = mxCreateString("get_point");
    mxArray* Y = call(1, {c_get, h})[0];
    mxArray* c_prim = mxCreateString("al_primal");
    std::vector<mxArray*> pr = call(2, {c_prim, h});
    mxArray* c_dual = mxCreateString("al_dual");
    mxArray* z = call(1, {c_dual, h, y})[0];
    mxArray* c_esc = mxCreateString("escape_eigs_dual");
    mxArray* kk = mxCreateDoubleScalar((double)k);
    mxArray* tol = mxCreateDoubleScalar(1e-10);
    mxArray* mit = mxCreateDoubleScalar(4000);
    std::vector<mxArray*> e = call(4, {c_esc, h, kk, tol, mit});
    mxArray* c_S = mxCreateString("get_dual_slack");
    mxArray* S = call(1, {c_S, h})[0];
    // one diagonal block of S (rows 3..n-2, 1-based first row 3): what ManiSDP_multiblock.m reads per block
    mxArray* c_Sb = mxCreateString("get_dual_slack_block");
    mxArray* b_r0 = mxCreateDoubleScalar(3.0);
    mxArray* b_nb = mxCreateDoubleScalar((double)(n - 4));
    mxArray* Sb = call(1, {c_Sb, h, b_r0, b_nb})[0];
    w.put(Y); w.put(pr[1]); w.put(z); w.put(e[0]); w.put(e[1]); w.put(S); w.put(Sb);
    printf("{\"rows\": %zu, \"cols\": %zu, \"co0\": %.17g, \"co1\": %.17g, \"cost\": %.17g, \"gradnorm\": %.17g, \"hessvecs\": %d, "
           "\"obj\": %.17g, \"zlen\": %zu, \"lmax\": %.17g, \"ok\": %d}\n",
           mxGetM(Y), mxGetN(Y), mxGetScalar(co0), mxGetScalar(co1), field(info, "cost"), field(info, "gradnorm"),
           (int)field(info, "hessvecs"), mxGetScalar(pr[0]), mxGetNumberOfElements(z), mxGetScalar(e[2]), (int)mxGetScalar(e[3]));
    // wrong layout must be refused: hand the transposed shape to set_point (only meaningful when p != n)
    int bad = 0;
    if (p != n) {
        mxArray* wrong = n_by_p ? mxCreateDoubleMatrix((mwSize)p, (mwSize)n, mxREAL) : mxCreateDoubleMatrix((mwSize)n, (mwSize)p, mxREAL);
        bad += expect_error("factor in the wrong layout", "ManiSDP:hip:layout", {c_set, h, wrong});
        mxDestroyArray(wrong);
    }
    // leave the handle alive: the exit hook (mexAtExit) must destroy it
    if (mex_stub_exit_hook()) mex_stub_exit_hook()();
    else { printf("no exit hook registered\n"); bad += 1; }
    bad += expect_error("use after the exit hook", "ManiSDP:hip:handle", {c_get, h});
    for (mxArray* a : {At, b, c, y, Y0, U, c_create, nn, h, c_mult, sg, c_set, c_ls, zero, al, co0, co1, c_rtr, opts, info, c_get, Y,
                       c_prim, pr[0], pr[1], c_dual, z, c_esc, kk, tol, mit, e[0], e[1], e[2], e[3], c_S, S})
        mxDestroyArray(a);
    return bad;
}

int run_dual(const char* in, const char* outp) {
    Reader r(in);
    const int64_t n = r.i64(), m = r.i64(), nnz = r.i64(), nnzB = r.i64(), p = r.i64(), maxiter = r.i64(), maxinner = r.i64();
    const double sigma = r.f64(), alpha = r.f64(), w0 = r.f64(), cf0 = r.f64();
    mxArray* At = sparse_from(r, (mwSize)(n * n), (mwSize)m, nnz);
    mxArray* B = sparse_from(r, (mwSize)m, 1, nnzB);
    mxArray* dAAt = full_from(r, (mwSize)m, 1);
    mxArray* b = full_from(r, (mwSize)m, 1);
    mxArray* c = full_from(r, (mwSize)(n * n), 1);
    mxArray* Y0 = full_from(r, (mwSize)p, (mwSize)n);                // p x n like ManiDSDP_unitdiag.m:64
    mxArray* U = full_from(r, (mwSize)p, (mwSize)n);
    Writer w(outp);
    mxArray* c_create = mxCreateString("create_dual_unitdiag");
    mxArray* nn = mxCreateDoubleScalar((double)n);
    mxArray* cf = mxCreateDoubleScalar(cf0);
    mxArray* h = call(1, {c_create, At, dAAt, b, c, nn, B, cf})[0];
    mxArray* c_pen = mxCreateString("dual_set_penalty");
    mxArray* sg = mxCreateDoubleScalar(sigma);
    mxArray* wf = mxCreateDoubleScalar(w0);
    call(0, {c_pen, h, sg, wf});
    mxArray* c_set = mxCreateString("set_point");
    call(0, {c_set, h, Y0});
    mxArray* c_ls = mxCreateString("linesearch_cost");
    mxArray* zero = mxCreateDoubleScalar(0.0);
    mxArray* al = mxCreateDoubleScalar(alpha);
    mxArray* co0 = call(1, {c_ls, h, U, zero})[0];
    mxArray* co1 = call(1, {c_ls, h, U, al})[0];
    mxArray* c_rtr = mxCreateString("rtr");
    mxArray* opts = rtr_opts((int)maxiter, (int)maxinner, 1e-8);
    mxArray* info = call(1, {c_rtr, h, opts})[0];
    mxArray* c_get = mxCreateString("get_point");
    mxArray* Y = call(1, {c_get, h})[0];
    mxArray* c_step = mxCreateString("dual_outer_step");
    std::vector<mxArray*> st = call(5, {c_step, h});
    mxArray* c_y = mxCreateString("dual_get_y");
    mxArray* y = call(1, {c_y, h})[0];
    mxArray* c_S = mxCreateString("get_dual_slack");
    mxArray* X = call(1, {c_S, h})[0];
    mxArray* c_kind = mxCreateString("kind");
    mxArray* kind = call(1, {c_kind, h})[0];
    w.put(Y); w.put(st[3]); w.put(st[4]); w.put(y); w.put(X);
    printf("{\"kind\": %d, \"rows\": %zu, \"cols\": %zu, \"co0\": %.17g, \"co1\": %.17g, \"cost\": %.17g, \"gradnorm\": %.17g, "
           "\"hessvecs\": %d, \"by\": %.17g, \"cex\": %.17g, \"as2\": %.17g}\n",
           (int)mxGetScalar(kind), mxGetM(Y), mxGetN(Y), mxGetScalar(co0), mxGetScalar(co1), field(info, "cost"), field(info, "gradnorm"),
           (int)field(info, "hessvecs"), mxGetScalar(st[0]), mxGetScalar(st[1]), mxGetScalar(st[2]));
    int bad = 0;
    // a second solve without a fresh dual_set_penalty must be refused (the multipliers have moved)
    bad += expect_error("rtr after dual_outer_step without dual_set_penalty", "ManiSDP:hip:call", {c_rtr, h, opts});
    mxArray* c_destroy = mxCreateString("destroy");
    call(0, {c_destroy, h});
    for (mxArray* a : {At, B, dAAt, b, c, Y0, U, c_create, nn, cf, h, c_pen, sg, wf, c_set, c_ls, zero, al, co0, co1, c_rtr, opts, info,
                       c_get, Y, c_step, st[0], st[1], st[2], st[3], st[4], c_y, y, c_S, X, c_kind, kind, c_destroy})
        mxDestroyArray(a);
    return bad;
}

}  // namespace

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "errors";
    try {
        if (mode == "errors") return run_errors();
        if (mode == "onlyunitdiag" && argc == 4) return run_onlyunitdiag(argv[2], argv[3]);
        if (mode == "affine" && argc == 4) return run_affine(argv[2], argv[3]);
        if (mode == "dual" && argc == 4) return run_dual(argv[2], argv[3]);
    } catch (const mex_stub_error& e) {
        fprintf(stderr, "mexErrMsgIdAndTxt(%s): %s\n", e.id.c_str(), e.what());
        return 3;
    }
    fprintf(stderr, "usage: mex_selftest errors | onlyunitdiag <in> <out> | affine <in> <out>\n");
    return 2;
}
