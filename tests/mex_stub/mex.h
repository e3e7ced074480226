/*
 * mex.h -- NOT MathWorks' header.  A small stand-in written for this repository's own tests
 * (tests/test_mex_shim.py): just enough of the mx* / mex* API, with real storage behind it, for
 * manisdp-matlab_amd/matlab/manisdp_mex.cpp to be compiled by g++ and DRIVEN by
 * tests/mex_stub/mex_selftest.cpp the way MATLAB would drive it (create arrays, call mexFunction,
 * read the outputs).  It is test infrastructure: nothing in the product includes it, and it makes
 * no claim about MATLAB's ABI -- only about the source-level API the shim uses.
 *
 * Semantics kept from MATLAB: column-major full double matrices; sparse = compressed columns with
 * mwIndex (size_t) Ir/Jc, 0-based; mexErrMsgIdAndTxt does not return (here: throws mex_stub_error).
 */
#ifndef MEX_STUB_H
#define MEX_STUB_H

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

typedef size_t mwSize;
typedef size_t mwIndex;

typedef enum { mxUNKNOWN_CLASS = 0, mxSTRUCT_CLASS = 2, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxUINT64_CLASS = 13 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

struct mxArray_tag {
    mxClassID cls = mxUNKNOWN_CLASS;
    mwSize m = 0, n = 0;
    bool sparse = false;
    std::vector<double> pr;                 /* doubles (full: m*n column-major; sparse: nnz values) */
    std::vector<mwIndex> ir, jc;            /* sparse pattern */
    std::vector<uint64_t> u64;              /* uint64 data */
    std::string str;                        /* char row vector */
    std::vector<std::pair<std::string, mxArray_tag*>> fields;   /* 1x1 struct */
};
typedef struct mxArray_tag mxArray;

struct mex_stub_error : std::runtime_error {
    std::string id;
    mex_stub_error(const std::string& i, const std::string& msg) : std::runtime_error(msg), id(i) {}
};

/* ---- creation / destruction */
inline mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) {
    mxArray* a = new mxArray(); a->cls = mxDOUBLE_CLASS; a->m = m; a->n = n; a->pr.assign((m * n) != 0 ? m * n : 1, 0.0); return a;
}
inline mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->pr[0] = v; return a; }
inline mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) {
    if (cls == mxDOUBLE_CLASS) return mxCreateDoubleMatrix(m, n, mxREAL);
    mxArray* a = new mxArray(); a->cls = cls; a->m = m; a->n = n; a->u64.assign((m * n) != 0 ? m * n : 1, 0); return a;
}
inline mxArray* mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity) {
    mxArray* a = new mxArray(); a->cls = mxDOUBLE_CLASS; a->m = m; a->n = n; a->sparse = true;
    a->pr.assign(nzmax ? nzmax : 1, 0.0); a->ir.assign(nzmax ? nzmax : 1, 0); a->jc.assign(n + 1, 0); return a;
}
inline mxArray* mxCreateString(const char* s) {
    mxArray* a = new mxArray(); a->cls = mxCHAR_CLASS; a->str = s; a->m = 1; a->n = a->str.size(); return a;
}
inline mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
    mxArray* a = new mxArray(); a->cls = mxSTRUCT_CLASS; a->m = m; a->n = n;
    for (int i = 0; i < nfields; ++i) a->fields.push_back({names[i], nullptr});
    return a;
}
inline void mxDestroyArray(mxArray* a) {
    if (!a) return;
    for (auto& f : a->fields) mxDestroyArray(f.second);
    delete a;
}
inline mxArray* mxDuplicateArray(const mxArray* a) {
    mxArray* b = new mxArray(*a);
    for (auto& f : b->fields) if (f.second) f.second = mxDuplicateArray(f.second);
    return b;
}

/* ---- queries */
inline bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
inline bool mxIsSparse(const mxArray* a) { return a->sparse; }
inline bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
inline bool mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
inline bool mxIsUint64(const mxArray* a) { return a->cls == mxUINT64_CLASS; }
inline bool mxIsEmpty(const mxArray* a) { return a->m == 0 || a->n == 0; }
inline mwSize mxGetM(const mxArray* a) { return a->m; }
inline mwSize mxGetN(const mxArray* a) { return a->n; }
inline size_t mxGetNumberOfElements(const mxArray* a) { return a->m * a->n; }
inline double* mxGetPr(const mxArray* a) { return const_cast<double*>(a->pr.data()); }
inline mwIndex* mxGetIr(const mxArray* a) { return const_cast<mwIndex*>(a->ir.data()); }
inline mwIndex* mxGetJc(const mxArray* a) { return const_cast<mwIndex*>(a->jc.data()); }
inline void* mxGetData(const mxArray* a) {
    return a->cls == mxUINT64_CLASS ? (void*)const_cast<uint64_t*>(a->u64.data()) : (void*)const_cast<double*>(a->pr.data());
}
inline double mxGetScalar(const mxArray* a) {
    if (a->cls == mxUINT64_CLASS) return (double)a->u64[0];
    return a->pr.empty() ? 0.0 : a->pr[0];
}
inline int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (a->cls != mxCHAR_CLASS || a->str.size() + 1 > buflen) return 1;
    memcpy(buf, a->str.c_str(), a->str.size() + 1);
    return 0;
}
inline mxArray* mxGetField(const mxArray* a, mwIndex, const char* name) {
    for (auto& f : a->fields) if (f.first == name) return f.second;
    return nullptr;
}
inline void mxSetField(mxArray* a, mwIndex, const char* name, mxArray* v) {
    for (auto& f : a->fields) if (f.first == name) { mxDestroyArray(f.second); f.second = v; return; }
    a->fields.push_back({name, v});
}

/* ---- mex* */
inline void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[2048];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    throw mex_stub_error(id, buf);
}
inline int mexPrintf(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); const int r = vprintf(fmt, ap); va_end(ap); return r;
}
typedef void (*mex_stub_exit_fn)(void);
inline mex_stub_exit_fn& mex_stub_exit_hook() { static mex_stub_exit_fn f = nullptr; return f; }
inline int mexAtExit(mex_stub_exit_fn f) { mex_stub_exit_hook() = f; return 0; }
inline void mexLock(void) {}
inline void mexUnlock(void) {}

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#endif /* MEX_STUB_H */
