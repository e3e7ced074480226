// Drives the REAL k_tcg_upd1 / k_tcg_upd2_obl / k_hess kernels with a synthetic Dev to bisect their cost.
#include "../manisdp-matlab_amd/csrc/msdp_kernels.hip"
#include <cstdio>
#include <cstring>
#include <vector>
void msdp_set_error(const char*, ...) {}
int msdp_dense_costgrad(msdp_handle, int) { return -1; }
int msdp_dense_hess(msdp_handle) { return -1; }
int msdp_affine_costgrad(msdp_handle, int) { return -1; }
int msdp_affine_hess(msdp_handle) { return -1; }
int msdp_sphere_upd2(msdp_handle) { return -1; }
int msdp_sphere_retract(msdp_handle) { return -1; }
int msdp_allreduce_partials(msdp_handle, int, int) { return 0; }
int msdp_allgather_rows(msdp_handle h, const double* l) { h->d.full = (double*)l; return 0; }
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <typename F>
float timeit(F enqueue, int reps, hipStream_t s) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < reps; ++i) enqueue();
    (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 20; ++w) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(a, s); for (int w = 0; w < 10; ++w) (void)hipGraphLaunch(ge, s); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms * 1e3f / reps / 10;
}

int main(int argc, char** argv) {
    Dev d; memset(&d, 0, sizeof(d));
    d.n = d.n_loc = 20000; d.p = 32; d.ld = 32; d.G = 320;
    d.variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t cnt = (size_t)d.n * d.ld;
    double** vecs[] = {&d.Y[0], &d.Y[1], &d.Gr[0], &d.Gr[1], &d.eta[0], &d.eta[1], &d.Heta[0], &d.Heta[1], &d.r, &d.md, &d.Hmd, &d.W0, &d.W1};
    std::vector<double> init(cnt);
    const int zero = argc > 2 ? atoi(argv[2]) : 0;
    for (size_t i = 0; i < cnt; ++i) init[i] = zero == 1 ? 0.0 : (zero == 2 ? 1.0 : 1e-3 * ((i * 2654435761u) % 1000) - 0.5);
    for (double** v : vecs) { CK(hipMalloc(v, cnt * 8)); CK(hipMemcpy(*v, init.data(), cnt * 8, hipMemcpyHostToDevice)); }
    d.full = d.md;
    CK(hipMalloc(&d.eG[0], d.n * 8)); CK(hipMalloc(&d.eG[1], d.n * 8));
    CK(hipMemset(d.eG[0], 0, d.n * 8));
    CK(hipMalloc(&d.P, MSDP_NPART * MSDP_MAX_GRID * 8)); CK(hipMemset(d.P, 0, MSDP_NPART * MSDP_MAX_GRID * 8));
    CK(hipMalloc(&d.ctl, sizeof(Ctl))); CK(hipMalloc(&d.F, 2 * sizeof(Frame)));
    Ctl c; memset(&c, 0, sizeof(c)); c.bench_mode = 1; c.Delta = 10; c.maxinner = 1 << 30; c.mininner = 1; c.kappa = 0.1; c.theta = 1;
    CK(hipMemcpy(d.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
    Frame f; memset(&f, 0, sizeof(f)); f.active = 1; f.z_r = 1; f.d_Pd = 1; f.norm_r0 = 1; f.alpha = 1e-3;
    Frame ff[2] = {f, f};
    CK(hipMemcpy(d.F, ff, sizeof(ff), hipMemcpyHostToDevice));
    // 5-point stencil CSR (G81-like)
    std::vector<int> rp(d.n + 1), ci; std::vector<double> cv;
    for (int i = 0; i < d.n; ++i) {
        rp[i] = (int)ci.size();
        int nb[5] = {i, (i + 1) % d.n, (i + d.n - 1) % d.n, (i + 200) % d.n, (i + d.n - 200) % d.n};
        for (int k = 0; k < 5; ++k) { ci.push_back(nb[k]); cv.push_back(0.25); }
    }
    rp[d.n] = (int)ci.size();
    int *drp, *dci; double* dcv;
    CK(hipMalloc(&drp, rp.size() * 4)); CK(hipMalloc(&dci, ci.size() * 4)); CK(hipMalloc(&dcv, cv.size() * 8));
    CK(hipMemcpy(drp, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcv, cv.data(), cv.size() * 8, hipMemcpyHostToDevice));
    d.rowptr = drp; d.colind = dci; d.cval = dcv; d.nnz = ci.size();
    hipStream_t s; CK(hipStreamCreate(&s));
    dim3 grid(d.G), blk(MSDP_BLOCK);
    if (argc > 3) {   // pmc mode: a few plain launches
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(k_tcg_upd1, grid, blk, 0, s, d);
        (void)hipStreamSynchronize(s);
        printf("pmc mode done\n");
        return 0;
    }
    float t_h = timeit([&] { hipLaunchKernelGGL((k_hess_sparse_obl<16, 1>), grid, blk, 0, s, d); }, 50, s);
    float t_1 = timeit([&] { hipLaunchKernelGGL(k_tcg_upd1, grid, blk, 0, s, d); }, 50, s);
    float t_2 = timeit([&] { hipLaunchKernelGGL((k_tcg_upd2_obl<16, 1>), grid, blk, 0, s, d); }, 50, s);
    float t_all = timeit([&] {
        hipLaunchKernelGGL((k_hess_sparse_obl<16, 1>), grid, blk, 0, s, d);
        hipLaunchKernelGGL(k_tcg_upd1, grid, blk, 0, s, d);
        hipLaunchKernelGGL((k_tcg_upd2_obl<16, 1>), grid, blk, 0, s, d); }, 30, s);
    printf("init %d variant %d: hess %.2f  upd1 %.2f  upd2 %.2f  trip %.2f us\n", zero, d.variant, t_h, t_1, t_2, t_all);
    return 0;
}
