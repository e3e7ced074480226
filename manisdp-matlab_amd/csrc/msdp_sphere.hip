// msdp_sphere.hip -- manifold kernels of the unit-Frobenius-norm sphere (unittrace) and of the Euclidean
// manifold of the generic ManiSDP.m (euclideanfactory.m:59-76: proj = identity, retr = x + d; the same kernels
// with the projection / normalisation terms switched off):
//   proj/tangent  d - x*(x(:)'*d(:))          manopt/manifolds/sphere/spherefactory.m:113
//   retr          (x+d)/norm(x+d,'fro')       spherefactory.m:220-232
// Unlike the oblique manifold the projection needs one GLOBAL inner product, so the tCG
// direction update and the retraction are each split in two launches around that reduction.
#include "msdp_device.h"
#include <math.h>

void msdp_frame_store_fwd();

__device__ __forceinline__ void sph_frame_store(Frame* o, double z_r, double d_Pd, double e_Pd, double e_Pe,
                                                double model_value, double norm_r0, double alpha, double beta,
                                                int active, int j, int stop, int eta_idx) {
    o->z_r = z_r; o->d_Pd = d_Pd; o->e_Pd = e_Pd; o->e_Pe = e_Pe; o->model_value = model_value;
    o->norm_r0 = norm_r0; o->alpha = alpha; o->beta = beta;
    o->active = active; o->j = j; o->stop = stop; o->eta_idx = eta_idx;
}

// tCG.m:227-273 (decisions identical to k_tcg_upd2_obl); v = r + beta*mdelta -> md, partial <Y, v> -> P_AUX
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_upd2a(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    __shared__ double shb[4];
    const Frame* fi = &d.F[1];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const int active = fi->active;
    const double z_r = fi->z_r, d_Pd = fi->d_Pd, e_Pd = fi->e_Pd, e_Pe = fi->e_Pe;
    const double model_value = fi->model_value, norm_r0 = fi->norm_r0, beta0 = fi->beta, alpha = fi->alpha;
    const int j0 = fi->j, stop0 = fi->stop, ix = fi->eta_idx;
    if (!active) {
        if (lead) {
            sph_frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 0, j0, stop0, ix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, d.ctl->k, j0, 0);
        }
        return;
    }
    const Ctl* c = d.ctl;
    const bool bench = c->bench_mode != 0;
    double s1, s2, r_r;
    msdp_sum_partials3_block(d.P, P_S1, P_S2, P_S3, d.G, shb, s1, s2, r_r);
    const double new_model = s1 + 0.5 * s2;                 // tCG.m:227
    const int j = j0 + 1;
    if (!bench && new_model >= model_value) {               // :228
        if (lead) {
            sph_frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta0, 0, j, 6, ix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j, 0);
        }
        return;
    }
    const int nix = ix ^ 1;
    const double norm_r = sqrt(r_r);
    const double nr0t = (c->theta == 1.0) ? norm_r0 : pow(norm_r0, c->theta);
    if (!bench && j >= c->mininner && norm_r <= norm_r0 * fmin(nr0t, c->kappa)) {   // :249
        if (lead) {
            sph_frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, new_model, norm_r0, alpha, beta0, 0, j,
                            (c->kappa < nr0t) ? 3 : 4, nix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j, 0);
        }
        return;
    }
    if (j >= c->maxinner) {
        if (lead) {
            sph_frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, new_model, norm_r0, alpha, beta0, 0, j, stop0, nix);
            d.ctl->tcg_running = 0;
            msdp_publish(d, c->k, j, 0);
        }
        return;
    }
    const double beta = r_r / z_r;                          // :272
    if (lead) {
        sph_frame_store(&d.F[0], r_r, r_r + beta * beta * d_Pd, beta * (e_Pd + alpha * d_Pd), e_Pe, new_model, norm_r0,
                        alpha, beta, 1, j, stop0, nix);
        msdp_publish(d, c->k, j, 1);
    }
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ Yl = c->cur ? d.Y[1] : d.Y[0];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    double pt = 0.0;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 rr = ld2(d.r + i), m = ld2(d.md + i), y = ld2(Yl + i);
        const double2 v = make_double2(rr.x + beta * m.x, rr.y + beta * m.y);    // :273
        st2(d.md + i, v);
        pt += v.x * y.x + v.y * y.y;
    }
    if (d.manifold == MANI_EUCLID) return;                  // tangent = identity: no projection pass follows
    msdp_put_partial(d.P, P_AUX, pt, sh);
}

// mdelta = tangent(x, mdelta)   tCG.m:283
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_upd2b(Dev d) {
    __shared__ double shb[2];
    if (!d.F[0].active) return;
    const double t = msdp_sum_partials_block(d.P, P_AUX, d.G, shb);
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const double* __restrict__ Yl = d.ctl->cur ? d.Y[1] : d.Y[0];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 v = ld2(d.md + i), y = ld2(Yl + i);
        st2(d.md + i, make_double2(v.x - t * y.x, v.y - t * y.y));
    }
}

// x + eta -> Y[prop] (unnormalised), |x+eta|^2 -> P_AUX, <eta, g + .5*Heta> -> P_RD (trustregions.m:549-550)
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_retract_a(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    const Ctl* c = d.ctl;
    if (c->done) return;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    const int cur = c->cur, ix = d.F[0].eta_idx;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ g = cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ eta = ix ? d.eta[1] : d.eta[0];
    const double* __restrict__ Heta = ix ? d.Heta[1] : d.Heta[0];
    double* __restrict__ Yp = cur ? d.Y[0] : d.Y[1];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    double pn = 0.0, prd = 0.0;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 y = ld2(Yl + i), e = ld2(eta + i), he = ld2(Heta + i), gv = ld2(g + i);
        const double2 x = make_double2(y.x + e.x, y.y + e.y);
        st2(Yp + i, x);
        pn += x.x * x.x + x.y * x.y;
        prd += e.x * (gv.x + 0.5 * he.x) + e.y * (gv.y + 0.5 * he.y);
    }
    msdp_put_partials3(d.P, P_AUX, pn, P_RD, prd, -1, 0.0, sh);
}
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_retract_b(Dev d) {
    __shared__ double shb[2];
    if (d.ctl->done) return;
    const double nn = sqrt(msdp_sum_partials_block(d.P, P_AUX, d.G, shb));
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi);
    double* __restrict__ Yp = d.ctl->cur ? d.Y[0] : d.Y[1];
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    for (int64_t i = e0 + 2 * threadIdx.x; i < e1; i += 2 * MSDP_BLOCK) {
        const double2 x = ld2(Yp + i);
        st2(Yp + i, make_double2(x.x / nn, x.y / nn));
    }
}

// test-only single-workgroup forms of proj / retr
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_proj_simple(const double* Y, const double* U, double* V, int64_t cnt) {
    __shared__ double sh[MSDP_WAVES];
    double p = 0.0;
    for (int64_t i = threadIdx.x; i < cnt; i += MSDP_BLOCK) p += Y[i] * U[i];
    const double t = msdp_block_sum(p, sh);
    for (int64_t i = threadIdx.x; i < cnt; i += MSDP_BLOCK) V[i] = U[i] - Y[i] * t;
}
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_retr_simple(const double* Y, const double* U, double* Z, double alpha,
                                                              int64_t cnt) {
    __shared__ double sh[MSDP_WAVES];
    double p = 0.0;
    for (int64_t i = threadIdx.x; i < cnt; i += MSDP_BLOCK) { const double x = Y[i] + alpha * U[i]; p += x * x; }
    const double nn = sqrt(msdp_block_sum(p, sh));
    for (int64_t i = threadIdx.x; i < cnt; i += MSDP_BLOCK) Z[i] = (Y[i] + alpha * U[i]) / nn;
}

int msdp_sphere_upd2(msdp_handle h) {
    hipLaunchKernelGGL(k_sph_upd2a, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    if (h->d.manifold == MANI_EUCLID) return 0;
    int rc = msdp_allreduce_partials(h, P_AUX, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sph_upd2b, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_sphere_retract(msdp_handle h) {
    hipLaunchKernelGGL(k_sph_retract_a, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    int rc = msdp_allreduce_partials(h, P_RD, 2);   // P_RD and P_AUX are adjacent
    if (rc) return rc;
    if (h->d.manifold == MANI_EUCLID) return 0;     // retr(x, d) = x + d (euclideanfactory.m:67-76)
    hipLaunchKernelGGL(k_sph_retract_b, dim3(h->d.G), dim3(MSDP_BLOCK), 0, h->stream, h->d);
    HIPCHK(hipGetLastError());
    return 0;
}
// Row-sharded forms: the partial sum of the local rows goes through the all-reduced partial array (one workgroup, slot 0
// of P_AUX; the other slots are zeroed), then every rank applies the global scalar to its rows.
__global__ __launch_bounds__(MSDP_BLOCK) void k_sph_simple_partial(const double* Y, const double* U, double alpha, int mode, int64_t cnt, double* P) {
    __shared__ double sh[MSDP_WAVES];
    double p = 0.0;
    for (int64_t i = threadIdx.x; i < cnt; i += MSDP_BLOCK) {
        if (mode == 0) p += Y[i] * U[i];
        else { const double x = Y[i] + alpha * U[i]; p += x * x; }
    }
    const double t = msdp_block_sum(p, sh);
    for (int i = threadIdx.x; i < MSDP_MAX_GRID; i += MSDP_BLOCK) P[P_AUX * MSDP_MAX_GRID + i] = (i == 0) ? t : 0.0;
}
__global__ void k_sph_simple_apply(const double* Y, const double* U, double* Z, double alpha, int mode, int64_t cnt, const double* P) {
    const double t = P[P_AUX * MSDP_MAX_GRID];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x)
        Z[i] = mode == 0 ? U[i] - Y[i] * t : (Y[i] + alpha * U[i]) / sqrt(t);
}
static int sph_simple_sharded(msdp_handle h, const double* Y, const double* U, double* Z, double alpha, int mode) {
    const int64_t cnt = (int64_t)h->d.n_loc * h->d.ld;
    hipLaunchKernelGGL(k_sph_simple_partial, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, Y, U, alpha, mode, cnt, h->d.P);
    HIPCHK(hipGetLastError());
    int rc = msdp_allreduce_partials(h, P_AUX, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sph_simple_apply, dim3(256), dim3(256), 0, h->stream, Y, U, Z, alpha, mode, cnt, (const double*)h->d.P);
    HIPCHK(hipGetLastError());
    return 0;
}
__global__ void k_euc_axpy(const double* Y, const double* U, double* Z, double alpha, double ycoef, int64_t cnt) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x)
        Z[i] = ycoef * Y[i] + alpha * U[i];
}
int msdp_sphere_proj(msdp_handle h, const double* Y, const double* U, double* V) {
    if (h->d.manifold == MANI_EUCLID) {
        hipLaunchKernelGGL(k_euc_axpy, dim3(256), dim3(256), 0, h->stream, Y, U, V, 1.0, 0.0, (int64_t)h->d.n_loc * h->d.ld);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->use_comm) return sph_simple_sharded(h, Y, U, V, 0.0, 0);
    hipLaunchKernelGGL(k_sph_proj_simple, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, Y, U, V, (int64_t)h->d.n_loc * h->d.ld);
    HIPCHK(hipGetLastError());
    return 0;
}
int msdp_sphere_retr(msdp_handle h, const double* Y, const double* U, double* Z, double alpha) {
    if (h->d.manifold == MANI_EUCLID) {
        hipLaunchKernelGGL(k_euc_axpy, dim3(256), dim3(256), 0, h->stream, Y, U, Z, alpha, 1.0, (int64_t)h->d.n_loc * h->d.ld);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->use_comm) return sph_simple_sharded(h, Y, U, Z, alpha, 1);
    hipLaunchKernelGGL(k_sph_retr_simple, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, Y, U, Z, alpha,
                       (int64_t)h->d.n_loc * h->d.ld);
    HIPCHK(hipGetLastError());
    return 0;
}
