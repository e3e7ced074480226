/*
 * ORACLE (test infrastructure, not product code) -- plain-C restatement of the onlyunitdiag hot path:
 * the closures of src/primal/ManiSDP_onlyunitdiag.m:117-130, the oblique manifold :132-156 and the
 * Riemannian trust-region / tCG loop of manopt7.0/manopt/solvers/trustregions/{trustregions.m:405-729,
 * tCG.m:95-292} (same restatement as oracle/manopt_rtr.py, which it is tested against).
 *
 * Purpose: (i) a second, independent implementation that pins the NumPy oracle; (ii) the CPU baseline
 * of bench.py ("port"), multi-threaded with OpenMP over rows so that the host cores of the GPU box are
 * actually used -- the reference's MATLAB built-ins (sparse*dense mtimes, elementwise ops) are
 * multithreaded too.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may call this.
 *
 * Layout: factors are n x p row-major (= MATLAB's p x n column-major).  C is CSR (symmetric).
 * Per-point state (YC, eG) is kept per point ("correct" variant of quirk Q1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int n, p;
    const int64_t* rp;
    const int32_t* ci;
    const double* cv;
} csr_t;

typedef struct {
    double cost, gradnorm, Delta;
    int iters, hessvecs, accepted, rejected, cost_evals, last_stop_inner;
} oc_stats;

/* Cap the OpenMP team: on a many-core host (the GPU box reports 256 logical CPUs, possibly under a cgroup
 * quota) a 256-thread team on these short row loops spends its time in barriers. */
void oc_set_threads(int t) {
#ifdef _OPENMP
    if (t >= 1) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

int oc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* W = C*X (n x p) */
static void spmm(const csr_t* c, const double* X, double* W) {
    const int n = c->n, p = c->p;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double* w = W + (size_t)i * p;
        for (int k = 0; k < p; ++k) w[k] = 0.0;
        for (int64_t t = c->rp[i]; t < c->rp[i + 1]; ++t) {
            const double v = c->cv[t];
            const double* x = X + (size_t)c->ci[t] * p;
            for (int k = 0; k < p; ++k) w[k] += v * x[k];
        }
    }
}

static double dot_all(const double* a, const double* b, size_t cnt) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < cnt; ++i) s += a[i] * b[i];
    return s;
}

/* cost: YC = C*Y; eG = rowsum(YC.*Y); f = .5*sum(eG)   (ManiSDP_onlyunitdiag.m:118-120) */
static double cost_fn(const csr_t* c, const double* Y, double* YC, double* eG) {
    const int n = c->n, p = c->p;
    spmm(c, Y, YC);
    double f = 0.0;
#pragma omp parallel for reduction(+ : f) schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < p; ++k) s += YC[(size_t)i * p + k] * Y[(size_t)i * p + k];
        eG[i] = s;
        f += s;
    }
    return 0.5 * f;
}

/* G = YC - Y.*eG   (:124) */
static void grad_fn(const csr_t* c, const double* Y, const double* YC, const double* eG, double* G) {
    const int n = c->n, p = c->p;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < p; ++k) G[(size_t)i * p + k] = YC[(size_t)i * p + k] - Y[(size_t)i * p + k] * eG[i];
}

/* H = eH - Y.*rowsum(Y.*eH) - U.*eG with eH = C*U   (:128-129) */
void oc_hessvec(int n, int p, const int64_t* rp, const int32_t* ci, const double* cv, const double* Y, const double* U,
                const double* eG, double* H) {
    csr_t c = {n, p, rp, ci, cv};
    spmm(&c, U, H);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < p; ++k) s += Y[(size_t)i * p + k] * H[(size_t)i * p + k];
        for (int k = 0; k < p; ++k)
            H[(size_t)i * p + k] = H[(size_t)i * p + k] - Y[(size_t)i * p + k] * s - U[(size_t)i * p + k] * eG[i];
    }
}

void oc_cost_state(int n, int p, const int64_t* rp, const int32_t* ci, const double* cv, const double* Y, double* f,
                   double* eG, double* G) {
    csr_t c = {n, p, rp, ci, cv};
    double* YC = (double*)malloc((size_t)n * p * sizeof(double));
    *f = cost_fn(&c, Y, YC, eG);
    grad_fn(&c, Y, YC, eG, G);
    free(YC);
}

/* tangent(X, U) = U - X.*rowsum(X.*U)  (:138-139), in place */
static void tangent_inplace(int n, int p, const double* X, double* U) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < p; ++k) s += X[(size_t)i * p + k] * U[(size_t)i * p + k];
        for (int k = 0; k < p; ++k) U[(size_t)i * p + k] -= X[(size_t)i * p + k] * s;
    }
}

/* y = (x + d)./rownorm  (:142-145) */
static void retract(int n, int p, const double* x, const double* d, double* y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < p; ++k) { const double v = x[(size_t)i * p + k] + d[(size_t)i * p + k]; y[(size_t)i * p + k] = v; s += v * v; }
        s = sqrt(s);
        for (int k = 0; k < p; ++k) y[(size_t)i * p + k] /= s;
    }
}

static void axpy_to(size_t cnt, const double* x, double a, const double* y, double* out) {  /* out = x + a*y */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cnt; ++i) out[i] = x[i] + a * y[i];
}

/* [Y, fx, info] = trustregions(problem, Y, opts) for the onlyunitdiag problem. */
int oc_rtr_onlyunitdiag(int n, int p, const int64_t* rp, const int32_t* ci, const double* cv, double* Y, int maxiter,
                        int maxinner, double tolgradnorm, oc_stats* st) {
    const double kappa = 0.1, theta = 1.0, rho_prime = 0.1, rho_regularization = 1e3, eps = 2.220446049250313e-16;
    const int mininner = 1;
    csr_t c = {n, p, rp, ci, cv};
    const size_t cnt = (size_t)n * p;
    double* buf = (double*)malloc((12 * cnt + 2 * (size_t)n) * sizeof(double));
    if (!buf) return -1;
    double *YC = buf, *YCp = YC + cnt, *g = YCp + cnt, *eta = g + cnt, *Heta = eta + cnt, *r = Heta + cnt,
           *md = r + cnt, *Hmd = md + cnt, *neta = Hmd + cnt, *nHeta = neta + cnt, *xp = nHeta + cnt, *tmp = xp + cnt;
    double *eG = tmp + cnt, *eGp = eG + n;
    memset(st, 0, sizeof(*st));
    const double Delta_bar = M_PI * sqrt((double)n);                 /* trustregions.m:363-369 with :137 */
    double Delta = Delta_bar / 8.0;                                   /* :370-372 */
    double fx = cost_fn(&c, Y, YC, eG);                               /* :405 */
    st->cost_evals = 1;
    grad_fn(&c, Y, YC, eG, g);
    double norm_grad = sqrt(dot_all(g, g, cnt));                      /* :406 */
    int k = 0;
    while (1) {                                                       /* :441 */
        if (norm_grad < tolgradnorm) break;                           /* stoppingcriterion.m:51-56 */
        if (k >= maxiter) break;                                      /* stoppingcriterion.m:67-72 */
        /* ---- tCG (tCG.m:102-289) */
        memset(eta, 0, cnt * sizeof(double));
        memset(Heta, 0, cnt * sizeof(double));
        memcpy(r, g, cnt * sizeof(double));
        memcpy(md, g, cnt * sizeof(double));
        double e_Pe = 0.0, r_r = dot_all(r, r, cnt), norm_r0 = sqrt(r_r), z_r = r_r, d_Pd = z_r, e_Pd = 0.0, model_value = 0.0;
        int stop = 5, j;
        for (j = 1; j <= maxinner; ++j) {
            oc_hessvec(n, p, rp, ci, cv, Y, md, eG, Hmd);             /* tCG.m:163 */
            const double d_Hd = dot_all(md, Hmd, cnt);                /* :166 */
            const double alpha = z_r / d_Hd;                          /* :170 */
            const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;   /* :173 */
            if (d_Hd <= 0 || e_Pe_new >= Delta * Delta) {             /* :183 */
                const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;   /* :188 */
                axpy_to(cnt, eta, -tau, md, eta);
                axpy_to(cnt, Heta, -tau, Hmd, Heta);
                stop = d_Hd <= 0 ? 1 : 2;
                break;
            }
            e_Pe = e_Pe_new;
            axpy_to(cnt, eta, -alpha, md, neta);                      /* :215 */
            axpy_to(cnt, Heta, -alpha, Hmd, nHeta);                   /* :220 */
            const double new_model = dot_all(neta, g, cnt) + 0.5 * dot_all(neta, nHeta, cnt);   /* :227 */
            if (new_model >= model_value) { stop = 6; break; }        /* :228 */
            memcpy(eta, neta, cnt * sizeof(double));
            memcpy(Heta, nHeta, cnt * sizeof(double));
            model_value = new_model;
            axpy_to(cnt, r, -alpha, Hmd, r);                          /* :238 */
            r_r = dot_all(r, r, cnt);
            const double norm_r = sqrt(r_r);
            if (j >= mininner && norm_r <= norm_r0 * fmin(pow(norm_r0, theta), kappa)) {   /* :249 */
                stop = kappa < pow(norm_r0, theta) ? 3 : 4;
                break;
            }
            const double zold = z_r;
            z_r = r_r;
            const double beta = z_r / zold;                           /* :272 */
            axpy_to(cnt, r, beta, md, md);                            /* :273 */
            tangent_inplace(n, p, Y, md);                             /* :283 */
            e_Pd = beta * (e_Pd + alpha * d_Pd);                      /* :286 */
            d_Pd = z_r + beta * beta * d_Pd;                          /* :287 */
        }
        if (j > maxinner) j = maxinner;
        st->hessvecs += j;
        st->last_stop_inner = stop;
        /* ---- trustregions.m:540-729 */
        retract(n, p, Y, eta, xp);
        const double fxp = cost_fn(&c, xp, YCp, eGp);
        st->cost_evals++;
        double rhonum = fx - fxp;
        axpy_to(cnt, g, 0.5, Heta, tmp);
        double rhoden = -dot_all(eta, tmp, cnt);
        const double rho_reg = fmax(1.0, fabs(fx)) * eps * rho_regularization;
        rhonum += rho_reg;
        rhoden += rho_reg;
        const int model_decreased = rhoden >= 0;
        const double rho = rhonum / rhoden;
        if (rho < 0.25 || !model_decreased || isnan(rho)) Delta = Delta / 4.0;
        else if (rho > 0.75 && (stop == 1 || stop == 2)) Delta = fmin(2.0 * Delta, Delta_bar);
        if (model_decreased && rho > rho_prime) {
            memcpy(Y, xp, cnt * sizeof(double));
            memcpy(YC, YCp, cnt * sizeof(double));
            memcpy(eG, eGp, (size_t)n * sizeof(double));
            fx = fxp;
            grad_fn(&c, Y, YC, eG, g);
            norm_grad = sqrt(dot_all(g, g, cnt));
            st->accepted++;
        } else {
            st->rejected++;
        }
        ++k;
    }
    st->cost = fx; st->gradnorm = norm_grad; st->Delta = Delta; st->iters = k;
    free(buf);
    return 0;
}
