// msdp_escape.hip -- few-eigenvector saddle escape on the device.
//
// The reference calls eig(full(S)) (O(n^3), dense n x n: 3.2 GB at n = 20000) once per outer
// iteration just to read lambda_min, lambda_max and <= delta bottom eigenvectors of the dual
// slack S = C - diag(z)  (ManiSDP_onlyunitdiag.m:49-51,74-83).  Here the same quantities come
// from a deflated Lanczos process that only needs S*v:
//   * at a stationary point of the rank-p problem S*Y = 0 (the Riemannian gradient is S*Y), so
//     span(Y) is the (near-)kernel where the eigenvalues cluster at 0 -- the cluster that made
//     plain ARPACK stall.  Q = orth(Y) is deflated: Lanczos runs on the orthogonal
//     complement, where a negative eigenvalue (escape direction) is an isolated extreme one;
//   * full re-orthogonalisation (classical Gram-Schmidt applied twice) against [Q | V] keeps the
//     process in the complement and the Ritz values free of ghosts;
//   * a final Rayleigh-Ritz on [Q | negative Ritz vectors] recouples the two blocks, so lambda_min
//     is accurate even when S*Y is only approximately zero.
// All length-n work (SpMV / dense GEMV, block dot products, block axpy) runs in HIP kernels; the host
// keeps the m x m tridiagonal and the (r+k) x (r+k) Rayleigh-Ritz matrix (Jacobi / implicit QL).
#include "msdp_device.h"
#include <math.h>
#include <algorithm>
#include <cstring>
#include <vector>

int msdp_dense_nS(int n);

// ---------------------------------------------------------------- kernels
// w = S*v for S = C - diag(z), sparse C (one thread per row; rows are short)
__global__ void k_sv_sparse(int n, const int* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ cv,
                            const double* __restrict__ z, const double* __restrict__ v, double* __restrict__ w) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int t = rp[i]; t < rp[i + 1]; ++t) acc = fma(cv[t], v[ci[t]], acc);
        w[i] = acc - z[i] * v[i];
    }
}
// dense C (n x nS row-major): one wave per row
__global__ __launch_bounds__(256) void k_sv_dense(int n, int nS, const double* __restrict__ Cd, const double* __restrict__ z,
                                                   const double* __restrict__ v, double* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* cr = Cd + (int64_t)row * nS;
    double acc = 0.0;
    for (int j = 2 * lane; j < n; j += 128) {
        const double2 c2 = ld2(cr + j);
        acc += c2.x * v[j] + ((j + 1 < n) ? c2.y * v[j + 1] : 0.0);
    }
    acc = msdp_wave_sum(acc);
    if (lane == 0) w[row] = acc - z[row] * v[row];
}
// h[c] = <B_c, w>, c = 0..nb-1 (columns contiguous, stride ldb): one workgroup per column
__global__ __launch_bounds__(MSDP_BLOCK) void k_multidot(int n, const double* __restrict__ B, int64_t ldb,
                                                        const double* __restrict__ w, double* __restrict__ h) {
    __shared__ double sh[MSDP_WAVES];
    const double* bc = B + (int64_t)blockIdx.x * ldb;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += MSDP_BLOCK) acc = fma(bc[i], w[i], acc);
    const double s = msdp_block_sum(acc, sh);
    if (threadIdx.x == 0) h[blockIdx.x] = s;
}
// w += sign * sum_c h[c] * B_c
__global__ void k_multiaxpy(int n, int nb, const double* __restrict__ B, int64_t ldb, const double* __restrict__ h,
                            double sign, double* __restrict__ w) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int c = 0; c < nb; ++c) acc = fma(h[c], B[(int64_t)c * ldb + i], acc);
        w[i] += sign * acc;
    }
}
__global__ void k_scale_copy(int n, const double* __restrict__ src, double s, double* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = s * src[i];
}
// column c of the row-major factor (n x ld) -> contiguous vector
__global__ void k_extract_col(int n, int ld, int c, const double* __restrict__ Y, double* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = Y[(int64_t)i * ld + c];
}
__global__ void k_fill_hash(int n, unsigned seed, double* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        dst[i] = (double)x / 4294967296.0 - 0.5;
    }
}

// ---------------------------------------------------------------- host helpers
// Symmetric tridiagonal eigen-decomposition, implicit QL with optional vectors (EISPACK tql2).
static bool tql2(int n, std::vector<double>& d, std::vector<double>& e, std::vector<double>* Z) {
    if (n == 0) return true;
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 200) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0 ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i], b = c * e[i];
                    e[i + 1] = (r = hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    if (Z) {
                        for (int k = 0; k < n; ++k) {
                            double* zk = Z->data() + (size_t)k * n;
                            f = zk[i + 1];
                            zk[i + 1] = s * zk[i] + c * f;
                            zk[i] = c * zk[i] - s * f;
                        }
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    return true;
}

// Dense symmetric eigen-decomposition by cyclic Jacobi (small matrices only). A is n x n row-major,
// overwritten; eigenvalues in w, eigenvectors in the columns of V (row-major).
static void jacobi_eig(int n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& V) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (off <= 1e-32 * (diag + 1e-300)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

struct EscCtx {
    msdp_handle h;
    int n;
    int64_t ldb;
    double* B;       // basis columns [Q | V]
    double* w;       // work vector
    double* w2;
    double* hbuf;    // device coefficient buffer
    std::vector<double> hhost;
    const double* z;
};

static int sapply(EscCtx& c, const double* v, double* w) {
    msdp_handle h = c.h;
    const Dev& d = h->d;
    if (d.costkind == COST_SPARSE)
        hipLaunchKernelGGL(k_sv_sparse, dim3((c.n + 255) / 256), dim3(256), 0, h->stream, c.n, d.rowptr, d.colind, d.cval, c.z, v, w);
    else
        hipLaunchKernelGGL(k_sv_dense, dim3((c.n + 3) / 4), dim3(256), 0, h->stream, c.n, msdp_dense_nS(c.n), d.Cd, c.z, v, w);
    HIPCHK(hipGetLastError());
    return 0;
}

// w <- w - B(:,0:nb) (B' w), twice (CGS2); returns the coefficients of the FIRST pass in hhost
static int orth_against(EscCtx& c, int nb, double* w, bool keep_coeffs) {
    msdp_handle h = c.h;
    for (int pass = 0; pass < 2 && nb > 0; ++pass) {
        hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(MSDP_BLOCK), 0, h->stream, c.n, c.B, c.ldb, w, c.hbuf);
        HIPCHK(hipGetLastError());
        if (pass == 0 && keep_coeffs) {
            c.hhost.resize(nb);
            HIPCHK(hipMemcpyAsync(c.hhost.data(), c.hbuf, nb * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        hipLaunchKernelGGL(k_multiaxpy, dim3((c.n + 255) / 256), dim3(256), 0, h->stream, c.n, nb, c.B, c.ldb, c.hbuf, -1.0, w);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

static int dev_norm(EscCtx& c, const double* w, double* out) {
    msdp_handle h = c.h;
    hipLaunchKernelGGL(k_multidot, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, c.n, w, (int64_t)0, w, c.hbuf);
    HIPCHK(hipGetLastError());
    double v = 0.0;
    HIPCHK(hipMemcpyAsync(&v, c.hbuf, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = sqrt(v > 0 ? v : 0.0);
    return 0;
}

int msdp_escape_impl(msdp_handle h, int k, double tol, int maxit, double* lam_out, double* V_out, double* lmax_out,
                     int* iters_out) {
    Dev& d = h->d;
    if (h->kind != MSDP_KIND_ONLYUNITDIAG) { msdp_set_error("escape_eigs: implemented for onlyunitdiag handles"); return MSDP_EUNSUPPORTED; }
    if (h->nranks != 1) { msdp_set_error("escape_eigs: single-GPU only in this build"); return MSDP_EUNSUPPORTED; }
    if (k < 1) { msdp_set_error("escape_eigs: k >= 1"); return MSDP_EINVAL; }
    const int n = d.n, p = d.p;
    if (maxit < 8) maxit = 8;
    if (maxit > n - p - 1) maxit = std::max(1, n - p - 1);
    const int cur = h->h_ctl->cur;
    const int kk = std::min(k + 4, maxit);                    // Ritz vectors kept for the final Rayleigh-Ritz
    const int maxcols = p + maxit + 1 + kk + (p + kk);
    EscCtx c;
    c.h = h; c.n = n; c.ldb = n; c.z = d.eG[cur];
    double* mem = nullptr;
    hipError_t me = hipMalloc((void**)&mem, ((size_t)maxcols * n + 2 * (size_t)n + (size_t)maxcols + 64) * sizeof(double));
    if (me != hipSuccess) { msdp_set_error("escape_eigs: workspace allocation failed"); return MSDP_ENOMEM; }
    c.B = mem; c.w = mem + (size_t)maxcols * n; c.w2 = c.w + n; c.hbuf = c.w2 + n;
    int rc = 0;
    std::vector<double> alpha, beta;      // tridiagonal
    int r = 0, m = 0;
    double lam_max = 0.0;
    std::vector<double> theta, S;         // Ritz values / vectors of T
#define ESC_CHECK(x) do { rc = (x); if (rc) goto done; } while (0)
#define ESC_HIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { msdp_set_error("%s: %s", #x, hipGetErrorString(_e)); rc = MSDP_EHIP; goto done; } } while (0)
    {
        // ---- Q = orth(columns of Y), modified Gram-Schmidt with re-orthogonalisation, rank-revealing
        // Deflate span(Y) only where it IS the near-kernel of S, i.e. at (near-)stationary points:
        // |S*Y|_F is the Riemannian gradient norm of the last RTR call.  Away from stationarity the
        // spectrum has no cluster at 0 and plain Lanczos is both sufficient and more accurate.
        const double gnorm = h->h_ctl->norm_grad;
        const bool deflate = h->gradnorm_valid && gnorm <= 1e-6 * std::max(1.0, fabs(h->h_ctl->fx));
        double ynorm_max = 0.0;
        for (int cidx = 0; deflate && cidx < p; ++cidx) {
            double* q = c.B + (size_t)r * n;
            hipLaunchKernelGGL(k_extract_col, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, d.ld, cidx, d.Y[cur], q);
            double n0; ESC_CHECK(dev_norm(c, q, &n0));
            ynorm_max = std::max(ynorm_max, n0);
            ESC_CHECK(orth_against(c, r, q, false));
            double n1; ESC_CHECK(dev_norm(c, q, &n1));
            if (n1 > 1e-8 * std::max(ynorm_max, 1e-300) && n1 > 1e-10 * n0) {
                hipLaunchKernelGGL(k_scale_copy, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, q, 1.0 / n1, q);
                ++r;
            }
        }
        // ---- Lanczos on the complement of span(Q) with full re-orthogonalisation
        double* V = c.B + (size_t)r * n;
        hipLaunchKernelGGL(k_fill_hash, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, 12345u, V);
        ESC_CHECK(orth_against(c, r, V, false));
        double nv; ESC_CHECK(dev_norm(c, V, &nv));
        if (nv == 0.0) { msdp_set_error("escape_eigs: start vector vanished"); rc = MSDP_EHIP; goto done; }
        hipLaunchKernelGGL(k_scale_copy, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, V, 1.0 / nv, V);
        int next_check = 16;
        bool converged = false;
        for (m = 0; m < maxit; ) {
            double* vj = V + (size_t)m * n;
            ESC_CHECK(sapply(c, vj, c.w));
            ESC_CHECK(orth_against(c, r + m + 1, c.w, true));          // coefficients: [Q'w | V'w]
            double bnext; ESC_CHECK(dev_norm(c, c.w, &bnext));        // syncs: hhost is valid
            alpha.push_back(c.hhost[r + m]);
            ++m;
            const bool breakdown = bnext <= 1e-13 * (fabs(alpha.back()) + 1.0);
            if (!breakdown && m < maxit) {
                beta.push_back(bnext);
                hipLaunchKernelGGL(k_scale_copy, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, c.w, 1.0 / bnext, V + (size_t)m * n);
            }
            if (m == next_check || m == maxit || breakdown) {
                std::vector<double> dd(alpha), ee(m, 0.0);
                for (int i = 1; i < m; ++i) ee[i] = beta[i - 1];
                S.assign((size_t)m * m, 0.0);
                for (int i = 0; i < m; ++i) S[(size_t)i * m + i] = 1.0;
                if (!tql2(m, dd, ee, &S)) { msdp_set_error("escape_eigs: QL failed"); rc = MSDP_EHIP; goto done; }
                theta = dd;
                std::vector<int> ord(m);
                for (int i = 0; i < m; ++i) ord[i] = i;
                std::sort(ord.begin(), ord.end(), [&](int a, int b) { return theta[a] < theta[b]; });
                lam_max = theta[ord[m - 1]];
                // residual estimates |beta_m * s_{m,i}| for the wanted (smallest) Ritz pairs
                const double bm = breakdown ? 0.0 : bnext;
                int want = std::min(k, m), ok = 0;
                const double scale = std::max(fabs(theta[ord[0]]), fabs(lam_max)) + 1e-300;
                for (int t = 0; t < want; ++t) {
                    const int i = ord[t];
                    const double res = fabs(bm * S[(size_t)(m - 1) * m + i]);
                    // only negative (escape) directions need to be resolved as vectors; once the smallest Ritz
                    // value is converged and non-negative the certificate lambda_min >= 0 is all that is needed
                    if (res <= tol * scale) ++ok;
                    else break;
                    if (theta[i] > tol * scale) { ok = want; break; }
                }
                const double res_top = fabs(bm * S[(size_t)(m - 1) * m + ord[m - 1]]);
                if ((ok == want && res_top <= 1e-3 * scale) || breakdown) { converged = true; break; }
                next_check = std::min(maxit, m + std::max(16, m / 4));
            }
        }
        (void)converged;
        // ---- Ritz vectors of the kk smallest Ritz values -> columns X right after the Lanczos basis
        std::vector<int> ord(m);
        for (int i = 0; i < m; ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return theta[a] < theta[b]; });
        const int nx = std::min(kk, m);
        double* X = V + (size_t)(m + 1) * n;
        std::vector<double> coef(m);
        for (int t = 0; t < nx; ++t) {
            for (int i = 0; i < m; ++i) coef[i] = S[(size_t)i * m + ord[t]];
            ESC_HIP(hipMemcpyAsync(c.hbuf, coef.data(), m * sizeof(double), hipMemcpyHostToDevice, h->stream));
            ESC_HIP(hipMemsetAsync(X + (size_t)t * n, 0, n * sizeof(double), h->stream));
            hipLaunchKernelGGL(k_multiaxpy, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, m, V, (int64_t)n, c.hbuf, 1.0, X + (size_t)t * n);
            ESC_HIP(hipStreamSynchronize(h->stream));
        }
        // ---- final Rayleigh-Ritz on Z = [Q | X]  (orthonormal: X is in the complement of Q)
        const int nz = r + nx;
        double* Z = X + (size_t)nx * n;                                  // contiguous copy [Q | X]
        ESC_HIP(hipMemcpyAsync(Z, c.B, (size_t)r * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        ESC_HIP(hipMemcpyAsync(Z + (size_t)r * n, X, (size_t)nx * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        std::vector<double> M((size_t)nz * nz, 0.0), col(nz);
        for (int j = 0; j < nz; ++j) {
            ESC_CHECK(sapply(c, Z + (size_t)j * n, c.w));
            hipLaunchKernelGGL(k_multidot, dim3(nz), dim3(MSDP_BLOCK), 0, h->stream, n, Z, (int64_t)n, c.w, c.hbuf);
            ESC_HIP(hipMemcpyAsync(col.data(), c.hbuf, nz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            ESC_HIP(hipStreamSynchronize(h->stream));
            for (int i = 0; i < nz; ++i) M[(size_t)i * nz + j] = col[i];
        }
        for (int i = 0; i < nz; ++i) for (int j = i + 1; j < nz; ++j) {
            const double s = 0.5 * (M[(size_t)i * nz + j] + M[(size_t)j * nz + i]);
            M[(size_t)i * nz + j] = s; M[(size_t)j * nz + i] = s;
        }
        std::vector<double> ew, EV;
        jacobi_eig(nz, M, ew, EV);
        std::vector<int> eo(nz);
        for (int i = 0; i < nz; ++i) eo[i] = i;
        std::sort(eo.begin(), eo.end(), [&](int a, int b) { return ew[a] < ew[b]; });
        const int nout = std::min(k, nz);
        std::vector<double> cz(nz);
        for (int t = 0; t < k; ++t) {
            if (t < nout) {
                lam_out[t] = ew[eo[t]];
                for (int i = 0; i < nz; ++i) cz[i] = EV[(size_t)i * nz + eo[t]];
                ESC_HIP(hipMemcpyAsync(c.hbuf, cz.data(), nz * sizeof(double), hipMemcpyHostToDevice, h->stream));
                ESC_HIP(hipMemsetAsync(c.w2, 0, n * sizeof(double), h->stream));
                hipLaunchKernelGGL(k_multiaxpy, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, nz, Z, (int64_t)n, c.hbuf, 1.0, c.w2);
                ESC_HIP(hipMemcpyAsync(V_out + (size_t)t * n, c.w2, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                ESC_HIP(hipStreamSynchronize(h->stream));
            } else {
                lam_out[t] = (nout > 0) ? lam_out[nout - 1] : 0.0;
                memset(V_out + (size_t)t * n, 0, n * sizeof(double));
            }
        }
        if (lmax_out) *lmax_out = std::max(lam_max, ew[eo[nz - 1]]);
        if (iters_out) *iters_out = m;
    }
done:
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(mem);
    return rc;
}
