// msdp_escape.hip -- few-eigenvector saddle escape on the device.
//
// The reference calls eig(full(S)) (O(n^3), dense n x n: 3.2 GB at n = 20000) once per outer
// iteration just to read lambda_min, lambda_max and <= delta bottom eigenvectors of the dual
// slack S = C - diag(z)  (ManiSDP_onlyunitdiag.m:49-51,74-83).  Here the same quantities come
// from deflated Lanczos processes that only need S*v:
//   * at a stationary point of the rank-p problem S*Y = 0 (the Riemannian gradient is S*Y), so
//     span(Y) is the (near-)kernel where the eigenvalues cluster at 0 -- the cluster that made
//     plain ARPACK stall.  Q = orth(Y) is deflated: Lanczos runs on the orthogonal
//     complement, where a negative eigenvalue (escape direction) is an isolated extreme one;
//   * each run is a plain three-term Lanczos recurrence (deflated against Q every step) whose scalars
//     stay on the device; extreme Ritz values remain valid without re-orthogonalisation, so the 10^4
//     steps that certifying lambda_min >= -1e-8 on an ill-conditioned instance (G81) needs are affordable;
//   * negative eigenpairs are peeled off one at a time (the accepted eigenvector joins Q), <= delta runs;
//   * a final Rayleigh-Ritz on [Q | accepted vectors] recouples the blocks, so lambda_min is accurate
//     even when S*Y is only approximately zero.
// All length-n work (SpMV / dense GEMV, block dot products, block axpy) runs in HIP kernels; the host
// keeps the tridiagonal (Sturm bisection + inverse iteration) and the small Rayleigh-Ritz matrix (Jacobi).
#include "msdp_device.h"
#include <math.h>
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

int msdp_dense_nS(int n);
void* msdp_uc_alloc(size_t bytes);                                           // msdp_api.hip: per-process pool of uncached blocks
void msdp_uc_release_pool();
void msdp_host_kits_release();
int msdp_allgather_rows(msdp_handle h, const double* local_rows);          // msdp_api.hip
int msdp_allgather_vec(msdp_handle h, const double* local, double* all, size_t count_per_rank);
// msdp_lanczos.hip: persistent kernel for the recurrence (sparse C, single rank)
size_t msdp_lanczos_slot_bytes();
int msdp_lanczos_persist_ok(msdp_handle h, int nq, const int* full_csr);
int msdp_lanczos_persist_run(msdp_handle h, const double* z, const double* Q, int nq, double* V, double* X, double* dalpha,
                             double* dbeta, unsigned long long* slots, int* err, int m0, int m1,
                             const int* rp, const int* ci, const double* cv);

// msdp_blockeig.hip: Chebyshev-filtered subspace iteration on a b-wide panel (sparse C)
int msdp_blockeig_eligible(msdp_handle h, const double* Mdev, bool w_loc);
int msdp_blockeig_run(msdp_handle h, int n, const int* rp, const int* ci, const double* cv, const double* z, bool own_rows,
                      const double* Ypt, int ld, int p, int k, double tol, int maxdeg, double lmax, double lmax_res, double lmin_est,
                      bool cold, bool use_y, double* lam, double* V_dev, int* degree_out, bool* conv_out, double* err_out,
                      double* lower_out, const double* Mdense);

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
// dense C (n x nS row-major, zero pad columns, nS % 16 == 0): one wave per row, four 1-KB pieces of the row in
// flight per lane (a single dependent accumulator chain left the loads of one piece at a time: 45 us per
// 200-MB product at n = 5000)
// Row shard: `nrows` rows starting at global row `row0` (Cd holds those rows only); z and v have all n entries, w gets the
// nrows local ones.  Unsharded: nrows = n, row0 = 0.
__global__ __launch_bounds__(256) void k_sv_dense(int nrows, int n, int nS, const double* __restrict__ Cd, const double* __restrict__ z,
                                                   const double* __restrict__ v, double* __restrict__ w, int row0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const double* cr = Cd + (int64_t)row * nS;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = 2 * lane;
    for (; j + 384 < n - 1; j += 512) {                          // all four pieces fully inside [0, n)
        const double2 c0 = ld2(cr + j), c1 = ld2(cr + j + 128), c2 = ld2(cr + j + 256), c3 = ld2(cr + j + 384);
        // v may sit at an odd multiple of 8 bytes inside the workspace: 8-byte loads
        a0 = fma(c0.x, v[j], fma(c0.y, v[j + 1], a0));
        a1 = fma(c1.x, v[j + 128], fma(c1.y, v[j + 129], a1));
        a2 = fma(c2.x, v[j + 256], fma(c2.y, v[j + 257], a2));
        a3 = fma(c3.x, v[j + 384], fma(c3.y, v[j + 385], a3));
    }
    for (; j < n; j += 128) {
        const double2 c2 = ld2(cr + j);
        a0 += c2.x * v[j] + ((j + 1 < n) ? c2.y * v[j + 1] : 0.0);
    }
    const double acc = msdp_wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) w[row] = acc - (z ? z[row0 + row] * v[row0 + row] : 0.0);
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
// Ritz-vector assembly x = V*s for thousands of columns: one thread per row walking all columns is a serial chain of
// m dependent FMAs on 79 workgroups (2 ms at m = 5184); here the columns are cut into gridDim.y chunks that write
// partial vectors, summed in chunk order afterwards (deterministic).
__global__ void k_multiaxpy_chunks(int n, int nb, const double* __restrict__ B, int64_t ldb, const double* __restrict__ h,
                                   double* __restrict__ part) {
    const int nch = gridDim.y, ch = blockIdx.y;
    const int c0 = (int)(((int64_t)nb * ch) / nch), c1 = (int)(((int64_t)nb * (ch + 1)) / nch);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int c = c0;
        for (; c + 3 < c1; c += 4) {
            a0 = fma(h[c], B[(int64_t)c * ldb + i], a0);
            a1 = fma(h[c + 1], B[(int64_t)(c + 1) * ldb + i], a1);
            a2 = fma(h[c + 2], B[(int64_t)(c + 2) * ldb + i], a2);
            a3 = fma(h[c + 3], B[(int64_t)(c + 3) * ldb + i], a3);
        }
        for (; c < c1; ++c) a0 = fma(h[c], B[(int64_t)c * ldb + i], a0);
        part[(int64_t)ch * n + i] = (a0 + a1) + (a2 + a3);
    }
}
__global__ void k_sum_chunks(int n, int nch, const double* __restrict__ part, double* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int ch = 0; ch < nch; ++ch) acc += part[(int64_t)ch * n + i];
        dst[i] = acc;
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
// Start vector of an undeflated check run: a hashed random combination of the columns of Y plus 5 % hashed noise.  At a
// near-stationary point span(Y) is the near-kernel of S -- exactly where a lambda_min that the deflated estimate missed
// would live -- so the Krylov space starts rich in the bottom cluster instead of having to amplify 1/sqrt(n) components.
__global__ void k_fill_ycomb(int n, int ld, int p, const double* __restrict__ Y, unsigned seed, double* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int c = 0; c < p; ++c) {
            unsigned x = (unsigned)(c + 1) * 2246822519u ^ seed;
            x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
            acc = fma(Y[(int64_t)i * ld + c], (double)x / 4294967296.0 - 0.5, acc);
        }
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        dst[i] = acc + 0.05 * ((double)x / 4294967296.0 - 0.5);
    }
}

// ---------------------------------------------------------------- host helpers
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

// ---- tridiagonal helpers (host): extreme eigenvalues by Sturm bisection, eigenvector by inverse iteration
static int sturm_count(const std::vector<double>& a, const std::vector<double>& b, int m, double x) {
    // number of eigenvalues of T_m smaller than x (b[i] couples i and i+1)
    int cnt = 0;
    double q = a[0] - x;
    if (q < 0) ++cnt;
    for (int i = 1; i < m; ++i) {
        const double bb = b[i - 1] * b[i - 1];
        q = a[i] - x - bb / (fabs(q) < 1e-300 ? (q < 0 ? -1e-300 : 1e-300) : q);
        if (q < 0) ++cnt;
    }
    return cnt;
}
// Sturm counts at NS shifts in one pass: the recurrence is a chain of dependent divisions (m of them, ~5 ns each), so
// eight independent chains cost little more than one -- the bisection below then cuts its bracket nine-fold per pass
// (15 passes instead of 47 for the 1e-14 bracket of a checkpoint; the host analysis was a third of a long Lanczos run).
#define STURM_NS 8
static void sturm_count_multi(const std::vector<double>& a, const std::vector<double>& b, int m, const double* x, int* cnt) {
    double q[STURM_NS];
    for (int s = 0; s < STURM_NS; ++s) { q[s] = a[0] - x[s]; cnt[s] = q[s] < 0 ? 1 : 0; }
    for (int i = 1; i < m; ++i) {
        const double bb = b[i - 1] * b[i - 1], ai = a[i];
        for (int s = 0; s < STURM_NS; ++s) {
            const double den = fabs(q[s]) < 1e-300 ? (q[s] < 0 ? -1e-300 : 1e-300) : q[s];
            q[s] = ai - x[s] - bb / den;
            cnt[s] += q[s] < 0 ? 1 : 0;
        }
    }
}
static double tri_eig_kth(const std::vector<double>& a, const std::vector<double>& b, int m, int kth, double lo, double hi,
                          double abstol = 0.0) {
    for (int it = 0; it < 200 && hi - lo > 4e-16 * std::max(fabs(lo), fabs(hi)) + 1e-300 + abstol; ++it) {
        if (m < 256) {                                   // short recurrences: plain bisection
            const double mid = 0.5 * (lo + hi);
            if (sturm_count(a, b, m, mid) > kth) hi = mid; else lo = mid;
            continue;
        }
        double x[STURM_NS]; int cnt[STURM_NS];
        const double hstep = (hi - lo) / (STURM_NS + 1);
        for (int s = 0; s < STURM_NS; ++s) x[s] = lo + hstep * (s + 1);
        sturm_count_multi(a, b, m, x, cnt);
        // the k-th eigenvalue lies between the last shift with count <= kth and the first one with count > kth
        double nlo = lo, nhi = hi;
        for (int s = 0; s < STURM_NS; ++s) { if (cnt[s] > kth) { nhi = x[s]; break; } nlo = x[s]; }
        if (!(nhi - nlo < hi - lo)) break;               // the bracket no longer shrinks (rounding)
        lo = nlo; hi = nhi;
    }
    return 0.5 * (lo + hi);
}
// eigenvector of T_m for eigenvalue theta (two steps of inverse iteration with a tiny shift), unit norm
static void tri_eigvec(const std::vector<double>& a, const std::vector<double>& b, int m, double theta, std::vector<double>& s) {
    s.assign(m, 1.0 / sqrt((double)m));
    const double sh = theta - 1e-14 * (fabs(theta) + 1.0);
    std::vector<double> cp(m), dp(m);
    for (int rep = 0; rep < 3; ++rep) {
        // Thomas algorithm on (T - sh I) x = s
        double den = a[0] - sh;
        if (fabs(den) < 1e-300) den = 1e-300;
        cp[0] = (m > 1 ? b[0] : 0.0) / den; dp[0] = s[0] / den;
        for (int i = 1; i < m; ++i) {
            den = a[i] - sh - b[i - 1] * cp[i - 1];
            if (fabs(den) < 1e-300) den = 1e-300;
            cp[i] = (i < m - 1 ? b[i] : 0.0) / den;
            dp[i] = (s[i] - b[i - 1] * dp[i - 1]) / den;
        }
        s[m - 1] = dp[m - 1];
        for (int i = m - 2; i >= 0; --i) s[i] = dp[i] - cp[i] * s[i + 1];
        double nn = 0.0;
        for (int i = 0; i < m; ++i) nn += s[i] * s[i];
        nn = sqrt(nn);
        for (int i = 0; i < m; ++i) s[i] /= nn;
    }
}

// ---- device-side Lanczos step kernels (scalars stay on the device; the host syncs only at checkpoints)
__global__ __launch_bounds__(MSDP_BLOCK) void k_dot1(int n, const double* __restrict__ x, const double* __restrict__ y,
                                                    double* __restrict__ out, int take_sqrt) {
    __shared__ double sh[MSDP_WAVES];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += MSDP_BLOCK) acc = fma(x[i], y[i], acc);
    const double s = msdp_block_sum(acc, sh);
    if (threadIdx.x == 0) *out = take_sqrt ? sqrt(s > 0 ? s : 0.0) : s;
}
// w <- w - alpha*v - beta*vprev   (alpha, beta read from device memory)
__global__ void k_lanczos_update(int n, double* __restrict__ w, const double* __restrict__ v, const double* __restrict__ vprev,
                                 const double* __restrict__ alpha, const double* __restrict__ beta) {
    const double a = *alpha, b = beta ? *beta : 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        w[i] = w[i] - a * v[i] - (vprev ? b * vprev[i] : 0.0);
}
// dst <- w / (*nrm)
__global__ void k_normalize_to(int n, const double* __restrict__ w, const double* __restrict__ nrm, double* __restrict__ dst) {
    const double s = *nrm;
    const double inv = s > 0 ? 1.0 / s : 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = w[i] * inv;
}

// One Lanczos step after w = S*v for vectors that fit one workgroup's LDS (n <= LZS_MAXN): alpha = <w,v>,
// w -= alpha*v + beta*vprev, one classical Gram-Schmidt pass against the nq deflation columns, beta' = |w|,
// vnext = w/beta'.  Replaces k_dot1, k_lanczos_update, k_multidot, k_multiaxpy, k_dot1, k_normalize_to: at
// n = 5000 each of those is a ~4.7 us launch for 40 KB of data (28 of the 57 us of a dense-S step, rocprofv3).
#define LZS_THREADS 1024
#define LZS_MAXN 16384
#define LZS_MAXQ 128
__device__ __forceinline__ double lzs_block_sum(double v, double* red) {
    v = msdp_wave_sum(v);
    __syncthreads();                                   // red may still be read by the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < LZS_THREADS / 64; ++k) s += red[k];
    return s;
}
template <int R>
__global__ __launch_bounds__(LZS_THREADS) void k_lz_step_small(int n, double* __restrict__ w, const double* __restrict__ v,
                                                               const double* __restrict__ vprev, const double* __restrict__ beta_in,
                                                               const double* __restrict__ Q, int nq,
                                                               double* __restrict__ alpha_out, double* __restrict__ beta_out,
                                                               double* __restrict__ vnext) {
    extern __shared__ double ws[];                     // n (used only with deflation columns)
    __shared__ double red[LZS_THREADS / 64];
    __shared__ double hq[LZS_MAXQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // each thread keeps its R elements of w in registers; all loads of a phase are issued together
    double wr[R], vr[R], pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * LZS_THREADS;
        const bool ok = i < n;
        const int ic = ok ? i : 0;
        const double a = w[ic], bq = v[ic], cq = vprev ? vprev[ic] : 0.0;
        wr[r] = ok ? a : 0.0; vr[r] = ok ? bq : 0.0; pr[r] = ok ? cq : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) acc = fma(wr[r], vr[r], acc);
    const double alpha = lzs_block_sum(acc, red);
    const double b = (vprev && beta_in) ? *beta_in : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = wr[r] - alpha * vr[r] - b * pr[r];
    if (nq > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { const int i = tid + r * LZS_THREADS; if (i < n) ws[i] = wr[r]; }
        __syncthreads();
        for (int c0 = 0; c0 < nq; c0 += LZS_MAXQ) {    // deflation columns in groups of LZS_MAXQ, one wave per column
            const int nc = min(LZS_MAXQ, nq - c0);
            for (int c = wave; c < nc; c += LZS_THREADS / 64) {
                const double* qc = Q + (int64_t)(c0 + c) * n;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                int i = lane;
                for (; i + 192 < n; i += 256) {
                    a0 = fma(qc[i], ws[i], a0); a1 = fma(qc[i + 64], ws[i + 64], a1);
                    a2 = fma(qc[i + 128], ws[i + 128], a2); a3 = fma(qc[i + 192], ws[i + 192], a3);
                }
                for (; i < n; i += 64) a0 = fma(qc[i], ws[i], a0);
                const double a = msdp_wave_sum((a0 + a1) + (a2 + a3));
                if (lane == 0) hq[c] = a;
            }
            __syncthreads();
            for (int c = 0; c < nc; ++c) {
                const double hc = hq[c];
                const double* qc = Q + (int64_t)(c0 + c) * n;
#pragma unroll
                for (int r = 0; r < R; ++r) { const int i = tid + r * LZS_THREADS; wr[r] -= hc * qc[i < n ? i : 0]; }
            }
            if (c0 + LZS_MAXQ < nq) {                  // another group follows: it needs the updated vector in LDS
#pragma unroll
                for (int r = 0; r < R; ++r) { const int i = tid + r * LZS_THREADS; if (i < n) ws[i] = wr[r]; }
            }
            __syncthreads();
        }
    }
    acc = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = tid + r * LZS_THREADS; acc = fma(i < n ? wr[r] : 0.0, wr[r], acc); }
    const double nn = lzs_block_sum(acc, red);
    const double beta = sqrt(nn > 0.0 ? nn : 0.0);
    const double inv = beta > 0.0 ? 1.0 / beta : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = tid + r * LZS_THREADS; if (i < n) { w[i] = wr[r]; vnext[i] = wr[r] * inv; } }
    if (tid == 0) { *alpha_out = alpha; *beta_out = beta; }
}

struct EscCtx {
    msdp_handle h;
    int n;
    const double* z;
    double* hbuf;
    const double* M;     // explicit dense S (n x nS, device) for the affine kinds; nullptr: S = C - diag(z)
    double* X;           // persistent Lanczos: exchange buffers (4 n), grid-sync slots, error flag
    unsigned long long* slots;
    int* err;
    double host_analysis_s = 0.0;   // time the host spent analysing T_m at the checkpoints (esc_debug statistics)
    const double* Ypt = nullptr;    // all rows of the resident point (n x ld), for the escape_start_y start vector
    int ld = 0, p = 0;
    const int* rp = nullptr; const int* ci = nullptr; const double* cv = nullptr;   // CSR of C (all rows)
    bool replicated_csr = false;    // rp/ci/cv are this rank's full copy of a row-sharded C (the handle's own CSR holds its rows only)
    // pre-sharded dense C: this rank multiplies ITS rows (w_loc), the ranks all-gather the pieces (w_all: nranks*cap), and
    // every rank continues the same recurrence on the same full-length vectors
    double* w_loc = nullptr; double* w_all = nullptr; int cap = 0;
};

static int sapply(EscCtx& c, const double* v, double* w) {
    msdp_handle h = c.h;
    const Dev& d = h->d;
    if (c.w_loc) {
        if (d.n_loc > 0)
            hipLaunchKernelGGL(k_sv_dense, dim3((d.n_loc + 3) / 4), dim3(256), 0, h->stream, d.n_loc, c.n, msdp_dense_nS(c.n), (const double*)d.Cd, c.z, v, c.w_loc, d.row0);
        HIPCHK(hipGetLastError());
        int rc = msdp_allgather_vec(h, c.w_loc, c.w_all, (size_t)c.cap);
        if (rc) return rc;
        HIPCHK(msdp_memcpy_async(w, c.w_all, (size_t)c.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        return 0;
    }
    if (c.M)
        hipLaunchKernelGGL(k_sv_dense, dim3((c.n + 3) / 4), dim3(256), 0, h->stream, c.n, c.n, msdp_dense_nS(c.n), c.M, (const double*)nullptr, v, w, 0);
    else if (d.costkind == COST_SPARSE)
        hipLaunchKernelGGL(k_sv_sparse, dim3((c.n + 255) / 256), dim3(256), 0, h->stream, c.n, c.rp, c.ci, c.cv, c.z, v, w);
    else
        hipLaunchKernelGGL(k_sv_dense, dim3((c.n + 3) / 4), dim3(256), 0, h->stream, c.n, c.n, msdp_dense_nS(c.n), (const double*)d.Cd, c.z, v, w, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// w <- w - Q (Q' w) for the nq deflation columns (one classical Gram-Schmidt pass, repeated `passes` times)
static int deflate(EscCtx& c, const double* Q, int nq, double* w, int passes) {
    msdp_handle h = c.h;
    for (int pass = 0; pass < passes && nq > 0; ++pass) {
        hipLaunchKernelGGL(k_multidot, dim3(nq), dim3(MSDP_BLOCK), 0, h->stream, c.n, Q, (int64_t)c.n, w, c.hbuf);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_multiaxpy, dim3((c.n + 255) / 256), dim3(256), 0, h->stream, c.n, nq, Q, (int64_t)c.n, c.hbuf, -1.0, w);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

static int dev_norm(EscCtx& c, const double* w, double* out) {
    msdp_handle h = c.h;
    hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, c.n, w, w, c.hbuf, 1);
    HIPCHK(hipGetLastError());
    double v = 0.0;
    HIPCHK(msdp_memcpy_async(&v, c.hbuf, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = v;
    return 0;
}

// Advance the recurrence from step m to step m1 (columns m+1..m1 of V, alpha[m..m1-1], beta[m+1..m1]).
struct LzMode { bool persist, fused_small; };
static LzMode lanczos_mode(EscCtx& c, int nq) {
    LzMode md;
    md.persist = !c.M && !c.w_loc && c.slots && msdp_lanczos_persist_ok(c.h, nq, c.replicated_csr ? c.rp : nullptr);
    md.fused_small = !md.persist && c.n <= LZS_MAXN;
    if (md.fused_small) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)k_lz_step_small<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * LZS_THREADS * (int)sizeof(double));
            (void)hipFuncSetAttribute((const void*)k_lz_step_small<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LZS_MAXN * (int)sizeof(double));
            attr_set = true;
        }
    }
    return md;
}
static int lanczos_advance(EscCtx& c, const LzMode& md, const double* Q, int nq, double* V, double* w, double* dalpha, double* dbeta,
                           int& m, int m1) {
    msdp_handle h = c.h;
    const int n = c.n;
    const dim3 gr((n + 255) / 256), bl(256);
    int rc;
    if (md.persist) {
        // all steps up to the next checkpoint in one launch (msdp_lanczos.hip)
        if ((rc = msdp_lanczos_persist_run(h, c.z, Q, nq, V, c.X, dalpha, dbeta, c.slots, c.err, m, m1, c.rp, c.ci, c.cv))) return rc;
        m = m1;
        return 0;
    }
    while (m < m1) {
        double* vj = V + (size_t)m * n;
        if ((rc = sapply(c, vj, w))) return rc;
        if (md.fused_small) {
            const size_t lds = nq > 0 ? (size_t)n * sizeof(double) : 0;
            const double* vp = m > 0 ? (const double*)(vj - n) : (const double*)nullptr;
            const double* bp = m > 0 ? (const double*)(dbeta + m) : (const double*)nullptr;
            if (n <= 4 * LZS_THREADS)
                hipLaunchKernelGGL(k_lz_step_small<4>, dim3(1), dim3(LZS_THREADS), lds, h->stream, n, w, (const double*)vj, vp, bp, Q, nq, dalpha + m, dbeta + m + 1, vj + n);
            else if (n <= 8 * LZS_THREADS)
                hipLaunchKernelGGL(k_lz_step_small<8>, dim3(1), dim3(LZS_THREADS), lds, h->stream, n, w, (const double*)vj, vp, bp, Q, nq, dalpha + m, dbeta + m + 1, vj + n);
            else
                hipLaunchKernelGGL(k_lz_step_small<16>, dim3(1), dim3(LZS_THREADS), lds, h->stream, n, w, (const double*)vj, vp, bp, Q, nq, dalpha + m, dbeta + m + 1, vj + n);
            HIPCHK(hipGetLastError());
        } else {
            hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, n, w, vj, dalpha + m, 0);
            hipLaunchKernelGGL(k_lanczos_update, gr, bl, 0, h->stream, n, w, vj, m > 0 ? vj - n : (const double*)nullptr,
                               dalpha + m, m > 0 ? dbeta + m : (const double*)nullptr);
            HIPCHK(hipGetLastError());
            if ((rc = deflate(c, Q, nq, w, 1))) return rc;
            hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, n, w, w, dbeta + m + 1, 1);
            hipLaunchKernelGGL(k_normalize_to, gr, bl, 0, h->stream, n, w, dbeta + m + 1, vj + n);
            HIPCHK(hipGetLastError());
        }
        ++m;
    }
    return 0;
}

// Largest eigenvalue of S by a short undeflated Lanczos run (the block eigen-solver of msdp_blockeig.hip needs the upper
// edge of the spectrum for its filter, the AL loop needs lambda_max for dinf).  Start: `warm` (the top vector of the
// previous call: S changes by a small diagonal between outer iterations) plus hashed noise, or noise alone.  Stops when the
// top Ritz pair has residual <= tol*scale.  Outputs: lambda_max, its residual bound, the smallest Ritz value of the run (a
// rough UPPER estimate of lambda_min: it only normalises the filter) and the top Ritz vector (device, n).
__global__ void k_add_hash(int n, unsigned seed, double amp, double* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        dst[i] += amp * ((double)x / 4294967296.0 - 0.5);
    }
}
static int lanczos_top(EscCtx& c, double* V, double* w, double* dalpha, double* dbeta, int maxit, double tol, unsigned seed,
                       const double* warm, double* lmax_out, double* res_out, double* lmin_out, double* top_out, int* m_out) {
    msdp_handle h = c.h;
    const int n = c.n;
    const dim3 gr((n + 255) / 256), bl(256);
    int rc;
    if (warm) {
        HIPCHK(msdp_memcpy_async(w, warm, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        // the stored vector has unit norm: entries ~ 1/sqrt(n); 10 % of that as noise keeps every eigen-direction alive
        hipLaunchKernelGGL(k_add_hash, gr, bl, 0, h->stream, n, seed, 0.35 / sqrt((double)n), w);
    } else {
        hipLaunchKernelGGL(k_fill_hash, gr, bl, 0, h->stream, n, seed, w);
    }
    hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, n, w, w, dbeta, 1);
    hipLaunchKernelGGL(k_normalize_to, gr, bl, 0, h->stream, n, w, dbeta, V);
    HIPCHK(hipGetLastError());
    const LzMode md = lanczos_mode(c, 0);
    std::vector<double> a, b, s, off;
    int m = 0, next_check = warm ? 16 : 64;
    double lmax = 0.0, lmin = 0.0, res = 1e300;
    while (m < maxit) {
        if ((rc = lanczos_advance(c, md, nullptr, 0, V, w, dalpha, dbeta, m, std::min(next_check, maxit)))) return rc;
        a.resize(m); b.resize(m + 1);
        HIPCHK(msdp_memcpy_async(a.data(), dalpha, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(msdp_memcpy_async(b.data(), dbeta, (m + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (md.persist) {
            int perr = 0;
            HIPCHK(msdp_memcpy(&perr, c.err, sizeof(int), hipMemcpyDeviceToHost));
            if (perr) { msdp_set_error("persistent Lanczos: grid synchronisation timed out"); return MSDP_EHIP; }
        }
        off.assign(m, 0.0);
        for (int i = 0; i + 1 < m; ++i) off[i] = b[i + 1];
        double glo = 1e300, ghi = -1e300;
        for (int i = 0; i < m; ++i) {
            const double rad = (i > 0 ? fabs(off[i - 1]) : 0.0) + (i + 1 < m ? fabs(off[i]) : 0.0);
            glo = std::min(glo, a[i] - rad); ghi = std::max(ghi, a[i] + rad);
        }
        lmax = tri_eig_kth(a, off, m, m - 1, glo, ghi);
        lmin = tri_eig_kth(a, off, m, 0, glo, ghi, 1e-10 * (fabs(glo) + fabs(ghi)));
        tri_eigvec(a, off, m, lmax, s);
        res = fabs(b[m] * s[m - 1]);
        const double scale = std::max(fabs(lmax), fabs(lmin)) + 1e-300;
        if (res <= tol * scale || b[m] <= 1e-14 * scale) break;
        next_check = std::min(maxit, 2 * m);
    }
    // top Ritz vector (warm start of the next call)
    if (top_out) {
        double* sdev = nullptr;
        HIPCHK(hipMalloc((void**)&sdev, (size_t)m * sizeof(double)));
        hipError_t e = msdp_memcpy_async(sdev, s.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(top_out, 0, (size_t)n * sizeof(double), h->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_multiaxpy, gr, bl, 0, h->stream, n, m, V, (int64_t)n, sdev, 1.0, top_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(sdev);
        if (e != hipSuccess) { msdp_set_error("escape: top Ritz vector assembly failed: %s", hipGetErrorString(e)); return MSDP_EHIP; }
    }
    *lmax_out = lmax; *res_out = res; *lmin_out = lmin; *m_out = m;
    return 0;
}

// Plain (three-term) Lanczos on the complement of the nq columns of Q: smallest Ritz pair and the largest
// Ritz value.  No re-orthogonalisation against earlier Lanczos vectors -- extreme Ritz values stay valid
// (ghosts only duplicate converged ones), which is what makes 10^4 steps affordable; the Lanczos vectors are
// kept (n*m doubles of HBM) so the Ritz vector is one block axpy.  Stops when the smallest Ritz pair has
// residual <= tol*scale, or when [theta - residual] is already above -tol*scale (certified non-negative).
static int lanczos_smallest(EscCtx& c, const double* Q, int nq, double* V /* maxit+1 columns */, double* w,
                            double* dalpha, double* dbeta, int maxit, double tol, unsigned seed,
                            double* theta_out, double* res_out, double* lmax_out, double* x_out, int* m_out,
                            int kwant = 1, int* nacc_out = nullptr, double* thetas_out = nullptr,
                            double* xstart = nullptr, bool* have_xstart = nullptr, bool* conv_out = nullptr) {
    msdp_handle h = c.h;
    const int n = c.n;
    const dim3 gr((n + 255) / 256), bl(256);
    int rc;
    // start vector: the sum of the still unconverged negative Ritz vectors of the previous run when there is one
    // (the Krylov space then starts with good approximations of exactly the directions that are still wanted),
    // else a hashed pseudo-random vector
    bool warm = false;
    if (xstart && have_xstart && *have_xstart) {
        HIPCHK(msdp_memcpy_async(w, xstart, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if ((rc = deflate(c, Q, nq, w, 2))) return rc;
        double nw = 0.0;
        if ((rc = dev_norm(c, w, &nw))) return rc;
        warm = nw > 1e-6;
        *have_xstart = false;
    }
    if (!warm) {
        if (h->tune.escape_start_y && c.Ypt && nq == 0)
            hipLaunchKernelGGL(k_fill_ycomb, gr, bl, 0, h->stream, n, c.ld, c.p, c.Ypt, seed, w);
        else
            hipLaunchKernelGGL(k_fill_hash, gr, bl, 0, h->stream, n, seed, w);
        if ((rc = deflate(c, Q, nq, w, 2))) return rc;
    }
    hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, n, w, w, dbeta, 1);
    hipLaunchKernelGGL(k_normalize_to, gr, bl, 0, h->stream, n, w, dbeta, V);
    HIPCHK(hipGetLastError());
    std::vector<double> a, b, s;
    int m = 0, next_check = 32;
    double theta = 0.0, res = 1e300, lmax = 0.0;
    bool converged = false;          // one of the three stop tests passed (else the run ended at maxit: theta is only an upper bound)
    const LzMode md = lanczos_mode(c, nq);
    const bool persist = md.persist;
    while (m < maxit) {
        if ((rc = lanczos_advance(c, md, Q, nq, V, w, dalpha, dbeta, m, std::min(next_check, maxit)))) return rc;
        if (m == next_check || m == maxit) {
            a.resize(m); b.resize(m + 1);
            HIPCHK(msdp_memcpy_async(a.data(), dalpha, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(msdp_memcpy_async(b.data(), dbeta, (m + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (persist) {
                int perr = 0;
                HIPCHK(msdp_memcpy(&perr, c.err, sizeof(int), hipMemcpyDeviceToHost));
                if (perr) { msdp_set_error("persistent Lanczos: grid synchronisation timed out"); return MSDP_EHIP; }
            }
            const auto t_an0 = std::chrono::steady_clock::now();
            struct AnTimer { std::chrono::steady_clock::time_point t0; double* acc; ~AnTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } an_timer{t_an0, &c.host_analysis_s};
            // T_m: diagonal a[0..m-1], couplings b[1..m-1]; b[m] closes the residual
            std::vector<double> off(m);
            for (int i = 0; i + 1 < m; ++i) off[i] = b[i + 1];
            double glo = 1e300, ghi = -1e300;
            for (int i = 0; i < m; ++i) {
                const double rad = (i > 0 ? fabs(off[i - 1]) : 0.0) + (i + 1 < m ? fabs(off[i]) : 0.0);
                glo = std::min(glo, a[i] - rad); ghi = std::max(ghi, a[i] + rad);
            }
            // Host cost of a checkpoint is ~m divisions per Sturm count: the largest Ritz value is converged after
            // a few hundred steps and only sets the scale, so it is refreshed up to m = 1024 and once more at the
            // end; the smallest one is bisected to 1e-14*scale (the residual test needs 1e-9*scale).
            if (m <= 1024 || lmax == 0.0) lmax = tri_eig_kth(a, off, m, m - 1, glo, ghi);
            const double scale0 = std::max(fabs(glo), fabs(lmax)) + 1e-300;
            theta = tri_eig_kth(a, off, m, 0, glo, ghi, 1e-14 * scale0);
            tri_eigvec(a, off, m, theta, s);
            res = fabs(b[m] * s[m - 1]);
            const double scale = std::max(fabs(theta), fabs(lmax)) + 1e-300;
            const bool breakdown = b[m] <= 1e-14 * scale;
            if (res <= tol * scale || theta - res > -tol * scale || breakdown) {
                if (m > 1024) lmax = tri_eig_kth(a, off, m, m - 1, glo, ghi);
                converged = true;
                break;
            }
            // doubling up to 1024 steps, then x1.5, x1.25 from 2048 on: a late checkpoint wastes steps, an early one costs a host analysis
            // (x1.125 from 8192 on: with the eight-shift bisection a checkpoint at m = 30000 costs 1.5 ms, a late stop 4 us per step)
            next_check = std::min(maxit, m < 1024 ? 2 * m : (m < 2048 ? m + m / 2 : (m < 8192 ? m + m / 4 : m + m / 8)));
        }
    }
    // Ritz vector x = V s
    double* sdev = nullptr;
    HIPCHK(hipMalloc((void**)&sdev, (size_t)m * sizeof(double)));
    auto assemble = [&](const std::vector<double>& sv, double* dst) -> int {
        hipError_t e = msdp_memcpy_async(sdev, sv.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, h->stream);
        const int NCH = 32;
        if (e == hipSuccess && m >= 256 && m + 2 + NCH <= maxit + 2) {
            double* part = V + (size_t)(m + 2) * n;              // unused tail of the Lanczos basis as scratch
            hipLaunchKernelGGL(k_multiaxpy_chunks, dim3(gr.x, NCH), bl, 0, h->stream, n, m, V, (int64_t)n, sdev, part);
            hipLaunchKernelGGL(k_sum_chunks, gr, bl, 0, h->stream, n, NCH, part, dst);
            e = hipGetLastError();
        } else {
            if (e == hipSuccess) e = hipMemsetAsync(dst, 0, (size_t)n * sizeof(double), h->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_multiaxpy, gr, bl, 0, h->stream, n, m, V, (int64_t)n, sdev, 1.0, dst);
                e = hipGetLastError();
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { msdp_set_error("escape: Ritz vector assembly failed: %s", hipGetErrorString(e)); return MSDP_EHIP; }
        return 0;
    };
    int nacc = 1;
    rc = assemble(s, x_out);
    // clean up the Ritz vector: project out Q once more and normalise
    if (!rc) rc = deflate(c, Q, nq, x_out, 1);
    double nx = 0.0;
    if (!rc) rc = dev_norm(c, x_out, &nx);
    if (!rc && nx > 0) hipLaunchKernelGGL(k_scale_copy, gr, bl, 0, h->stream, n, x_out, 1.0 / nx, x_out);
    if (thetas_out) thetas_out[0] = theta;
    // ---- further negative Ritz pairs of the SAME Krylov space.  On G81 the first outer iteration has 8 negative
    // eigenvalues 1e-6 apart: peeling them one run at a time cost 8 x 8192 steps, while the run that resolves
    // the smallest one has (nearly) resolved its neighbours too.  Candidates are taken in increasing order;
    // ghost copies (Ritz values equal to 1e-10*scale) are skipped, a candidate must have a converged Ritz
    // estimate, survive orthogonalisation against everything accepted so far and pass a TRUE residual check
    // |S x - (x'Sx) x| <= 1e-6*scale; the first failure ends the extraction (the next run picks up from there).
    if (!rc && kwant > 1 && m > 1) {
        const double scale = std::max(fabs(theta), fabs(lmax)) + 1e-300;
        if (theta < -tol * scale) {
            std::vector<double> off(m), s2;
            for (int i = 0; i + 1 < m; ++i) off[i] = b[i + 1];
            double last = theta;
            for (int idx = 1; idx < m && nacc < kwant && !rc; ++idx) {
                const double th = tri_eig_kth(a, off, m, idx, last, 0.0);
                if (!(th < -tol * scale)) break;
                if (th - last <= 1e-10 * scale) continue;                 // ghost copy
                last = th;
                tri_eigvec(a, off, m, th, s2);
                if (fabs(b[m] * s2[m - 1]) > 1e-7 * scale) break;          // not converged in this Krylov space
                double* xk = x_out + (size_t)nacc * n;
                if ((rc = assemble(s2, xk))) break;
                double n0, n1;
                if ((rc = dev_norm(c, xk, &n0))) break;
                if ((rc = deflate(c, Q, nq + nacc, xk, 2))) break;          // Q and the vectors accepted in this call are contiguous
                if ((rc = dev_norm(c, xk, &n1))) break;
                if (!(n1 > 0.5 * n0) || !(n1 > 0)) continue;                // a direction we already have
                hipLaunchKernelGGL(k_scale_copy, gr, bl, 0, h->stream, n, xk, 1.0 / n1, xk);
                // true residual
                if ((rc = sapply(c, xk, w))) break;
                hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, n, w, xk, c.hbuf + 8, 0);
                hipLaunchKernelGGL(k_lanczos_update, gr, bl, 0, h->stream, n, w, xk, (const double*)nullptr, c.hbuf + 8, (const double*)nullptr);
                double rq = 0.0, rn = 0.0;
                if ((rc = dev_norm(c, w, &rn))) break;
                if (msdp_memcpy(&rq, c.hbuf + 8, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { msdp_set_error("escape: memcpy failed"); rc = MSDP_EHIP; break; }
                if (!(rn <= 1e-6 * scale) || !(rq < -tol * scale)) break;
                if (thetas_out) thetas_out[nacc] = rq;
                ++nacc;
            }
        }
    }
    // warm start for the next run: sum of the Ritz vectors of the next distinct negative Ritz values that were NOT
    // accepted (V*s is linear in s: one block axpy)
    if (!rc && xstart && have_xstart && kwant > nacc && m > 1) {
        const double scale = std::max(fabs(theta), fabs(lmax)) + 1e-300;
        if (theta < -tol * scale) {
            std::vector<double> off(m), s2, ssum(m, 0.0);
            for (int i = 0; i + 1 < m; ++i) off[i] = b[i + 1];
            double last = theta;
            int distinct = 1, taken = 0;
            for (int idx = 1; idx < m && taken < kwant - nacc && idx < 64 * kwant; ++idx) {
                const double th = tri_eig_kth(a, off, m, idx, last, 0.0);
                if (!(th < -tol * scale)) break;
                if (th - last <= 1e-10 * scale) continue;
                last = th;
                ++distinct;
                if (distinct <= nacc) continue;                     // already accepted in this run
                tri_eigvec(a, off, m, th, s2);
                for (int i = 0; i < m; ++i) ssum[i] += s2[i];
                ++taken;
            }
            if (taken > 0) {
                rc = assemble(ssum, xstart);
                if (!rc) *have_xstart = true;
            }
        }
    }
    (void)hipFree(sdev);
    if (rc) return rc;
    if (nacc_out) *nacc_out = nacc;
    if (conv_out) *conv_out = converged;
    *theta_out = theta; *res_out = res; *lmax_out = lmax; *m_out = m;
    return 0;
}

// One parked workspace per process: msdp_destroy hands the escape workspace of a handle here instead of freeing it, the
// next handle that needs one of at most that size takes it over (the device is the process's current one; a workspace
// parked on another device is released instead).
#include <mutex>
static std::mutex g_ws_mutex;
static double* g_ws_ptr = nullptr;
static size_t g_ws_cap = 0;
static int g_ws_dev = -1;
void msdp_escape_workspace_park(double* ptr, size_t cap_doubles) {
    if (!ptr) return;
    int dev = -1;
    (void)hipGetDevice(&dev);
    // Nothing larger than an eighth of the device memory (36 GB of 288) is kept beyond the life of its handle: other users of
    // the GPU in the same process (torch in bench.py, MATLAB's own gpuArrays) must not find the memory gone
    {
        size_t freeb = 0, totb = 0;
        if (hipMemGetInfo(&freeb, &totb) != hipSuccess) { (void)hipGetLastError(); totb = 0; }
        if (totb && cap_doubles * sizeof(double) > totb / 8) { (void)hipFree(ptr); return; }
    }
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (g_ws_ptr && g_ws_cap >= cap_doubles && g_ws_dev == dev) { (void)hipFree(ptr); return; }    // keep the larger one
    if (g_ws_ptr) (void)hipFree(g_ws_ptr);
    g_ws_ptr = ptr; g_ws_cap = cap_doubles; g_ws_dev = dev;
}
extern "C" int msdp_release_cache(void) {
    msdp_uc_release_pool();
    msdp_xfer_release();                                      // the pinned staging buffer of msdp_xfer.hip
    msdp_host_kits_release();                                 // cached streams + pinned control blocks (msdp_api.hip)
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (g_ws_ptr) (void)hipFree(g_ws_ptr);
    g_ws_ptr = nullptr; g_ws_cap = 0; g_ws_dev = -1;
    return 0;
}
double* msdp_escape_workspace_take(size_t need_doubles, size_t* cap_out) {
    int dev = -1;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (!g_ws_ptr || g_ws_dev != dev || g_ws_cap < need_doubles) return nullptr;
    double* p = g_ws_ptr;
    *cap_out = g_ws_cap;
    g_ws_ptr = nullptr; g_ws_cap = 0; g_ws_dev = -1;
    return p;
}

static int escape_impl_once(msdp_handle h, int k, double tol, int maxit, double* lam_out, double* V_out, double* lmax_out,
                     int* iters_out, const double* Mdev) {
    Dev& d = h->d;
    if (!Mdev && h->kind != MSDP_KIND_ONLYUNITDIAG) { msdp_set_error("escape_eigs: affine handles pass S explicitly (msdp_escape_eigs_matrix)"); return MSDP_EUNSUPPORTED; }
    // Row-sharded handles: the escape on an explicit dense S (affine kinds) runs replicated -- every rank holds S and all
    // rows of the point (yfull, refreshed by msdp_al_dual) and finds the same vectors; S = C - diag(z) is not sharded yet
    const bool rep_rows = (h->nranks != 1 || h->use_comm) && Mdev && h->yfull[h->h_ctl->cur];
    // Row-sharded onlyunitdiag with sparse C: replicated as well -- every rank keeps a full copy of C's CSR arrays for this
    // purpose (1.2 MB for G81), gathers z and the rows of Y, and runs the same Lanczos recurrence -- a single-GPU computation
    // on every rank, so the persistent kernels of msdp_lanczos.hip apply with the replicated CSR in place of the handle's own
    // (a Lanczos process with sharded vectors and an all-gather per step is the follow-up)
    // (taken with ANY communicator, also of size 1, so that one GPU exercises the gathers and the replicated copy)
    const bool rep_sparse = h->use_comm && !Mdev && d.costkind == COST_SPARSE && !h->h_rowptr.empty();
    // Pre-sharded dense C (config 5): the matrix-vector product is sharded -- every rank multiplies its rows of C - diag(z)
    // with the full vector and one all-gather of n doubles per Lanczos step puts the product back together;
    // the recurrence itself (dots, axpys, the stored basis, the host analysis) runs replicated on identical numbers
    const bool shard_dense = h->use_comm && !Mdev && d.costkind == COST_DENSE && h->presharded;
    if (h->nranks != 1 && !rep_rows && !rep_sparse && !shard_dense) { msdp_set_error("escape_eigs: row-sharded handles need a communicator (sparse or pre-sharded dense C) or an explicit S"); return MSDP_EUNSUPPORTED; }
    const size_t shard_cap = (size_t)((d.n + h->nranks - 1) / h->nranks);
    if (shard_dense) {
        if (!h->esc_z) {
            // [z of all rows | this rank's rows of the product | the gathered product]
            if (hipMalloc((void**)&h->esc_z, (2 * shard_cap * h->nranks + shard_cap) * sizeof(double)) != hipSuccess) { msdp_set_error("escape_eigs: allocation of the gather buffers failed"); return MSDP_ENOMEM; }
        }
        int rcg = msdp_allgather_vec(h, d.eG[h->h_ctl->cur], h->esc_z, shard_cap);
        if (!rcg) rcg = msdp_allgather_rows(h, d.Y[h->h_ctl->cur]);
        if (rcg) return rcg;
    }
    if (rep_sparse) {
        const size_t nnz = h->h_cval.size(), cap = (size_t)((d.n + h->nranks - 1) / h->nranks);
        if (!h->esc_rp) {
            if (hipMalloc((void**)&h->esc_rp, ((size_t)d.n + 1) * sizeof(int)) != hipSuccess || hipMalloc((void**)&h->esc_ci, (nnz ? nnz : 1) * sizeof(int)) != hipSuccess ||
                hipMalloc((void**)&h->esc_cv, (nnz ? nnz : 1) * sizeof(double)) != hipSuccess || hipMalloc((void**)&h->esc_z, cap * h->nranks * sizeof(double)) != hipSuccess) {
                msdp_set_error("escape_eigs: allocation of the replicated copy of C failed"); return MSDP_ENOMEM;
            }
            HIPCHK(msdp_memcpy(h->esc_rp, h->h_rowptr.data(), ((size_t)d.n + 1) * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(msdp_memcpy(h->esc_ci, h->h_colind.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(msdp_memcpy(h->esc_cv, h->h_cval.data(), nnz * sizeof(double), hipMemcpyHostToDevice));
        }
        int rcg = msdp_allgather_vec(h, d.eG[h->h_ctl->cur], h->esc_z, cap);
        if (!rcg) rcg = msdp_allgather_rows(h, d.Y[h->h_ctl->cur]);        // all rows of the point in the gather buffer
        if (rcg) return rcg;
    }
    if (k < 1) { msdp_set_error("escape_eigs: k >= 1"); return MSDP_EINVAL; }
    const int n = d.n, p = d.p;
    if (maxit < 8) maxit = 8;
    // Sparse C: the block eigen-solver does the work (msdp_blockeig.hip); the Lanczos machinery below then only serves its
    // short lambda_max run, and `maxit` is the budget of filter steps
    const bool use_block = msdp_blockeig_eligible(h, Mdev, shard_dense) != 0;
    const int maxdeg = maxit;
    if (use_block && maxit > 2048) maxit = 2048;
    h->esc_method_last = use_block ? 1 : 0;
    // Without re-orthogonalisation the process may need more than n steps for the extreme pairs (ghost copies use
    // up steps); small matrices are cheap, so allow 4 n there
    if (maxit > 4 * n) maxit = 4 * n;
    // keep the stored Lanczos basis below ~24 GB, or below half of what the device has free where that is more (288 GB of
    // HBM: at n = 80 000 the 24-GB bound would end the independent lambda_min run after 37 500 steps, unconverged)
    double basis_bytes = 24.0e9;
    {
        size_t freeb = 0, totb = 0;
        if (hipMemGetInfo(&freeb, &totb) == hipSuccess) {
            size_t parked = 0;
            { std::lock_guard<std::mutex> lk(g_ws_mutex); parked = g_ws_cap * sizeof(double); }
            const double avail = (double)freeb + (double)h->esc_cap * sizeof(double) + (double)parked;    // what this call could hold
            if (0.5 * avail > basis_bytes) basis_bytes = 0.5 * avail;
        } else (void)hipGetLastError();
    }
    const int64_t cap = (int64_t)(basis_bytes / 8.0 / n);
    if (maxit > cap) maxit = (int)std::max<int64_t>(64, cap);
    const int cur = h->h_ctl->cur;
    EscCtx c;
    c.h = h; c.n = n; c.z = Mdev ? nullptr : ((rep_sparse || shard_dense) ? (const double*)h->esc_z : (const double*)d.eG[cur]); c.M = Mdev;
    if (shard_dense) { c.cap = (int)shard_cap; c.w_loc = h->esc_z + shard_cap * h->nranks; c.w_all = c.w_loc + shard_cap; }
    c.rp = rep_sparse ? h->esc_rp : d.rowptr; c.ci = rep_sparse ? h->esc_ci : d.colind; c.cv = rep_sparse ? h->esc_cv : d.cval;
    c.replicated_csr = rep_sparse;
    c.Ypt = (rep_sparse || shard_dense) ? (const double*)h->full_buf
                       : (((h->nranks != 1 || h->use_comm) && h->yfull[cur]) ? (const double*)h->yfull[cur] : (const double*)d.Y[cur]);
    c.ld = d.ld; c.p = d.p;
    const int qcap = p + k + 1;
    double* mem = nullptr;
    const size_t slot_doubles = msdp_lanczos_slot_bytes() / sizeof(double);
    // [prev | Q | V | Z | w | alpha | beta | hbuf | X | slots | err]; prev (n doubles) = warm start kept between calls;
    // X = 4 n doubles: the exchange buffers of the persistent Lanczos kernels (msdp_lanczos.hip)
    const size_t total = (size_t)n + (size_t)(qcap + maxit + 2 + qcap) * n + (size_t)n + 2 * (size_t)(maxit + 2) + 4096 + 4 * (size_t)n + slot_doubles + 16;
    if (h->esc_cap < total) {
        // grow with head room for 16 more factor columns: the factor width changes every outer iteration and a
        // reallocation of this size stalls the stream for ~0.1 s
        size_t want = total + (size_t)32 * n;
        double* nm = nullptr;
        (void)hipStreamSynchronize(h->stream);
        // a workspace parked by a destroyed handle of this process (same device) is taken over when it is large enough:
        // allocating the 10-GB Lanczos basis of a G81-sized problem costs 0.05-0.5 s, once per process instead of per solve
        size_t got = 0;
        nm = msdp_escape_workspace_take(total, &got);
        hipError_t me = hipSuccess;
        if (nm) want = got;
        else {
            me = hipMalloc((void**)&nm, want * sizeof(double));
            if (me != hipSuccess) { (void)hipGetLastError(); me = hipMalloc((void**)&nm, total * sizeof(double)); if (me == hipSuccess) want = total; }
            if (me != hipSuccess) { msdp_set_error("escape_eigs: workspace allocation (%zu MB) failed", total * 8 >> 20); return MSDP_ENOMEM; }
        }
        if (h->esc_mem && h->esc_prev_n == n) (void)msdp_memcpy(nm, h->esc_mem, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice);
        else h->esc_prev_n = 0;
        if (h->esc_mem) (void)hipFree(h->esc_mem);
        h->esc_mem = nm;
        h->esc_cap = want;
    }
    mem = h->esc_mem + n;
    h->esc_prev = h->esc_mem;
    double* Q = mem;                                   // deflation set: orth(Y) then accepted eigenvectors
    double* V = Q + (size_t)qcap * n;                   // Lanczos vectors
    double* Z = V + (size_t)(maxit + 2) * n;            // Rayleigh-Ritz basis copy
    double* w = Z + (size_t)qcap * n;
    double* dalpha = w + n;
    double* dbeta = dalpha + (maxit + 2);
    c.hbuf = dbeta + (maxit + 2);
    c.X = c.hbuf + 4096;
    c.slots = reinterpret_cast<unsigned long long*>(c.X + 4 * (size_t)n);
    // the slots proper live in uncached device memory (sc1 accesses skip the L2 look-up: -0.75 us per grid reduction,
    // tools/microbench_sync.hip); the workspace copy above is the fallback
    if (!h->lz_slots) {
        h->lz_slots = (unsigned long long*)msdp_uc_alloc(msdp_lanczos_slot_bytes());       // per-process pool (msdp_api.hip)
    }
    if (h->lz_slots) c.slots = h->lz_slots;
    c.err = reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(c.X + 4 * (size_t)n) + slot_doubles);   // always in the workspace
    int rc = 0, r = 0, nfound = 0, total_steps = 0;
    double lam_max = -1e300;
    const dim3 gr((n + 255) / 256), bl(256);
#define ESC_CHECK(x) do { rc = (x); if (rc) goto done; } while (0)
#define ESC_HIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { msdp_set_error("%s: %s", #x, hipGetErrorString(_e)); rc = MSDP_EHIP; goto done; } } while (0)
    if (use_block) {
        // "cold" = what the independent check of the host loops asks for (escape_deflate = 0, escape_warm = 0): hashed noise
        // only -- no column of Y unless escape_start_y says so, nothing read from or left for other calls
        const bool cold = !h->tune.escape_deflate && !h->tune.escape_warm;
        const bool use_y = cold ? (h->tune.escape_start_y != 0) : true;
        const bool dbg = h->tune.esc_debug != 0;
        const auto tb0 = std::chrono::steady_clock::now();
        double lmx = 0.0, lres = 0.0, lmin_est = 0.0, err = 0.0, lower = -INFINITY;
        int mtop = 0, deg = 0;
        bool conv = false;
        const double* warm_top = (!cold && h->tune.escape_warm && h->esc_top && h->esc_top_n == n) ? h->esc_top : nullptr;
        rc = lanczos_top(c, V, w, dalpha, dbeta, maxit, 1e-7, 777u, warm_top, &lmx, &lres, &lmin_est, Z, &mtop);
        if (!rc && !cold) {
            if (!h->esc_top || h->esc_top_n != n) {
                if (h->esc_top) (void)hipFree(h->esc_top);
                h->esc_top = nullptr; h->esc_top_n = 0;
                if (hipMalloc((void**)&h->esc_top, (size_t)n * sizeof(double)) == hipSuccess) h->esc_top_n = n; else (void)hipGetLastError();
            }
            if (h->esc_top) (void)msdp_memcpy_async(h->esc_top, Z, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
        }
        const auto tb1 = std::chrono::steady_clock::now();
        // dense operand: the explicit S of the affine kinds (no z), or the dense cost matrix of onlyunitdiag (S = C - diag(z))
        const double* Mdense = Mdev ? Mdev : (d.costkind == COST_DENSE ? (const double*)d.Cd : (const double*)nullptr);
        if (!rc) rc = msdp_blockeig_run(h, n, c.rp, c.ci, c.cv, c.z, !rep_sparse, c.Ypt, c.ld, c.p, k, tol, maxdeg, lmx, lres, lmin_est, cold, use_y,
                                        lam_out, Q, &deg, &conv, &err, &lower, Mdense);
        if (!rc) {
            hipError_t e2 = msdp_memcpy(V_out, Q, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) { msdp_set_error("escape_eigs: download of the eigenvectors failed: %s", hipGetErrorString(e2)); rc = MSDP_EHIP; }
        }
        if (!rc) {
            int nvalid = 0;
            for (int t = 0; t < k; ++t) if (std::isfinite(lam_out[t])) ++nvalid;
            h->esc_nvalid = nvalid; h->esc_converged = conv ? 1 : 0; h->esc_maxres = conv ? 0.0 : err;
            // honoured by the host loops only as the outcome of a cold, undeflated call: an estimate, not a certificate
            h->esc_lower = (cold && conv) ? lower : -INFINITY;
            if (lmax_out) *lmax_out = lmx;
            if (iters_out) *iters_out = deg + mtop;
            if (dbg) fprintf(stderr, "[escape] block path: lambda_max run %d steps (%.2f ms, %s start, residual %.1e), block iteration %.2f ms\n", mtop,
                             1e3 * std::chrono::duration<double>(tb1 - tb0).count(), warm_top ? "warm" : "cold", lres,
                             1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tb1).count());
        }
        goto done;
    }
    {
        // Deflate span(Y) only where it IS the near-kernel of S, i.e. at (near-)stationary points:
        // |S*Y|_F is the Riemannian gradient norm.  Away from stationarity the spectrum has no cluster
        // at 0 and plain Lanczos is both sufficient and more accurate.
        const double gnorm = h->h_ctl->norm_grad;
        // Validated on G81/G11/G1: with this threshold the AL loop converges exactly as with the undeflated
        // process (dinf trace to < 1e-8) while the Lanczos runs are ~6x shorter.
        const double dthr = 1e-6;
        const bool deflate_y = h->tune.escape_deflate && h->gradnorm_valid && gnorm <= dthr * std::max(1.0, fabs(h->h_ctl->fx));
        const bool dbg = h->tune.esc_debug != 0;
        auto tp0 = std::chrono::steady_clock::now();
        double ynorm_max = 0.0;
        for (int cidx = 0; deflate_y && cidx < p; ++cidx) {
            double* q = Q + (size_t)r * n;
            hipLaunchKernelGGL(k_extract_col, gr, bl, 0, h->stream, n, d.ld, cidx, (rep_rows || rep_sparse || shard_dense) ? c.Ypt : (const double*)d.Y[cur], q);
            double n0; ESC_CHECK(dev_norm(c, q, &n0));
            ynorm_max = std::max(ynorm_max, n0);
            ESC_CHECK(deflate(c, Q, r, q, 2));
            double n1; ESC_CHECK(dev_norm(c, q, &n1));
            if (n1 > 1e-8 * std::max(ynorm_max, 1e-300) && n1 > 1e-10 * n0) {
                hipLaunchKernelGGL(k_scale_copy, gr, bl, 0, h->stream, n, q, 1.0 / n1, q);
                ++r;
            }
        }
        const int ry = r;
        auto tp1 = std::chrono::steady_clock::now();
        // sequential deflation: smallest eigenpair of the complement; if negative keep it and repeat (<= k times)
        std::vector<double> found;
        bool have_xstart = false;                       // Z doubles as the warm-start buffer until the final Rayleigh-Ritz
        // Across calls: S changes little from one outer iteration to the next, so the first run starts from the
        // bottom eigenvectors the previous call found (deflated against the current Q inside lanczos_smallest)
        if (h->tune.escape_warm && h->esc_prev && h->esc_prev_n == n) {
            ESC_HIP(msdp_memcpy_async(Z, h->esc_prev, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            have_xstart = true;
        }
        h->esc_converged = 1; h->esc_maxres = 0.0; h->esc_nvalid = 0; h->esc_lower = -INFINITY;
        double last_theta = 0.0, last_res = 0.0;
        bool complement_nonneg = false;                   // the last run ended on a Ritz value >= -tol*scale
        for (int t = 0; t < k;) {
            double theta, res, lmx; int m, nacc = 1;
            double thetas[64];
            double* x = Q + (size_t)r * n;
            bool conv = false;
            ESC_CHECK(lanczos_smallest(c, Q, r, V, w, dalpha, dbeta, maxit, tol, 12345u + 7919u * t, &theta, &res, &lmx, x, &m,
                                       std::min(k - t, 64), &nacc, thetas, Z, &have_xstart, &conv));
            total_steps += m;
            if (!conv) {
                h->esc_converged = 0;
                h->esc_maxres = std::max(h->esc_maxres, res / (std::max(fabs(theta), fabs(lmx)) + 1e-300));
            }
            if (dbg) fprintf(stderr, "[escape] run at t=%d: nq=%d steps=%d theta=%.6e res=%.2e lmax=%.4f accepted=%d (host checkpoint analysis so far %.1f ms)\n", t, r, m, theta, res, lmx, nacc, 1e3 * c.host_analysis_s);
            lam_max = std::max(lam_max, lmx);
            for (int i = 0; i < nacc; ++i) found.push_back(thetas[i]);
            r += nacc; nfound += nacc; t += nacc;
            const double scale = std::max(fabs(theta), fabs(lam_max)) + 1e-300;
            last_theta = theta; last_res = res;
            if (!(theta < -tol * scale)) { complement_nonneg = conv; break; }    // no further negative direction
        }
        auto tp2 = std::chrono::steady_clock::now();
        // remember what was found for the next call's warm start (sum of the accepted vectors)
        if (r > ry) {
            if (h->esc_prev) {
                std::vector<double> ones(r - ry, 1.0);
                ESC_HIP(msdp_memcpy_async(c.hbuf, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
                ESC_HIP(hipMemsetAsync(h->esc_prev, 0, (size_t)n * sizeof(double), h->stream));
                hipLaunchKernelGGL(k_multiaxpy, gr, bl, 0, h->stream, n, r - ry, Q + (size_t)ry * n, (int64_t)n, c.hbuf, 1.0, h->esc_prev);
                ESC_HIP(hipStreamSynchronize(h->stream));
                h->esc_prev_n = n;
            }
        }
        // ---- final Rayleigh-Ritz on Z = [Q_Y | X]: recouples the blocks when S*Y is only approximately zero
        const int nz = r;
        ESC_HIP(msdp_memcpy_async(Z, Q, (size_t)nz * n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        // ... and measures how far span(Z) is from invariant: coupling = |(I - ZZ') S Z|_F, from which a lower ESTIMATE of
        // lambda_min follows (Weyl): S = [A E'; E B] in the basis [Z, complement], lambda_min(S) >= min(lambda_min(A),
        // lambda_min(B)) - |E|_2, with A = Z'SZ known exactly and |E|_2 <= |E|_F; lambda_min(B) is taken as theta - res of
        // the last (converged, non-negative) run -- an estimate: a converged Ritz pair proves that some eigenvalue of B lies
        // within res of theta, not that none lies below.
        std::vector<double> M((size_t)nz * nz, 0.0), col(nz + 1);
        double coupling2 = 0.0;
        for (int j = 0; j < nz; ++j) {
            ESC_CHECK(sapply(c, Z + (size_t)j * n, w));
            hipLaunchKernelGGL(k_multidot, dim3(nz), dim3(MSDP_BLOCK), 0, h->stream, n, Z, (int64_t)n, w, c.hbuf);
            hipLaunchKernelGGL(k_dot1, dim3(1), dim3(MSDP_BLOCK), 0, h->stream, n, w, w, c.hbuf + nz, 0);
            ESC_HIP(msdp_memcpy_async(col.data(), c.hbuf, (nz + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            ESC_HIP(hipStreamSynchronize(h->stream));
            double inside = 0.0;
            for (int i = 0; i < nz; ++i) { M[(size_t)i * nz + j] = col[i]; inside += col[i] * col[i]; }
            coupling2 += std::max(0.0, col[nz] - inside);
        }
        for (int i = 0; i < nz; ++i) for (int j = i + 1; j < nz; ++j) {
            const double sm = 0.5 * (M[(size_t)i * nz + j] + M[(size_t)j * nz + i]);
            M[(size_t)i * nz + j] = sm; M[(size_t)j * nz + i] = sm;
        }
        std::vector<double> ew, EV;
        jacobi_eig(nz, M, ew, EV);
        std::vector<int> eo(nz);
        for (int i = 0; i < nz; ++i) eo[i] = i;
        std::sort(eo.begin(), eo.end(), [&](int a2, int b2) { return ew[a2] < ew[b2]; });
        const int nout = std::min(k, nz);
        std::vector<double> cz(nz);
        for (int t = 0; t < k; ++t) {
            if (t < nout) {
                lam_out[t] = ew[eo[t]];
                for (int i = 0; i < nz; ++i) cz[i] = EV[(size_t)i * nz + eo[t]];
                ESC_HIP(msdp_memcpy_async(c.hbuf, cz.data(), nz * sizeof(double), hipMemcpyHostToDevice, h->stream));
                ESC_HIP(hipMemsetAsync(w, 0, n * sizeof(double), h->stream));
                hipLaunchKernelGGL(k_multiaxpy, gr, bl, 0, h->stream, n, nz, Z, (int64_t)n, c.hbuf, 1.0, w);
                ESC_HIP(msdp_memcpy_async(V_out + (size_t)t * n, w, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
                ESC_HIP(hipStreamSynchronize(h->stream));
            } else {
                // fewer than k Ritz pairs exist: +inf marks the missing values (never counted as negative) and the
                // vectors are zero; msdp_escape_info reports how many pairs are real
                lam_out[t] = INFINITY;
                memset(V_out + (size_t)t * n, 0, n * sizeof(double));
            }
        }
        h->esc_nvalid = nout;
        // reported only for a cold-started, undeflated call (the independent check of the host loops): a deflated or
        // warm-started run that missed the bottom of the spectrum would report a bound that is too high
        if (complement_nonneg && nz > 0 && !h->tune.escape_deflate && !h->tune.escape_warm)
            h->esc_lower = std::min(ew[eo[0]], last_theta - last_res) - sqrt(coupling2);
        if (dbg) fprintf(stderr, "[escape] Rayleigh-Ritz bottom %.6e, complement theta %.6e (res %.1e), coupling %.3e -> lower bound %.6e\n",
                         nz > 0 ? ew[eo[0]] : 0.0, last_theta, last_res, sqrt(coupling2), h->esc_lower);
        (void)ry; (void)nfound;
        if (dbg) {
            auto tp3 = std::chrono::steady_clock::now();
            auto ms = [](std::chrono::steady_clock::time_point a2, std::chrono::steady_clock::time_point b2) { return std::chrono::duration<double>(b2 - a2).count() * 1e3; };
            fprintf(stderr, "[escape] orth(Y) %.2f ms (%d cols)  lanczos runs %.2f ms  Rayleigh-Ritz + output %.2f ms\n", ms(tp0, tp1), ry, ms(tp1, tp2), ms(tp2, tp3));
        }
        if (lmax_out) *lmax_out = std::max(lam_max, ew[eo[nz - 1]]);
        if (iters_out) *iters_out = total_steps;
    }
done:
    (void)hipStreamSynchronize(h->stream);
    (void)mem;                                       // kept in the handle for the next call
    return rc;
}


// The escape call proper.  A block-eigen-solver call that ends unconverged (msdp_blockeig.hip gives up when its error estimate
// stops moving: the near-kernel of S wider than the panel and no gap behind it) is repeated on the Lanczos path -- span(Y)
// deflated vector by vector, whatever its width -- inside the same call, so that no host loop has to know the difference.
int msdp_escape_impl(msdp_handle h, int k, double tol, int maxit, double* lam_out, double* V_out, double* lmax_out,
                     int* iters_out, const double* Mdev) {
    int total = 0;
    int rc = escape_impl_once(h, k, tol, maxit, lam_out, V_out, lmax_out, &total, Mdev);
    if (!rc && h->esc_method_last == 1 && !h->esc_converged && h->tune.escape_method != 2) {
        if (h->tune.esc_debug) fprintf(stderr, "[escape] block path did not converge (%d steps): Lanczos path\n", total);
        const int keep = h->tune.escape_method;
        h->tune.escape_method = 1;
        int more = 0;
        rc = escape_impl_once(h, k, tol, maxit, lam_out, V_out, lmax_out, &more, Mdev);
        h->tune.escape_method = keep;
        total += more;
    }
    if (iters_out) *iters_out = total;
    return rc;
}
