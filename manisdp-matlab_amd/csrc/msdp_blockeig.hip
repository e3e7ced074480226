// msdp_blockeig.hip -- block eigen-solver of the saddle-escape step: Chebyshev-filtered subspace iteration.
//
// The reference reads lambda_min, lambda_max and <= delta bottom eigenvectors of the dual slack S = C - diag(z) off a
// dense eig(full(S)) (ManiSDP_onlyunitdiag.m:49-51,74-83).  msdp_escape.hip replaced that by single-vector Lanczos
// runs: one grid reduction per step and one vector per step -- 22 000 steps of 4 us to certify G81, two thirds of
// its time to KKT.  At a near-stationary point the bottom of the spectrum of S is a dense cluster (G81: 80
// eigenvalues below 7e-5 of a spectrum of width 1.86, a dozen of them within 1e-8 of zero), which a single Krylov
// vector resolves one copy at a time.  Here a BLOCK of b = 64 / 128 vectors is iterated as one n x b row-major panel:
//   * filter:  X <- T_d((S - c)/e) X, the scaled Chebyshev polynomial that damps [a, b_up] and grows fastest at the
//     bottom of the spectrum: d steps of the three-term recurrence, each ONE launch of the panel SpMM of the Hess-vec
//     (k_be_step: gather of the neighbour rows, 16 B per lane) and NO reduction -- the recurrence scalars depend on
//     (a, b_up, a0, step) only and are computed by the host ahead of the launches;
//   * Rayleigh-Ritz every d steps: G = X'X and H = X'SX (two b x b Gram matrices, deterministic two-stage reduction),
//     the b x b generalised eigenproblem on the host (Cholesky + Householder/QL), X <- X W and SX <- SX W on the device,
//     residual norms |S x_i - theta_i x_i| from the same pass;
//   * the lower edge a of the damped interval follows the largest Ritz value of the block (Zhou & Saad), so the filter
//     sharpens round by round; convergence of the wanted pairs is governed by the gap to the eigenvalues OUTSIDE the
//     block (lambda_{b+1} - lambda_i), not by the gaps inside the cluster.
// Start block: the columns of the resident factor Y (at a stationary point S*Y = 0: span(Y) IS the near-kernel), the
// bottom vectors of the previous call, hashed noise in the remaining columns -- nothing is deflated, the iteration runs
// on S itself.  The independent check that precedes "Optimality is reached!" (solvers.py, msdp_al_engine.m) starts
// from hashed noise alone: no knowledge of Y, nothing carried over.
// lambda_max (the denominator of dinf, and the upper edge of the filter) comes from a short Lanczos run
// (msdp_escape.hip: lanczos_top) warm-started with the previous call's top vector.
#include "msdp_device.h"
#include <math.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// ---------------------------------------------------------------- device side
struct BeOp {                            // S = C - diag(z) with sparse C (all n rows), or S = M - diag(z) with a dense M
    int n, b, G;                         // b: block width = row stride of the panels (doubles)
    const int* rp; const int* ci; const double* cv; const double* z;     // z may be null (explicit S of the affine kinds)
    int ellW; int64_t ell_stride; const int* ellc; const double* ellv;   // optional ELL copy of the same rows
    const double* M;                     // dense operand (n x nS row-major, msdp_dense.hip layout) or null
};

#define BE_ELL_MAXW 8
// acc[ch] += sum_k C[row,k] * X[k, columns of this lane] (the CSR / ELL gather of msdp_kernels.hip on a b-wide panel)
template <int LPR, int NCH, bool ELL>
__device__ __forceinline__ void be_spmm_row(const BeOp& a, int row, int sub, const double* __restrict__ X, double2 (&acc)[NCH]) {
    if (ELL) {
        int c[BE_ELL_MAXW];
        double v[BE_ELL_MAXW];
#pragma unroll
        for (int w = 0; w < BE_ELL_MAXW; ++w) {
            const bool ok = w < a.ellW;
            c[w] = ok ? a.ellc[(int64_t)w * a.ell_stride + row] : row;
            v[w] = ok ? a.ellv[(int64_t)w * a.ell_stride + row] : 0.0;
        }
#pragma unroll
        for (int w = 0; w < BE_ELL_MAXW; ++w) {
            if (w < a.ellW) {
                const double* src = X + (int64_t)c[w] * a.b + 2 * sub;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const double2 x = ld2(src + ch * 2 * LPR);
                    acc[ch].x = fma(v[w], x.x, acc[ch].x);
                    acc[ch].y = fma(v[w], x.y, acc[ch].y);
                }
            }
        }
        return;
    }
    const int start = a.rp[row], end = a.rp[row + 1];
#pragma unroll 4
    for (int k = start; k < end; ++k) {
        const int c = a.ci[k];
        const double v = a.cv[k];
        const double* src = X + (int64_t)c * a.b + 2 * sub;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const double2 x = ld2(src + ch * 2 * LPR);
            acc[ch].x = fma(v, x.x, acc[ch].x);
            acc[ch].y = fma(v, x.y, acc[ch].y);
        }
    }
}

// One step of the filter recurrence on the whole panel (2*LPR*NCH == b):
//   Xio[row] <- f1 * (S*Xin)[row] - f1*cs * Xin[row] - f2 * Xio[row]        (PREV = false: the last term is absent)
// f1 = 1, cs = 0, PREV = false gives the plain product S*Xin (Rayleigh-Ritz).  Row-local in Xio: in place.
template <int LPR, int NCH, bool ELL, bool PREV>
__global__ __launch_bounds__(MSDP_BLOCK) void k_be_step(BeOp a, const double* __restrict__ Xin, double* __restrict__ Xio,
                                                         double f1, double cs, double f2) {
    int lo, hi;
    msdp_chunk_rows(a.n, a.G, lo, hi);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPW = 64 / LPR;
    const int sub = lane & (LPR - 1), rsub = lane / LPR;
    for (int row0 = lo + wave * RPW; row0 < hi; row0 += MSDP_WAVES * RPW) {
        const int row = row0 + rsub;
        if (row < hi) {
            double2 acc[NCH], x[NCH], pv[NCH];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                acc[ch] = make_double2(0.0, 0.0);
                const int64_t o = (int64_t)row * a.b + 2 * sub + ch * 2 * LPR;
                x[ch] = ld2(Xin + o);
                pv[ch] = PREV ? ld2(Xio + o) : make_double2(0.0, 0.0);
            }
            const double g = -f1 * (a.z[row] + cs);
            be_spmm_row<LPR, NCH, ELL>(a, row, sub, Xin, acc);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                double2 o2;
                o2.x = fma(f1, acc[ch].x, fma(g, x[ch].x, -f2 * pv[ch].x));
                o2.y = fma(f1, acc[ch].y, fma(g, x[ch].y, -f2 * pv[ch].y));
                st2(Xio + (int64_t)row * a.b + 2 * sub + ch * 2 * LPR, o2);
            }
        }
    }
}

// Dense S: the panel product M*Xin comes out of the split-K fp64-MFMA contraction of msdp_dense.hip as SK slabs; this epilogue
// sums them (slab order: deterministic) and applies the recurrence,  Xio <- f1*(sum slabs) - f1*(z + cs)*Xin - f2*Xio.
__global__ __launch_bounds__(256) void k_be_dense_epi(int n, int b, const double* __restrict__ slab, int64_t stride, int SK,
                                                       const double* __restrict__ z, const double* __restrict__ Xin, double* __restrict__ Xio,
                                                       double f1, double cs, double f2, int prev) {
    const int64_t tot = (int64_t)n * b / 2;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = 2 * e;
        const int row = (int)(o / b);
        const double2 acc = msdp_sum_slabs(slab, stride, SK, o);
        const double2 x = ld2(Xin + o);
        const double2 pv = prev ? ld2(Xio + o) : make_double2(0.0, 0.0);
        const double g = -f1 * ((z ? z[row] : 0.0) + cs);
        st2(Xio + o, make_double2(fma(f1, acc.x, fma(g, x.x, -f2 * pv.x)), fma(f1, acc.y, fma(g, x.y, -f2 * pv.y))));
    }
}
int msdp_dense_gemm_at(msdp_handle h, hipStream_t stream, int slab_first, int slabs_reserve, int nmat, const double* const* M,
                       const double* const* X, const double* scale, const int* active_flag, const double** slab_out,
                       int64_t* stride_out, int* SK_out);          // msdp_dense.hip

// (A persistent form of the filter -- panels of the recurrence in registers, neighbour rows through an exchange buffer with sc1
// accesses, one grid barrier per step, a whole round per launch -- was built and measured in round 3: 7.2 us per step against
// 8.1 us for one launch per step on G81.  At b = 64 a step exchanges 10 MB and gathers 51 MB through the coherent path, which
// the launch-per-step form serves from each XCD's L2; the 5 % did not justify a second co-residency-dependent kernel.)
// Partial Gram matrices of one row chunk: part[blk][0] = X'X, part[blk][1] = X'(SX) over the rows of workgroup blk.
// 1024 threads as a TB x TB grid (TB = B/TI), thread (ti, tj) owns the TI x TI outputs (ti*TI + u, tj*TI + v); the rows are
// staged through LDS 2048/B at a time.  Summed over the workgroups in index order by k_be_gram_sum: deterministic.
template <int B, int TI>
__global__ __launch_bounds__(MSDP_BLOCK) void k_be_gram(int n, const double* __restrict__ X, const double* __restrict__ SX,
                                                         double* __restrict__ part) {
    constexpr int BE_GR = 2048 / B;                     // rows staged per pass (2 x 16.6 KB of LDS at any B)
    __shared__ double xs[BE_GR][B + 2], ss[BE_GR][B + 2];
    constexpr int TB = B / TI;                          // TB*TB threads are active (== 1024 for B = 32/64/128 with TI = 1/2/4)
    const int t = threadIdx.x;
    const int ti = t / TB, tj = t - ti * TB;
    const bool act = t < TB * TB;
    const int rows = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows, r1 = min(n, r0 + rows);
    double g[TI][TI], h[TI][TI];
#pragma unroll
    for (int u = 0; u < TI; ++u)
#pragma unroll
        for (int v = 0; v < TI; ++v) { g[u][v] = 0.0; h[u][v] = 0.0; }
    for (int rb = r0; rb < r1; rb += BE_GR) {
        const int nr = min(BE_GR, r1 - rb);
        __syncthreads();
        for (int e = t; e < BE_GR * B; e += MSDP_BLOCK) {
            const int rr = e / B, cc = e - rr * B;
            const bool ok = rr < nr;
            xs[rr][cc] = ok ? X[(int64_t)(rb + rr) * B + cc] : 0.0;
            ss[rr][cc] = ok ? SX[(int64_t)(rb + rr) * B + cc] : 0.0;
        }
        __syncthreads();
        if (act) {
#pragma unroll 4
            for (int rr = 0; rr < BE_GR; ++rr) {
                double xi[TI], xj[TI], sj[TI];
#pragma unroll
                for (int u = 0; u < TI; ++u) { xi[u] = xs[rr][ti * TI + u]; xj[u] = xs[rr][tj * TI + u]; sj[u] = ss[rr][tj * TI + u]; }
#pragma unroll
                for (int u = 0; u < TI; ++u)
#pragma unroll
                    for (int v = 0; v < TI; ++v) { g[u][v] = fma(xi[u], xj[v], g[u][v]); h[u][v] = fma(xi[u], sj[v], h[u][v]); }
            }
        }
    }
    if (act) {
        double* pg = part + (int64_t)blockIdx.x * 2 * B * B;
#pragma unroll
        for (int u = 0; u < TI; ++u)
#pragma unroll
            for (int v = 0; v < TI; ++v) {
                pg[(ti * TI + u) * B + tj * TI + v] = g[u][v];
                pg[B * B + (ti * TI + u) * B + tj * TI + v] = h[u][v];
            }
    }
}
__global__ void k_be_gram_sum(int cnt, int nblk, const double* __restrict__ part, double* __restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int q = 0; q < nblk; ++q) acc += part[(int64_t)q * cnt + e];
        out[e] = acc;
    }
}

// Xn = X*W (W: B x B row-major, staged in LDS) and the partial residual sums rpart[blk][j] = sum over the rows of workgroup
// blk of ((SX*W)[row][j] - theta[j]*Xn[row][j])^2 -- S*Xn is formed in registers only.  One thread = one row x CT columns per pass.
template <int B>
__global__ __launch_bounds__(256) void k_be_rotate(int n, const double* __restrict__ X, const double* __restrict__ SX,
                                                    const double* __restrict__ W, const double* __restrict__ theta,
                                                    double* __restrict__ Xn, double* __restrict__ rpart) {
    extern __shared__ double lds[];
    double* Ws = lds;                                   // [B][B]
    double* red = Ws + B * B;                           // [256 / (B/4)][B]  (row groups x columns)
    constexpr int CT = 4;                               // columns per thread
    constexpr int TPR = B / CT;                         // threads per row
    constexpr int RPB = 256 / TPR;                      // rows per pass
    for (int e = threadIdx.x; e < B * B; e += 256) Ws[e] = W[e];
    __syncthreads();
    const int tr = threadIdx.x / TPR, tc = (threadIdx.x - tr * TPR) * CT;
    const int rows = (n + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows, r1 = min(n, r0 + rows);
    double th[CT], rs[CT];
#pragma unroll
    for (int v = 0; v < CT; ++v) { th[v] = theta[tc + v]; rs[v] = 0.0; }
    for (int row = r0 + tr; row < r1; row += RPB) {
        const double* xr = X + (int64_t)row * B;
        const double* sr = SX + (int64_t)row * B;
        double ax[CT], as[CT];
#pragma unroll
        for (int v = 0; v < CT; ++v) { ax[v] = 0.0; as[v] = 0.0; }
#pragma unroll 4
        for (int i = 0; i < B; i += 2) {
            const double2 xv = ld2(xr + i), sv = ld2(sr + i);
#pragma unroll
            for (int v = 0; v < CT; ++v) {
                const double w0 = Ws[i * B + tc + v], w1 = Ws[(i + 1) * B + tc + v];
                ax[v] = fma(xv.x, w0, ax[v]); ax[v] = fma(xv.y, w1, ax[v]);
                as[v] = fma(sv.x, w0, as[v]); as[v] = fma(sv.y, w1, as[v]);
            }
        }
#pragma unroll
        for (int v = 0; v < CT; v += 2) st2(Xn + (int64_t)row * B + tc + v, make_double2(ax[v], ax[v + 1]));
#pragma unroll
        for (int v = 0; v < CT; ++v) { const double r = fma(-th[v], ax[v], as[v]); rs[v] = fma(r, r, rs[v]); }
    }
#pragma unroll
    for (int v = 0; v < CT; ++v) red[tr * B + tc + v] = rs[v];
    __syncthreads();
    if (threadIdx.x < B) {
        double s = 0.0;
        for (int q = 0; q < RPB; ++q) s += red[q * B + threadIdx.x];
        rpart[(int64_t)blockIdx.x * B + threadIdx.x] = s;
    }
}

__device__ __forceinline__ double be_hash(unsigned row, unsigned col, unsigned seed) {
    unsigned x = row * 2654435761u ^ (col + 1u) * 2246822519u ^ seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (double)x / 4294967296.0 - 0.5;
}
// Start block: columns [0, ny) = the first ny columns of the factor (row stride ld), [ny, ny + nprev) = the previous call's
// bottom vectors (column-major, stride n), the rest (and every column j >= fill_from, used to refill dropped columns) =
// hashed noise.
__global__ void k_be_init(int n, int b, const double* __restrict__ Y, int ld, int ny, const double* __restrict__ prevV, int nprev,
                          unsigned seed, int fill_from, double* __restrict__ X) {
    const int64_t tot = (int64_t)n * b;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / b), j = (int)(e - (int64_t)row * b);
        if (j >= fill_from) X[e] = be_hash((unsigned)row, (unsigned)j, seed);
        else if (fill_from < b) continue;                 // refill pass: only the columns >= fill_from change
        else if (j < ny) X[e] = Y[(int64_t)row * ld + j];
        else if (j < ny + nprev) X[e] = prevV[(int64_t)(j - ny) * n + row];
        else X[e] = be_hash((unsigned)row, (unsigned)j, seed);
    }
}
// columns [0, k) of the row-major panel -> column-major n x k
__global__ void k_be_extract(int n, int b, int k, const double* __restrict__ X, double* __restrict__ V) {
    const int64_t tot = (int64_t)n * k;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / n), row = (int)(e - (int64_t)c * n);
        V[e] = X[(int64_t)row * b + c];
    }
}

// ---------------------------------------------------------------- host side: small dense symmetric algebra
// Symmetric eigen-decomposition, Householder tridiagonalisation + implicit QL (the classical EISPACK tred2 / tql2 pair).
// A: n x n row-major, symmetric, destroyed.  w ascending, Z row-major with the eigenvectors in its ROWS
// (row i of Z = eigenvector of w[i]): the QL rotations then act on two contiguous rows.
static void be_sym_eig(int n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& Z) {
    std::vector<double> d(n), e(n);
    auto V = [&](int i, int j) -> double& { return A[(size_t)i * n + j]; };
    for (int j = 0; j < n; ++j) d[j] = V(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = V(i - 1, j); V(i, j) = 0.0; V(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1], g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g; h -= f * g; d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j]; V(j, i) = f; g = e[j] + V(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += V(k, j) * d[k]; e[k] += V(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) V(k, j) -= (f * e[k] + g * d[k]);
                d[j] = V(i - 1, j); V(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        V(n - 1, i) = V(i, i); V(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = V(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += V(k, i + 1) * V(k, j);
                for (int k = 0; k <= i; ++k) V(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) V(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = V(n - 1, j); V(n - 1, j) = 0.0; }
    V(n - 1, n - 1) = 1.0; e[0] = 0.0;
    // Z = V' (rows = the vectors the rotations mix)
    Z.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Z[(size_t)j * n + i] = A[(size_t)i * n + j];
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
        int m = l;
        while (m < n) { if (fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m > l) {
            int iter = 0;
            do {
                ++iter;
                double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r); d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i]; h = c * p; r = hypot(p, e[i]);
                    e[i + 1] = s * r; s = e[i] / r; c = p / r; p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    double* zi = &Z[(size_t)i * n];
                    double* zi1 = &Z[(size_t)(i + 1) * n];
                    for (int k = 0; k < n; ++k) { const double hk = zi1[k]; zi1[k] = s * zi[k] + c * hk; zi[k] = c * zi[k] - s * hk; }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p; d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1 && iter < 200);
        }
        d[l] += f; e[l] = 0.0;
    }
    std::vector<int> o(n);
    for (int i = 0; i < n; ++i) o[i] = i;
    std::sort(o.begin(), o.end(), [&](int x, int y) { return d[x] < d[y]; });
    w.resize(n);
    std::vector<double> Zs((size_t)n * n);
    for (int i = 0; i < n; ++i) { w[i] = d[o[i]]; memcpy(&Zs[(size_t)i * n], &Z[(size_t)o[i] * n], (size_t)n * sizeof(double)); }
    Z.swap(Zs);
}

// Generalised problem H c = theta G c of the Rayleigh-Ritz stage (G = X'X, H = X'SX, both b x b row-major, symmetrised
// here).  Returns theta ascending and W (b x b row-major, column j = coefficients of Ritz vector j, W'GW = I) and the
// number of usable directions r <= b: the first r columns of W are Ritz vectors, the remaining b - r are zero (the
// caller refills those panel columns with noise).  Cholesky of G where it is well conditioned, else the eigen-basis
// of G with the directions below 1e-13 * max dropped (a start block whose warm-start columns lie in span(Y)).
static int be_ritz(int b, std::vector<double>& Gm, std::vector<double>& Hm, std::vector<double>& theta, std::vector<double>& W) {
    for (int i = 0; i < b; ++i) for (int j = i + 1; j < b; ++j) {
        const double g = 0.5 * (Gm[(size_t)i * b + j] + Gm[(size_t)j * b + i]); Gm[(size_t)i * b + j] = g; Gm[(size_t)j * b + i] = g;
        const double h = 0.5 * (Hm[(size_t)i * b + j] + Hm[(size_t)j * b + i]); Hm[(size_t)i * b + j] = h; Hm[(size_t)j * b + i] = h;
    }
    double dmax = 0.0;
    for (int i = 0; i < b; ++i) dmax = std::max(dmax, Gm[(size_t)i * b + i]);
    if (!(dmax > 0.0) || !std::isfinite(dmax)) return -1;
    // T (b x r, row-major): X*T is orthonormal
    std::vector<double> T;
    int r = b;
    {
        std::vector<double> L(Gm);                       // lower Cholesky factor, row-major
        bool ok = true;
        for (int j = 0; j < b && ok; ++j) {
            double s = L[(size_t)j * b + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)j * b + k] * L[(size_t)j * b + k];
            if (!(s > 1e-11 * dmax)) { ok = false; break; }
            const double ljj = sqrt(s);
            L[(size_t)j * b + j] = ljj;
            for (int i = j + 1; i < b; ++i) {
                double t = L[(size_t)i * b + j];
                for (int k = 0; k < j; ++k) t -= L[(size_t)i * b + k] * L[(size_t)j * b + k];
                L[(size_t)i * b + j] = t / ljj;
            }
        }
        if (ok) {
            // T = L^{-T}: solve L' T = I column by column (upper triangular result)
            T.assign((size_t)b * b, 0.0);
            for (int c = 0; c < b; ++c) {
                for (int i = c; i >= 0; --i) {
                    double t = (i == c) ? 1.0 : 0.0;
                    for (int k = i + 1; k <= c; ++k) t -= L[(size_t)k * b + i] * T[(size_t)k * b + c];
                    T[(size_t)i * b + c] = t / L[(size_t)i * b + i];
                }
            }
        } else {
            std::vector<double> Gc(Gm), gw, gz;
            be_sym_eig(b, Gc, gw, gz);
            const double gtop = gw[b - 1];
            std::vector<int> keep;
            for (int i = b - 1; i >= 0; --i) if (gw[i] > 1e-13 * gtop) keep.push_back(i);
            r = (int)keep.size();
            if (r < 1) return -1;
            T.assign((size_t)b * r, 0.0);
            for (int c = 0; c < r; ++c) {
                const double sc = 1.0 / sqrt(gw[keep[c]]);
                for (int i = 0; i < b; ++i) T[(size_t)i * r + c] = gz[(size_t)keep[c] * b + i] * sc;
            }
        }
    }
    // A = T' H T (r x r)
    std::vector<double> HT((size_t)b * r, 0.0), A((size_t)r * r, 0.0);
    for (int i = 0; i < b; ++i)
        for (int k = 0; k < b; ++k) {
            const double hik = Hm[(size_t)i * b + k];
            if (hik == 0.0) continue;
            const double* tk = &T[(size_t)k * r];
            double* o = &HT[(size_t)i * r];
            for (int c = 0; c < r; ++c) o[c] += hik * tk[c];
        }
    for (int i = 0; i < b; ++i) {
        const double* ti = &T[(size_t)i * r];
        const double* hi = &HT[(size_t)i * r];
        for (int a2 = 0; a2 < r; ++a2) {
            const double t = ti[a2];
            if (t == 0.0) continue;
            double* o = &A[(size_t)a2 * r];
            for (int c = 0; c < r; ++c) o[c] += t * hi[c];
        }
    }
    for (int i = 0; i < r; ++i) for (int j = i + 1; j < r; ++j) {
        const double s = 0.5 * (A[(size_t)i * r + j] + A[(size_t)j * r + i]); A[(size_t)i * r + j] = s; A[(size_t)j * r + i] = s;
    }
    std::vector<double> aw, az;
    be_sym_eig(r, A, aw, az);
    theta.assign(b, INFINITY);
    W.assign((size_t)b * b, 0.0);
    for (int j = 0; j < r; ++j) {
        theta[j] = aw[j];
        const double* q = &az[(size_t)j * r];            // eigenvector j (a row of az)
        for (int i = 0; i < b; ++i) {
            const double* ti = &T[(size_t)i * r];
            double s = 0.0;
            for (int c = 0; c < r; ++c) s += ti[c] * q[c];
            W[(size_t)i * b + j] = s;
        }
    }
    return r;
}

// ---------------------------------------------------------------- host side: the iteration
struct BeMem {                                          // device workspace of one handle (msdp_handle_s::be)
    double* base = nullptr; size_t cap = 0;             // doubles
    double* hpin = nullptr; size_t hpin_cap = 0;        // pinned host staging (Gram matrices, residual partials, W, theta)
    double* prevV = nullptr; int prev_n = 0, prev_k = 0;   // bottom vectors of the previous call (n x prev_k column-major), own allocation
    double prev_a = 0.0;                                // lower filter edge the previous call ended with
};

void msdp_blockeig_release(msdp_handle h) {
    BeMem* m = (BeMem*)h->be;
    if (!m) return;
    if (m->base) (void)hipFree(m->base);
    if (m->prevV) (void)hipFree(m->prevV);
    if (m->hpin) (void)hipHostFree(m->hpin);
    delete m;
    h->be = nullptr;
}

#define BE_GRAM_BLOCKS 128
#define BE_ROT_BLOCKS 512

template <int LPR, int NCH>
static void be_launch_step(msdp_handle h, const BeOp& a, const double* Xin, double* Xio, double f1, double cs, double f2, bool prev) {
    const dim3 grid(a.G), block(MSDP_BLOCK);
    if (a.ellW > 0) {
        if (prev) hipLaunchKernelGGL((k_be_step<LPR, NCH, true, true>), grid, block, 0, h->stream, a, Xin, Xio, f1, cs, f2);
        else hipLaunchKernelGGL((k_be_step<LPR, NCH, true, false>), grid, block, 0, h->stream, a, Xin, Xio, f1, cs, f2);
    } else {
        if (prev) hipLaunchKernelGGL((k_be_step<LPR, NCH, false, true>), grid, block, 0, h->stream, a, Xin, Xio, f1, cs, f2);
        else hipLaunchKernelGGL((k_be_step<LPR, NCH, false, false>), grid, block, 0, h->stream, a, Xin, Xio, f1, cs, f2);
    }
}
// lanes per row: the fewest (>= 8) that put a workgroup's rows into one pass of its 16 waves, cf. lpr_rebalance
static int be_step(msdp_handle h, const BeOp& a, const double* Xin, double* Xio, double f1, double cs, double f2, bool prev) {
    if (a.M) {
        // the contraction kernels take the panel geometry from the handle: borrow it for the launch (single rank, no graph)
        Dev& d = h->d;
        const int ld0 = d.ld, p0 = d.p;
        d.ld = a.b; d.p = a.b;
        const double* Mm[1] = {a.M}; const double* Xx[1] = {Xin}; const double sc[1] = {1.0};
        const double* slab = nullptr; int64_t stride = 0; int SK = 0;
        const double* slab_before = h->slab;
        int rc = msdp_dense_gemm_at(h, h->stream, 0, 0, 1, Mm, Xx, sc, nullptr, &slab, &stride, &SK);
        d.ld = ld0; d.p = p0;
        if (h->slab != slab_before) h->chunk_len = 0;      // the slab buffer grew: captured tCG chunks hold the old pointer
        if (rc) return rc;
        hipLaunchKernelGGL(k_be_dense_epi, dim3(std::min<int64_t>(2048, ((int64_t)a.n * a.b / 2 + 255) / 256)), dim3(256), 0, h->stream, a.n, a.b, slab, stride, SK,
                           a.z, Xin, Xio, f1, cs, f2, prev ? 1 : 0);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int rows_wg = (a.n + a.G - 1) / a.G;
    int lpr = a.b / 2;                                   // one double2 per lane
    if (h->tune.be_lpr > 0) lpr = h->tune.be_lpr;
    else while (lpr > 8 && MSDP_WAVES * (64 / lpr) < rows_wg) lpr >>= 1;
    const int nch = a.b / (2 * lpr);
    if (a.b == 64 && lpr == 32 && nch == 1) be_launch_step<32, 1>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 64 && lpr == 16) be_launch_step<16, 2>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 64 && lpr == 8) be_launch_step<8, 4>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 128 && lpr == 64) be_launch_step<64, 1>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 128 && lpr == 32) be_launch_step<32, 2>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 128 && lpr == 16) be_launch_step<16, 4>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 128 && lpr == 8) be_launch_step<8, 8>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 32 && lpr == 16) be_launch_step<16, 1>(h, a, Xin, Xio, f1, cs, f2, prev);
    else if (a.b == 32 && lpr == 8) be_launch_step<8, 2>(h, a, Xin, Xio, f1, cs, f2, prev);
    else { msdp_set_error("block eigen-solver: no kernel for block width %d with %d lanes per row", a.b, lpr); return MSDP_EUNSUPPORTED; }
    HIPCHK(hipGetLastError());
    return 0;
}

struct BeRR {                                           // outcome of one Rayleigh-Ritz stage
    std::vector<double> theta, res;                     // Ritz values ascending (+inf beyond `rank`), residual norms
    int rank = 0;
};

// Rayleigh-Ritz on the panel X (SX = scratch, receives S*X; Xn receives the Ritz vectors, columns >= rank refilled with
// noise).  X, SX and Xn are three different panels.
static int be_rayleigh_ritz(msdp_handle h, const BeOp& a, BeMem& m, const double* X, double* SX, double* Xn,
                            double* gpart, double* gout, double* Wd, double* rpart, unsigned seed, BeRR& out,
                            double* host_s) {
    const int b = a.b, n = a.n;
    int rc = be_step(h, a, X, SX, 1.0, 0.0, 0.0, false);
    if (rc) return rc;
    if (b == 32) hipLaunchKernelGGL((k_be_gram<32, 1>), dim3(BE_GRAM_BLOCKS), dim3(MSDP_BLOCK), 0, h->stream, n, X, (const double*)SX, gpart);
    else if (b == 64) hipLaunchKernelGGL((k_be_gram<64, 2>), dim3(BE_GRAM_BLOCKS), dim3(MSDP_BLOCK), 0, h->stream, n, X, (const double*)SX, gpart);
    else hipLaunchKernelGGL((k_be_gram<128, 4>), dim3(BE_GRAM_BLOCKS), dim3(MSDP_BLOCK), 0, h->stream, n, X, (const double*)SX, gpart);
    hipLaunchKernelGGL(k_be_gram_sum, dim3((2 * b * b + 255) / 256), dim3(256), 0, h->stream, 2 * b * b, BE_GRAM_BLOCKS, (const double*)gpart, gout);
    HIPCHK(hipGetLastError());
    double* hp = m.hpin;
    HIPCHK(msdp_memcpy_async(hp, gout, (size_t)2 * b * b * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> Gm(hp, hp + (size_t)b * b), Hm(hp + (size_t)b * b, hp + (size_t)2 * b * b), W;
    const int r = be_ritz(b, Gm, Hm, out.theta, W);
    if (host_s) *host_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (r < 1) { msdp_set_error("block eigen-solver: the Rayleigh-Ritz stage broke down (non-finite or zero Gram matrix)"); return MSDP_EHIP; }
    out.rank = r;
    double* hw = hp + (size_t)2 * b * b;                 // W, then theta (finite values only: +inf -> 0 for the residual pass)
    memcpy(hw, W.data(), (size_t)b * b * sizeof(double));
    for (int j = 0; j < b; ++j) hw[(size_t)b * b + j] = j < r ? out.theta[j] : 0.0;
    HIPCHK(msdp_memcpy_async(Wd, hw, ((size_t)b * b + b) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // theta sits right behind W on the device (Wd + b*b)
    const size_t lds = ((size_t)b * b + (size_t)(256 / (b / 4)) * b) * sizeof(double);
    if (b == 32) hipLaunchKernelGGL((k_be_rotate<32>), dim3(BE_ROT_BLOCKS), dim3(256), lds, h->stream, n, X, (const double*)SX, (const double*)Wd, (const double*)(Wd + (size_t)b * b), Xn, rpart);
    else if (b == 64) hipLaunchKernelGGL((k_be_rotate<64>), dim3(BE_ROT_BLOCKS), dim3(256), lds, h->stream, n, X, (const double*)SX, (const double*)Wd, (const double*)(Wd + (size_t)b * b), Xn, rpart);
    else hipLaunchKernelGGL((k_be_rotate<128>), dim3(BE_ROT_BLOCKS), dim3(256), lds, h->stream, n, X, (const double*)SX, (const double*)Wd, (const double*)(Wd + (size_t)b * b), Xn, rpart);
    HIPCHK(hipGetLastError());
    if (r < b) {
        hipLaunchKernelGGL(k_be_init, dim3(1024), dim3(256), 0, h->stream, n, b, (const double*)nullptr, 0, 0, (const double*)nullptr, 0, seed, r, Xn);
        HIPCHK(hipGetLastError());
    }
    double* hr = hw + (size_t)b * b + b;
    HIPCHK(msdp_memcpy_async(hr, rpart, (size_t)BE_ROT_BLOCKS * b * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out.res.assign(b, INFINITY);
    for (int j = 0; j < r; ++j) {
        double s = 0.0;
        for (int q = 0; q < BE_ROT_BLOCKS; ++q) s += hr[(size_t)q * b + j];
        out.res[j] = sqrt(s > 0.0 ? s : 0.0);
    }
    return 0;
}

int msdp_blockeig_eligible(msdp_handle h, const double* Mdev, bool w_loc) {
    if (h->tune.escape_method == 1) return 0;
    if (w_loc) return 0;                                                  // sharded dense product: Lanczos path
    if (Mdev || h->d.costkind != COST_SPARSE) {
        // dense operand (explicit S of the affine kinds, dense C): the filter step is the fp64-MFMA panel product; single rank,
        // matrix order a multiple of nothing in particular (the contraction pads), panels of 64 columns
        if (h->nranks != 1 || h->use_comm || h->lgroup) return 0;
        if (!Mdev && (h->d.costkind != COST_DENSE || !h->d.Cd)) return 0;
        // The explicit S of the affine kinds keeps the Lanczos path unless asked (escape_method = 2): the outer loop of those
        // kinds is chaotic in its end game ("Trajectory sensitivity"), any change of the escape's last digits
        // moves which starts converge, and BQP d = 60 from the reference's default start -- a pinned test -- converges with the
        // Lanczos vectors and ends in "Slow progress" with the block solver's (equally valid) ones.
        if (Mdev && h->tune.escape_method != 2) return 0;
        if (h->d.n < 1024 && h->tune.escape_method != 2) return 0;
        // the panel has to hold span(Y) -- the near-kernel of S at a stationary point -- and noise columns beside it: a factor
        // wider than 48 columns takes the Lanczos path (n = 50000, p = 64..78: the block sat INSIDE the 70-dimensional kernel
        // and every call ran its budget out, round 3)
        if (h->d.p + 16 > 64 && h->tune.escape_method != 2) return 0;
        return h->d.n >= 256;
    }
    if (h->d.p + 16 > 128 && h->tune.escape_method != 2) return 0;           // same, 128-wide panels
    if (h->lgroup) return 0;                                              // in-process ranks share one GPU and one set of tests: Lanczos path
    if (h->d.n < 512 && h->tune.escape_method != 2) return 0;            // small problems: a Lanczos run is a few hundred steps (round 6: 2048 -> 512; G1, n = 800: escape 34 -> 19 ms per solve)
    if (h->d.n < 256) return 0;
    return 1;
}

// The k smallest eigenpairs of S = C - diag(z) by Chebyshev-filtered subspace iteration.
//   rp/ci/cv/z: all n rows of C and z on the device;  Ypt (n x ld, p columns): all rows of the resident factor or nullptr;
//   lmax / lmin_est: lambda_max (converged) and an estimate of lambda_min from the short Lanczos run;
//   cold: start from hashed noise only, keep nothing for the next call;  use_y: put the columns of Y into the start block.
// Outputs: lam[k] ascending, V_dev (n x k column-major, device), degree_out = filter steps + products spent,
// conv_out = every wanted pair passed the stop test, err_out = largest error estimate among the wanted pairs (relative
// to the spectral scale), lower_out = theta_0 minus its error estimate (an ESTIMATE of a lower bound, not a certificate).
int msdp_blockeig_run(msdp_handle h, int n, const int* rp, const int* ci, const double* cv, const double* z, bool own_rows,
                      const double* Ypt, int ld, int p, int k, double tol, int maxdeg, double lmax, double lmax_res, double lmin_est,
                      bool cold, bool use_y, double* lam, double* V_dev, int* degree_out, bool* conv_out, double* err_out,
                      double* lower_out, const double* Mdense) {
    if (!h->be) h->be = new BeMem();
    BeMem& m = *(BeMem*)h->be;
    const bool dbg = h->tune.esc_debug != 0;
    const auto t_start = std::chrono::steady_clock::now();
    // ---- block width
    int ny = (use_y && Ypt) ? p : 0;
    int nprev = (!cold && h->tune.escape_warm && m.prevV && m.prev_n == n) ? m.prev_k : 0;
    int b = h->tune.be_width > 0 ? h->tune.be_width : 64;
    if (b != 32 && b != 64 && b != 128) b = 64;
    if (h->tune.be_width <= 0 && ny + nprev + 8 > b && !Mdense) b = 128;
    if (k + 8 > b && !Mdense) b = 128;
    if (Mdense) b = 64;
    if (k > 64) { msdp_set_error("block eigen-solver: at most 64 eigenpairs per call"); return MSDP_EINVAL; }
    if (ny > b - 8 - std::min(nprev, 8)) ny = b - 8 - std::min(nprev, 8);     // leave room for noise columns (and some warm ones)
    if (ny + nprev > b - 8) nprev = b - 8 - ny;
    // ---- workspace: three panels, Gram partials, Gram sums, [W | theta], residual partials
    const size_t panel = (size_t)n * b;
    const size_t need = 3 * panel + (size_t)BE_GRAM_BLOCKS * 2 * b * b + (size_t)2 * b * b + (size_t)b * b + b + (size_t)BE_ROT_BLOCKS * b + 64;
    if (m.cap < need) {
        (void)hipStreamSynchronize(h->stream);
        if (m.base) (void)hipFree(m.base);
        m.base = nullptr; m.cap = 0;
        if (hipMalloc((void**)&m.base, need * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("block eigen-solver: workspace allocation (%zu MB) failed", need * 8 >> 20); return MSDP_ENOMEM; }
        m.cap = need;
    }
    const size_t hneed = (size_t)2 * b * b + (size_t)b * b + b + (size_t)BE_ROT_BLOCKS * b + 64;
    if (m.hpin_cap < hneed) {
        if (m.hpin) (void)hipHostFree(m.hpin);
        m.hpin = nullptr; m.hpin_cap = 0;
        if (hipHostMalloc((void**)&m.hpin, hneed * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("block eigen-solver: pinned staging allocation failed"); return MSDP_ENOMEM; }
        m.hpin_cap = hneed;
    }
    double* P0 = m.base; double* P1 = P0 + panel; double* P2 = P1 + panel;
    double* gpart = P2 + panel;
    double* gout = gpart + (size_t)BE_GRAM_BLOCKS * 2 * b * b;
    double* Wd = gout + (size_t)2 * b * b;
    double* rpart = Wd + (size_t)b * b + b;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void*)k_be_rotate<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)128 * 128 + 8 * 128) * sizeof(double))));
        HIPCHK(hipFuncSetAttribute((const void*)k_be_rotate<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)64 * 64 + 16 * 64) * sizeof(double))));
        attr_set = true;
    }
    BeOp a;
    a.n = n; a.b = b; a.rp = rp; a.ci = ci; a.cv = cv; a.z = z;
    a.ellW = 0; a.ell_stride = 0; a.ellc = nullptr; a.ellv = nullptr;
    a.M = Mdense;
    if (Mdense && b != 64) { msdp_set_error("block eigen-solver: the dense route works on 64-wide panels"); return MSDP_EUNSUPPORTED; }
    if (!Mdense && own_rows && h->d.ellW > 0 && h->d.n_loc == n) { a.ellW = h->d.ellW; a.ell_stride = h->d.ell_stride; a.ellc = h->d.ellc; a.ellv = h->d.ellv; }
    {
        // grid: one workgroup per CU while its rows fit one pass of eight lanes per row (16 waves x 8 rows = 128 rows), two
        // beyond; measured on G81 (tools/archive/blockeig_tune.py): 256 workgroups x 8 lanes 19.2 ms for the cold check, 512 x 16 lanes
        // 21.8, 128 x 8 lanes 24.1
        int G = (((n + 15) / 16 + 7) / 8) * 8;
        if (G > 256) G = (n > 256 * 128) ? 512 : 256;
        if (G < 8) G = 8;
        if (h->tune.be_grid > 0) G = ((h->tune.be_grid + 7) / 8) * 8;
        a.G = G;
    }
    const unsigned seed = cold ? 0x9e3779b9u : 0x1234567u;
    hipLaunchKernelGGL(k_be_init, dim3(1024), dim3(256), 0, h->stream, n, b, Ypt, ld, ny, (const double*)m.prevV, nprev, seed, b, P0);
    HIPCHK(hipGetLastError());
    // ---- iteration
    const double scale_top = std::max(fabs(lmax), 1e-300);
    double host_s = 0.0;
    int degree = 0, rounds = 0, rc = 0;
    BeRR rr, rr_prev;
    double* X = P0; double* T1 = P1; double* T2 = P2;     // X: current panel; T1, T2: scratch
    // first Rayleigh-Ritz: orthonormalises the start block (Ritz vectors -> T2)
    if ((rc = be_rayleigh_ritz(h, a, m, X, T1, T2, gpart, gout, Wd, rpart, seed + 17u, rr, &host_s))) return rc;
    std::swap(X, T2);                                     // X = Ritz vectors; T1, T2 = scratch
    degree += 1;
    double a0 = std::min(lmin_est, rr.theta[0]);
    const double width0 = std::max(lmax - a0, 1e-300);
    // upper edge of the damped interval: lambda_max of the Lanczos run plus its residual and half a percent of the width
    // (an eigenvalue above it would be amplified like a wanted one)
    const double bup = lmax + std::max(2.0 * lmax_res, 0.005 * width0);
    const int d_round = h->tune.be_degree > 0 ? h->tune.be_degree : (cold ? 400 : 200);
    const double relacc = cold ? 0.0 : 0.02;              // relative accuracy asked of the wanted Ritz values (escape directions, printed dinf)
    const double abstol = (cold ? 0.25 : 1.0) * tol * scale_top;
    bool converged = false;
    double worst = INFINITY;
    std::vector<double> hist;
    std::vector<int> hist_deg;
    if (maxdeg > 20000) maxdeg = 20000;                    // (the callers' budgets are Lanczos-sized; G81's cold check takes 2400 steps)
    while (degree < maxdeg) {
        const int r = rr.rank;
        // lower edge: the largest Ritz value of the block; a block that was cut short (rank < b) uses its own top
        double aedge = rr.theta[r - 1];
        if (rounds == 0 && !cold && m.prev_a > 0.0 && m.prev_n == n) aedge = std::min(aedge, a0 + 8.0 * (m.prev_a - std::min(a0, m.prev_a)) + 1e-3 * width0);
        a0 = std::min(a0, rr.theta[0]);
        const double wid = bup - a0;
        if (aedge > a0 + 0.5 * wid) aedge = a0 + 0.5 * wid;              // never damp less than the upper half
        if (aedge < a0 + 1e-10 * wid) aedge = a0 + 1e-10 * wid;
        // Degree of this round: the filter grows like cosh(d*acosh(x0)) at a0 relative to the damped interval, x0 = (c - a0)/e.
        // Beyond ~1e7 the columns of the filtered block all point at the few lowest eigenvectors and the Gram matrix loses rank
        // (a random point: spectrum [-1.09, 1.15] without a cluster -- 200 steps would be a factor e^88), so d is capped there;
        // at a near-stationary point (G81: x0 - 1 = 5e-5) the cap is 1600 and d_round decides.
        const double x0 = 1.0 + 2.0 * (aedge - a0) / std::max(bup - aedge, 1e-300);
        const int d_cond = (int)std::max(4.0, std::floor(16.0 / std::max(acosh(x0), 1e-6)));
        // A restarted filter only compounds what each round gains, and cosh is flat near 0: where the usual degree amplifies a0 by
        // less than cosh(1) = 1.5 per round (n = 160 000 of the G81 family: 64 eigenvalues within 1e-6 of the spectrum's width,
        // acosh(x0) = 1.9e-3 -- twelve rounds of 200 steps gained a factor 2.3 and the call gave up) the round is made long enough
        // for cosh(3) = 10
        int d_want = d_round;
        if ((double)d_round * acosh(x0) < 1.0) d_want = (int)std::min(2500.0, std::ceil(3.0 / std::max(acosh(x0), 1e-6)));
        const int d = std::min(std::min(d_want, d_cond), std::max(2, maxdeg - degree));
        // scaled Chebyshev recurrence (Zhou & Saad, 2007): the value at a0 stays 1 whatever the degree
        const double e = 0.5 * (bup - aedge), c = 0.5 * (bup + aedge);
        double sigma = e / (a0 - c);
        const double tau = 2.0 / sigma;
        // step 1: T1 = (S X - c X) * sigma/e
        if ((rc = be_step(h, a, X, T1, sigma / e, c, 0.0, false))) return rc;
        double* Xp = X; double* Xc = T1;                                 // previous, current
        for (int i = 2; i <= d; ++i) {
            const double sn = 1.0 / (tau - sigma);
            // new = (S Xc - c Xc) * 2 sn/e - sigma sn Xp, written over Xp
            if ((rc = be_step(h, a, Xc, Xp, 2.0 * sn / e, c, sigma * sn, true))) return rc;
            std::swap(Xp, Xc);
            sigma = sn;
        }
        degree += d;
        ++rounds;
        // Xc holds the filtered block; Xp and T2 are scratch
        rr_prev = rr;
        double* other = T2;                                              // the third panel (Xp and Xc are X and T1 in some order)
        // Ritz vectors -> `other`; S*X scratch = Xp
        if ((rc = be_rayleigh_ritz(h, a, m, Xc, Xp, other, gpart, gout, Wd, rpart, seed + 31u * (unsigned)rounds, rr, &host_s))) return rc;
        degree += 1;
        X = other; T1 = Xp; T2 = Xc;
        // ---- stop test on the wanted pairs: index 0 and every negative one among the first k
        const int rk = rr.rank;
        const double gap_ref = rr.theta[std::min(rk - 1, std::max(k, rk / 2))];   // a Ritz value well inside the block: converged far better than the top
        worst = 0.0;
        bool ok = rk >= std::min(k, b);
        for (int i = 0; i < std::min(k, rk) && ok; ++i) {
            if (i > 0 && !(rr.theta[i] < 0.0)) break;
            // Beyond index 0 only SIGNIFICANTLY negative values are waited for.  The near-kernel of S can be wider than the block
            // (dense C at n = 50000, p = 64: one eigenvalue at -1.3e-3, then more than 64 within 1e-8 of zero): Ritz values inside
            // such a cluster have no gap to the outside of the block and never pass a gap-based test -- and as escape directions
            // (ManiSDP_onlyunitdiag.m:74: nne = min(#negative, delta)) they are as good as they will get.  Round 3: that call
            // crawled through its whole 60 000-step budget, four times over.
            if (i > 0 && rr.theta[i] > -10.0 * abstol) continue;
            const double dth = (i < rr_prev.rank) ? fabs(rr.theta[i] - rr_prev.theta[i]) : INFINITY;
            const double gap = std::max(gap_ref - rr.theta[i], 1e-300);
            // |theta - lambda| <= res always (some eigenvalue lies within the residual), <= res^2/gap when the rest of the spectrum is a gap away
            const double err = std::max(dth, std::min(rr.res[i], rr.res[i] * rr.res[i] / gap));
            const double thr = std::max(abstol, relacc * fabs(rr.theta[i]));
            worst = std::max(worst, err / std::max(scale_top, 1e-300));
            if (!(err <= thr)) ok = false;
        }
        // a call that stopped making progress ends as "not converged" (msdp_escape_info) instead of walking its whole budget
        // (measured against the last round that lies at least eight rounds and 2400 filter steps back: less than a factor two since)
        hist.push_back(ok ? 0.0 : worst);
        hist_deg.push_back(degree);
        if (!ok) {
            int j = (int)hist.size() - 9;
            while (j >= 0 && degree - hist_deg[j] < 2400) --j;
            if (j >= 0 && worst > 0.5 * hist[j]) {
                if (dbg) fprintf(stderr, "[blockeig] no progress over %d rounds / %d steps (worst %.2e, was %.2e): giving up\n", (int)hist.size() - 1 - j, degree - hist_deg[j], worst, hist[j]);
                break;
            }
        }
        if (dbg) fprintf(stderr, "[blockeig] round %d deg %d: a=%.3e a0=%.3e theta0=%.9e theta[k-1]=%.3e top=%.3e res0=%.2e rank=%d worst=%.2e %s\n",
                         rounds, degree, aedge, a0, rr.theta[0], rr.theta[std::min(k, rk) - 1], rr.theta[rk - 1], rr.res[0], rk, worst, ok ? "converged" : "");
        if (ok && rounds >= (cold ? 3 : 2)) { converged = true; break; }
    }
    // ---- outputs
    const int rk = rr.rank;
    for (int t = 0; t < k; ++t) lam[t] = t < rk ? rr.theta[t] : INFINITY;
    hipLaunchKernelGGL(k_be_extract, dim3(1024), dim3(256), 0, h->stream, n, b, std::min(k, rk), (const double*)X, V_dev);
    HIPCHK(hipGetLastError());
    if (rk < k) HIPCHK(hipMemsetAsync(V_dev + (size_t)rk * n, 0, (size_t)(k - rk) * n * sizeof(double), h->stream));
    if (!cold) {
        // warm start of the next call: the bottom vectors found now, and the filter edge reached
        const int kp = std::min(std::min(k, rk), 16);
        if (!m.prevV || m.prev_n != n || m.prev_k < kp) {
            if (m.prevV) { (void)hipStreamSynchronize(h->stream); (void)hipFree(m.prevV); m.prevV = nullptr; }
            if (hipMalloc((void**)&m.prevV, (size_t)n * 16 * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); m.prevV = nullptr; m.prev_n = 0; m.prev_k = 0; }
        }
        if (m.prevV) {
            HIPCHK(msdp_memcpy_async(m.prevV, V_dev, (size_t)n * kp * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            m.prev_n = n; m.prev_k = kp; m.prev_a = rr.theta[rk - 1];
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (degree_out) *degree_out = degree;
    if (h->tune.fail_block) { h->tune.fail_block = 0; converged = false; }      // test hook: report this call as unconverged
    if (conv_out) *conv_out = converged;
    if (err_out) *err_out = worst;
    if (lower_out) *lower_out = rr.theta[0] - worst * scale_top;
    if (dbg) fprintf(stderr, "[blockeig] b=%d ny=%d nprev=%d cold=%d: %d rounds, degree %d, %.2f ms (host algebra %.2f ms), theta0=%.9e, %s\n", b, ny, nprev,
                     (int)cold, rounds, degree, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), 1e3 * host_s,
                     rr.theta[0], converged ? "converged" : "NOT converged");
    return 0;
}

// ---------------------------------------------------------------- test-only entry points (host code only: no GPU needed)
extern "C" int msdp_debug_sym_eig(int32_t n, const double* A, double* w, double* Z) {
    if (n < 1 || !A || !w || !Z) { msdp_set_error("debug_sym_eig: bad argument"); return MSDP_EINVAL; }
    std::vector<double> a(A, A + (size_t)n * n), ww, zz;
    be_sym_eig(n, a, ww, zz);
    memcpy(w, ww.data(), (size_t)n * sizeof(double));
    memcpy(Z, zz.data(), (size_t)n * n * sizeof(double));
    return 0;
}
extern "C" int msdp_debug_ritz(int32_t b, const double* G, const double* H, double* theta, double* W, int32_t* rank) {
    if (b < 1 || !G || !H || !theta || !W || !rank) { msdp_set_error("debug_ritz: bad argument"); return MSDP_EINVAL; }
    std::vector<double> g(G, G + (size_t)b * b), hm(H, H + (size_t)b * b), th, w;
    const int r = be_ritz(b, g, hm, th, w);
    if (r < 1) { msdp_set_error("debug_ritz: breakdown"); return MSDP_EINVAL; }
    memcpy(theta, th.data(), (size_t)b * sizeof(double));
    memcpy(W, w.data(), (size_t)b * b * sizeof(double));
    *rank = r;
    return 0;
}
