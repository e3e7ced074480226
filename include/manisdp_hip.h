/*
 * manisdp_hip.h -- C ABI of libmanisdp_hip.so, the MI355X (gfx950) hot path of
 * ManiSDP's three primal entry points.
 *
 * The reference (wangjie212/ManiSDP-matlab) has no FFI on this path: its operator
 * interface is Manopt's problem struct consumed by trustregions()
 *   problem.cost / problem.grad / problem.hess / problem.M.*
 *   (src/primal/ManiSDP_onlyunitdiag.m:28-30,39,43; ManiSDP_unitdiag.m:41-43,53,57;
 *    ManiSDP_unittrace.m:42-44,53,57).
 * This library is cut AT the trustregions() call: the factor Y stays resident in
 * HBM for the whole Riemannian trust-region solve and never crosses PCIe inside
 * tCG.  The only native-call precedent in the reference is the MEX gateway of
 * src/C-files/<fn>.cpp (plain double* via mxGetPr, errors via mexErrMsgIdAndTxt);
 * the MEX shim in manisdp-matlab_amd/matlab/manisdp_mex.cpp binds exactly the
 * entry points declared here.
 *
 * Conventions
 *  - every function returns 0 on success, a negative MSDP_E* code otherwise;
 *    msdp_last_error() returns a human-readable message (no exceptions cross the ABI);
 *  - all floating-point data is IEEE double (the reference is fp64 throughout);
 *  - sparse inputs are MATLAB-style compressed columns with 64-bit indices
 *    (mwIndex Jc/Ir, 0-based);
 *  - factor layout at the boundary is the reference's own:
 *      oblique entry points (onlyunitdiag, unitdiag): Y is p x n column-major
 *        (each point's p-vector contiguous; ManiSDP_unitdiag.m:59 X = Y'*Y);
 *      unittrace: Y is n x p column-major (ManiSDP_unittrace.m:59 X = Y*Y');
 *  - the library is single-caller per handle (MATLAB's interpreter thread).
 */
#ifndef MANISDP_HIP_H
#define MANISDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDP_OK              0
#define MSDP_EINVAL         -1   /* bad argument                                 */
#define MSDP_EHIP           -2   /* HIP runtime error                            */
#define MSDP_ENOMEM         -3   /* device or host allocation failed             */
#define MSDP_ESTATE         -4   /* call order violated (e.g. no point resident) */
#define MSDP_ECOMM          -5   /* RCCL error                                   */
#define MSDP_EUNSUPPORTED   -6

typedef struct msdp_handle_s* msdp_handle;

/* Problem kinds (which reference entry point the handle serves). */
#define MSDP_KIND_ONLYUNITDIAG 1   /* src/primal/ManiSDP_onlyunitdiag.m */
#define MSDP_KIND_UNITDIAG     2   /* src/primal/ManiSDP_unitdiag.m     */
#define MSDP_KIND_UNITTRACE    3   /* src/primal/ManiSDP_unittrace.m    */
#define MSDP_KIND_GENERIC      4   /* src/primal/ManiSDP.m (Euclidean manifold) */
#define MSDP_KIND_MULTIBLOCK   5   /* src/primal/ManiSDP_multiblock.m (product manifold) */
#define MSDP_KIND_DUAL_UNITDIAG 6  /* src/dual/ManiDSDP_unitdiag.m (dual approach, diag(S) = 1) */

/* Options of one Riemannian trust-region solve: the fields ManiSDP sets
 * (ManiSDP_unitdiag.m:44-47) plus Manopt's defaults that are in force
 * (manopt7.0/manopt/solvers/trustregions/trustregions.m:340-351,363-372).
 * Delta_bar <= 0 / Delta0 <= 0 select the reference's defaults
 * (M.typicaldist(), Delta_bar/8). */
typedef struct {
    int32_t maxiter;          /* opts.maxiter  = options.TR_maxiter            */
    int32_t maxinner;         /* opts.maxinner = options.TR_maxinner           */
    int32_t mininner;         /* Manopt default 1                              */
    int32_t reserved0;
    double  tolgradnorm;      /* opts.tolgradnorm                              */
    double  kappa;            /* 0.1                                           */
    double  theta;            /* 1.0                                           */
    double  rho_prime;        /* 0.1                                           */
    double  rho_regularization; /* 1e3                                         */
    double  Delta_bar;        /* <=0: typicaldist                              */
    double  Delta0;           /* <=0: Delta_bar/8                              */
} msdp_rtr_opts;

/* What the AL loop reads back from trustregions(): info(end).gradnorm and the
 * cost, plus the counters the reference does not report. */
typedef struct {
    double  cost;             /* f at the returned point                       */
    double  gradnorm;         /* info(end).gradnorm                            */
    double  Delta;            /* final trust-region radius                     */
    double  seconds;          /* wall-clock of the solve (host timer)          */
    int32_t iters;            /* TR iterations performed                       */
    int32_t hessvecs;         /* sum(info.numinner)                            */
    int32_t accepted;
    int32_t rejected;
    int32_t cost_evals;
    int32_t last_stop_inner;  /* tCG stop code 1..6 of the last TR iteration   */
    int32_t reserved[2];
} msdp_rtr_stats;

void msdp_rtr_default_opts(msdp_rtr_opts* o);

/* ---------------------------------------------------------------- life cycle */

/* Select the HIP device of the calling thread (one process per GPU: LOCAL_RANK). */
int msdp_set_device(int32_t device);
int msdp_device_count(int32_t* count);

/* min <C,X>, diag X = 1 with sparse symmetric C (n x n) given as MATLAB CSC
 * (= CSR by symmetry).  Replaces the closures of ManiSDP_onlyunitdiag.m:117-130.
 * pcap: initial capacity for the factor width (grown on demand). */
int msdp_create_onlyunitdiag_csc(int64_t n, const int64_t* jc, const int64_t* ir,
                                 const double* pr, int32_t pcap, msdp_handle* out);

/* Same with a dense symmetric C (n x n, column-major == row-major). */
int msdp_create_onlyunitdiag_dense(int64_t n, const double* C, int32_t pcap,
                                   msdp_handle* out);

/* Pre-sharded synthetic dense problem (BASELINE config 5: n = 100000, p = 64 over 8 GPUs; the full C would be
 * 80 GB): rank `rank` of `nranks` fills ITS rows of a dense symmetric C on the device from a counter-based
 * generator; msdp_synthetic_dense_entry is the same generator on the host (parity tests). */
int msdp_create_onlyunitdiag_dense_synthetic(int64_t n, uint64_t seed, int32_t nranks, int32_t rank,
                                             int32_t pcap, msdp_handle* out);
double msdp_synthetic_dense_entry(int64_t n, int64_t i, int64_t j, uint64_t seed);
/* Test-only: stand in for the all-gather on a communicator-free shard (one process = rank r of N). */
int msdp_debug_set_full_rows(msdp_handle h, const double* rows_host);
/* Test-only: the point-to-point calls of the halo exchange (one ncclGroupStart .. ncclSend + ncclRecv .. ncclGroupEnd on the
 * handle's communicator and stream) with the handle's own rank as peer: `count` doubles in -> out through two device
 * buffers.  Lets a single-GPU box run the RCCL half of the exchange; the send / receive lists are covered by the
 * in-process ranks (tests/test_gpu_local_ranks.py). */
int msdp_debug_p2p_self(msdp_handle h, int64_t count, const double* in_host, double* out_host);
/* Test-only: eta and Heta of the last tCG solve of msdp_rtr (reference layout), valid after a call with maxiter = 1 on the
 * chunked path or on the persistent path with option fused_rtr = 0.  Lets a test check tCG's invariant Heta = Hess(eta)
 * (tCG.m:192-220) on the device, i.e. bound what the persistent kernel's linearity trick for C*mdelta may drift. */
int msdp_debug_get_tcg_step(msdp_handle h, double* eta, double* Heta);
/* Test-only, host code only (no GPU touched): the dense symmetric eigen-solver (Householder + implicit QL; w ascending, row i
 * of Z = eigenvector i) and the generalised Rayleigh-Ritz problem H c = theta G c (theta ascending, +inf beyond *rank; W b x b
 * row-major, column j = coefficients of Ritz vector j, W'GW = I on the first *rank columns) of the block eigen-solver. */
int msdp_debug_sym_eig(int32_t n, const double* A, double* w, double* Z);
int msdp_debug_ritz(int32_t b, const double* G, const double* H, double* theta, double* W, int32_t* rank);
/* Test-only: make a sparse-C handle rank `rank` of `nranks` WITHOUT a communicator (same row split, local CSR/ELL
 * rows with global column indices and gather buffer as msdp_comm_init sets up), so that one GPU can check every
 * shard's kernels against the unsharded result.  Call right after create, before any point is set. */
int msdp_debug_shard(msdp_handle h, int32_t nranks, int32_t rank);

/* min <C,X>, A(X) = b, diag X = 1 (kind = MSDP_KIND_UNITDIAG;
 * ManiSDP_unitdiag.m:152-171) or tr X = 1 (kind = MSDP_KIND_UNITTRACE;
 * ManiSDP_unittrace.m:156-177), or no further structure at all (kind =
 * MSDP_KIND_GENERIC; ManiSDP.m:149-164 on euclideanfactory(n, p): proj = identity,
 * retr = Y + U).  At is n^2 x m CSC (column k = vec(A_k)), b dense m, c dense n^2
 * (the shim densifies a sparse c).  Y is n x p column-major for UNITTRACE and GENERIC. */
int msdp_create_affine(int32_t kind, int64_t n, int64_t m,
                       const int64_t* at_jc, const int64_t* at_ir, const double* at_pr,
                       const double* b, const double* c, int32_t pcap, msdp_handle* out);

/* min <C,X>, A(X) = b over block-diagonal X = diag(X_1..X_nb) with diag(X_i) = 1 for the first `nob` blocks
 * (ManiSDP_multiblock.m:1-7; closures cost/grad/hess :208-249; manifold = multiblockmanifold.m:1-42 over the MEX
 * helpers src/C-files/{innerc,lincombc,projc,retrc,randc,zerovecc}.cpp: oblique factors for the first nob blocks,
 * Euclidean ones for the rest).  SeDuMi data for several SDP blocks: At is (sum n_i^2) x m CSC whose rows are the
 * concatenated column-major vecs of the blocks, c likewise, b dense m.
 * The factor crosses the boundary as ONE p x N column-major array, N = sum n_i: block i occupies columns
 * [N_i, N_i + n_i) and, when its own width p_i is smaller than p, rows p_i..p-1 of those columns are zero (zero
 * rows of a block stay zero under cost, gradient, Hess-vec, projection and retraction, so padding changes nothing).
 * Internally the problem is the unit-diagonal affine kind of order N whose constraint matrices are block diagonal,
 * with a per-row flag that switches the projection / normalisation terms off for the Euclidean blocks.
 * Storage: up to 15 blocks with N < 4096 the operands are embedded N x N arrays; beyond that every operand is the
 * concatenation of its diagonal blocks (memory and work ~ sum n_i^2; environment MSDP_MULTIBLOCK_BLOCKED = 0 / 1 forces
 * either form), msdp_get_dual_slack_block is then the way to read S and msdp_get_dual_slack returns MSDP_EUNSUPPORTED. */
int msdp_create_multiblock(int32_t nb, const int64_t* block_n, int32_t nob, int64_t m,
                           const int64_t* at_jc, const int64_t* at_ir, const double* at_pr,
                           const double* b, const double* c, int32_t pcap, msdp_handle* out);

/* Dual approach with a unit-diagonal dual slack (src/dual/ManiDSDP_unitdiag.m:8-220):
 *     sup <C,X> + <cf,w>   s.t.  A(X) + B(w) = b,  X psd,  the variable of the Riemannian subproblem is S = Y'Y with
 * diag(S) = 1 (obliquefactoryNTrans, :196-219), the multipliers are x = vec(X) and w.  Closures replaced:
 * cost :174-181, grad :183-187, hess :189-194, co/line_search :155-172; outer-step bookkeeping :70-85.
 * at_* is A' (n^2 x m CSC: column k = vec of the k-th row of the PSD part A(:, K.f+1:end)), dAAt = diag(A*A') (:37,
 * options.dAAt), c the PSD part of the cost (dense n^2), B = A(:, 1:K.f) as m x nf CSC (nf may be 0), cf its costs. */
int msdp_create_dual_unitdiag(int64_t n, int64_t m, const int64_t* at_jc, const int64_t* at_ir, const double* at_pr,
                              const double* dAAt, const double* b, const double* c, int32_t nf, const int64_t* b_jc,
                              const int64_t* b_ir, const double* b_pr, const double* cf, int32_t pcap, msdp_handle* out);
/* sigma and the free multipliers w (nf values) for the following trustregions() calls (the matrix multiplier x lives on
 * the device and starts at 0, :48).  Must be called before the first solve and after every msdp_dual_outer_step. */
int msdp_dual_set_penalty(msdp_handle h, double sigma, const double* w);
/* Outer step :70-81 at the resident point with the current sigma: y = iA'*(S(:) - c); As = A'y - sc; x -= sigma*As on the
 * device; eX = x + bA; z = sum(S.*eX); X = eX - diag(z) stays on the device for msdp_escape_eigs_dual /
 * msdp_get_dual_slack.  scal[0] = b'y, scal[1] = <C, eX>, scal[2] = |As|^2; Af[nf] = B'y - cf; z[n]. */
int msdp_dual_outer_step(msdp_handle h, double* scal, double* Af, double* z);
/* y of the last msdp_dual_outer_step (m values; data.y, :134) */
int msdp_dual_get_y(msdp_handle h, double* y);

int msdp_destroy(msdp_handle h);
/* The library keeps ONE device allocation beyond the life of the handles: the Lanczos workspace of the escape (up to 24 GB for
 * n = 20000; allocating it costs 0.05-0.5 s) is parked by msdp_destroy for the next handle of the process.  This call
 * frees it (a long-lived host such as MATLAB calls it when it unloads the binding). */
int msdp_release_cache(void);
/* The uncached device memory behind the grid synchronisations and row exchanges comes from per-process arenas with a coalescing
 * sub-allocator (msdp_api.hip): bytes the arenas hold, bytes handed out to live handles, number of arenas.  The pool grows to
 * the high-water mark of what was live together; arenas return to the driver only while no uncached block of the process is
 * live (beyond MSDP_UC_POOL_CAP bytes -- default 1 GiB -- when the last one is freed, all of them in msdp_release_cache). */
int msdp_debug_pool_stats(int64_t* pool_bytes, int64_t* live_bytes, int64_t* arenas);
/* Free and total bytes of the current device (hipMemGetInfo), for hosts and tests that size a problem or watch for leaks
 * without a second HIP runtime in the process. */
int msdp_debug_mem_info(int64_t* free_bytes, int64_t* total_bytes);

/* AL state that changes between trustregions() calls: y and sigma
 * (ManiSDP_unitdiag.m:64,108-112).  No-op error for onlyunitdiag handles. */
int msdp_set_multipliers(msdp_handle h, const double* y, double sigma);

/* ------------------------------------------------------- resident point I/O */

/* Upload the current point in the reference layout (see header comment). */
int msdp_set_point(msdp_handle h, int32_t p, const double* Y);
int msdp_get_point(msdp_handle h, double* Y);
/* Row-sharded handles (msdp_comm_init): every row of the resident point on EVERY rank (one all-gather, then the
 * download).  The host loops of the row-sharded affine kinds run replicated on all ranks and take their rank / escape
 * decisions on identical data.  Without a communicator: the same as msdp_get_point. */
int msdp_get_point_all(msdp_handle h, double* Y);

/* The rank decision and the re-shaping of the factor between two trustregions() calls without the factor leaving the
 * device (ManiSDP_onlyunitdiag.m:52-54 svd(Y), :70-73 Y = V(:,1:r)'.*e(1:r), :78-83 Y = [Y; alpha*vS'],
 * Y = Y./sqrt(sum(Y.^2)); the same lines of ManiSDP_unitdiag.m:72-74,93-106).
 * msdp_factor_gram: G (p x p, symmetric) = Gram matrix of the p columns of the factor; its eigen-decomposition
 *   G = Q diag(e.^2) Q' gives the singular values e of Y and V(:,k)*e_k = Y*Q(:,k).
 * msdp_factor_rotate: Y <- Y*Q with Q p x r ROW-major (the rank cut; new width r).
 * msdp_factor_append: Y <- [Y, alpha*V], V n x k column-major (as msdp_escape_eigs returns it), then every row scaled to
 *   unit norm if normalize != 0 (oblique kinds).  MSDP_EUNSUPPORTED when p + k exceeds the width the handle has
 *   allocated: re-enter through msdp_set_point then. */
int msdp_factor_gram(msdp_handle h, double* G);
int msdp_factor_rotate(msdp_handle h, int32_t r, const double* Q);
int msdp_factor_append(msdp_handle h, int32_t k, const double* V, double alpha, int32_t normalize);
int msdp_get_p(msdp_handle h, int32_t* p);
/* MSDP_KIND_* of the handle: tells a binding which factor layout the handle expects at the boundary
 * (p x n for ONLYUNITDIAG / UNITDIAG, n x p for UNITTRACE / GENERIC) without guessing from array shapes. */
int msdp_get_kind(msdp_handle h, int32_t* kind);

/* ------------------------------------------------------------------ hot path */

/* Keep / bring back a device-side copy of the resident point (same width): restart a solve from the same
 * start point without another host upload (measurement; line searches that want to back-track on the device). */
int msdp_point_snapshot(msdp_handle h);
int msdp_point_restore(msdp_handle h);

/* [Y, ~, info] = trustregions(problem, Y, opts) on the resident point: the whole
 * RTR/tCG loop (trustregions.m:441-767, tCG.m:160-289) runs on the device. */
int msdp_rtr(msdp_handle h, const msdp_rtr_opts* opts, msdp_rtr_stats* stats);

/* Convenience: set_point + rtr + get_point. */
int msdp_rtr_host(msdp_handle h, int32_t p, double* Y_inout,
                  const msdp_rtr_opts* opts, msdp_rtr_stats* stats);

/* Fine-grained operators at the resident point (parity tests; reference layout).
 *   cost   : problem.cost(Y)                       -> *f
 *   rgrad  : problem.grad(Y) (after cost)          -> G
 *   hessvec: problem.hess(Y, U)                    -> H
 *   proj   : problem.M.proj(Y, U)                  -> V
 *   retr   : problem.M.retr(Y, U)                  -> Z (does not move the point) */
int msdp_cost(msdp_handle h, double* f);
int msdp_rgrad(msdp_handle h, double* G);
int msdp_hessvec(msdp_handle h, const double* U, double* H);
int msdp_proj(msdp_handle h, const double* U, double* V);
int msdp_retr(msdp_handle h, const double* U, double* Z);

/* Per-point vector the AL step needs after RTR without recomputing it on the host:
 * onlyunitdiag: z = sum((Y*C).*Y) (ManiSDP_onlyunitdiag.m:46-47), length n. */
int msdp_get_z(msdp_handle h, double* z);
/* Row-sharded handles: z of all rows on every rank (one all-gather); otherwise the same as msdp_get_z. */
int msdp_get_z_all(msdp_handle h, double* z);

/* co(Y) of the line search (ManiSDP_onlyunitdiag.m:99-101, ManiSDP_unitdiag.m:131-136,
 * ManiSDP_unittrace.m:135-140) evaluated at the retraction of Y + alpha*U, U given in
 * the reference layout; used by line_search(). alpha = 0 evaluates co(Y). */
int msdp_linesearch_cost(msdp_handle h, const double* U, double alpha, double* val);
/* Adopt the trial point of the last msdp_linesearch_cost call (nY of line_search)
 * as the resident point. */
int msdp_linesearch_accept(msdp_handle h);

/* ----------------------------------------------------------------- escape */

/* Few-eigenvector saddle escape (replaces eig(S) of ManiSDP_onlyunitdiag.m:50,
 * ManiSDP_unitdiag.m:68, ManiSDP_unittrace.m:68 for large n): the k smallest
 * eigenpairs and the largest eigenvalue of the dual slack S at the resident point,
 * by block Lanczos / LOBPCG on the device re-using the S*U kernel.
 * lam_min[k] ascending, V is n x k column-major, *lam_max the top eigenvalue. */
int msdp_escape_eigs(msdp_handle h, int32_t k, double tol, int32_t maxit,
                     double* lam_min, double* V, double* lam_max, int32_t* iters);

/* Same escape for an explicit dense symmetric S (n x n, column-major = row-major) formed by the AL loop of the
 * affine kinds (ManiSDP_unitdiag.m:65-68, ManiSDP_unittrace.m:65-68: S = eS - diag(z) / eS - z*I, eig(S)). */
int msdp_escape_eigs_matrix(msdp_handle h, const double* S, int32_t k, double tol, int32_t maxit,
                            double* lam_min, double* V, double* lam_max, int32_t* iters);

/* Which eigen-solver the LAST escape call on this handle ran: 1 = block Chebyshev-filtered subspace iteration
 * (msdp_blockeig.hip: sparse C; a 64- or 128-wide panel, reduction-free filter steps on the S*U kernel, Rayleigh-Ritz every
 * few hundred steps), 0 = deflated single-vector Lanczos runs (msdp_escape.hip: dense S, pre-sharded dense C, small n). */
int msdp_escape_method(msdp_handle h, int32_t* method);

/* Collective calls this handle has issued on its communicator so far (row exchange, all-reduce, all-gather; operations grouped
 * into one RCCL launch count once).  A row-sharded tCG trip of onlyunitdiag with sparse C costs two (msdp_trip1.hip: the rows
 * of the projected residual with every rank's partial sums riding along, then the all-reduce of <mdelta, H mdelta> -- tCG.m:166),
 * three with option trip1 = 0; tests/test_gpu_local_ranks.py asserts it. */
int msdp_debug_collective_calls(msdp_handle h, int64_t* calls);
/* (new, measurement only) Device time of the LAST msdp_rtr call on this handle, by HIP events on the handle's stream around what the
 * call enqueued behind the cost / gradient evaluation of its start point (trustregions.m:441-767): on the fused path exactly the ONE launch
 * that runs every tCG trip, retraction, cost evaluation and decision of the call -- bench.py's dominant kernel, whose roofline figure is
 * algorithmic bytes of the launch / this time.  *ms < 0: no call yet. */
int msdp_debug_last_rtr_device_ms(msdp_handle h, double* ms);
/* Measurement only: average stream time (us) of one collective call: which = 0 row exchange, 1 all-reduce of one partial-sum
 * array, 2 row exchange with the sums riding along, 3 all-reduce of three arrays, 4 rows and sums as two separate all-gathers. */
int msdp_debug_time_collective(msdp_handle h, int32_t which, int32_t reps, double* avg_us);
/* Measurement only: the phases of a persistent tCG trip (tCG.m:160-289 inside k_tcg_persist_obl).  Runs `reps` trips with the
 * exits disabled on the traced instance of the kernel (17 <= p <= 32, rows of <= 5 entries: the G81 shape) and returns thread 0's
 * s_memtime stamps of every workgroup at 7 phase boundaries -- 0 top of the trip, 1 gathers + row arithmetic done, 2 first grid
 * reduction returned, 3 trial step formed and residual rows stored, 4 those stores performed, 5 second grid reduction returned,
 * 6 new direction formed -- for the trips dims[2] .. dims[2] + dims[1] - 1: out[(g * dims[1] + t) * 8 + phase], g < dims[0].
 * avg_ms = the average trip time of the same launch (HIP events).  tools/persist_timeline.py turns it into profiles/r4_persist_timeline.md.
 * reps <= 0 (round 6): the FUSED launch instead -- one trustregions() call (trustregions.m:441-767) with the options of the handle's last
 * msdp_rtr, stamps of the first dims[1] TR iterations (dims[2] = 0): 0 iteration starts, 1 first trip's products formed, 2 tCG ended
 * (bits 56..63: its trips), 3 proposal rows stored and performed, 4 barrier returned, 5 cost / gradient rows of the proposal formed,
 * 6 the iteration's reduction returned, 7 decision taken; avg_ms = the call's time.  tools/fused_timeline.py. */
int msdp_debug_persist_trace(msdp_handle h, int32_t reps, uint64_t* out, int64_t cap, int32_t* dims, double* avg_ms);

/* Outcome of the LAST msdp_escape_eigs / _matrix / _dual call on this handle.  The reference's eig(S) is exact;
 * a Lanczos run that reaches `maxit` without passing a stop test only yields an UPPER bound of lambda_min, so
 * dinf = max(0,-lambda_min)/(1+lambda_max) (ManiSDP_onlyunitdiag.m:51) would be under-estimated: the AL loop
 * must not declare optimality on it.
 *   *nvalid    : how many of the k requested pairs are real (the others are returned as lam = +inf, V = 0);
 *   *converged : 1 if every Lanczos run passed a stop test, 0 if one ended at maxit;
 *   *residual  : largest relative residual |S x - theta x| / max(|theta|, |lam_max|) among the unconverged runs. */
int msdp_escape_info(msdp_handle h, int32_t* nvalid, int32_t* converged, double* residual);
/* Lower ESTIMATE of lambda_min(S) from the LAST escape call; -inf unless that call was cold-started and undeflated
 * (options escape_deflate = 0, escape_warm = 0) and converged.  Block path: theta_0 minus its error estimate.  Lanczos path:
 * with Z = [found vectors] and S = [A E'; E B] in the basis [Z, complement],
 *   min(lambda_min(A), theta - res) - |E|_F
 * where A = Z'SZ is known exactly, |E|_F = |(I - ZZ')SZ|_F is measured and (theta, res) is the last converged,
 * non-negative Ritz pair of the complement.  NOT a certificate: a converged Ritz pair proves that SOME eigenvalue lies within
 * res of theta, not that none lies below; a warm-started or deflated run that missed the bottom of the
 * spectrum would report a bound that is too high, which is why such runs report -inf and the host loops always finish with
 * the independent, cold-started check. */
int msdp_escape_lower_bound(msdp_handle h, double* lam_lower);

/* ------------------------------------------------------------- multi-GPU */

/* Row sharding: call on every rank right after create, before any
 * point is set.  unique_id is the 128-byte RCCL id produced on rank 0 by
 * msdp_comm_unique_id and broadcast by the host launcher. */
int msdp_comm_unique_id(void* id128);
int msdp_comm_init(msdp_handle h, int32_t nranks, int32_t rank, const void* id128);
/* Test / diagnostic stand-in for the communicator: member `rank` of the in-process group `group_id` of `nranks` handles
 * (one process, ONE GPU, one host thread per handle).  Same row partition and the same code paths as msdp_comm_init --
 * row offsets, replicated operator state, lock-step tCG, order and number of collective calls -- with the three
 * collectives carried out by a host barrier and device copies / a summation kernel between the members' buffers, so a
 * single GPU executes the N-rank paths.  A member that never makes the matching call breaks the group after 120 s
 * (MSDP_ECOMM; environment variable MSDP_LOCAL_BARRIER_TIMEOUT = seconds, for legitimately slow members on a loaded GPU)
 * instead of hanging the process; a member that detects an error itself breaks the group at once. */
int msdp_comm_init_local(msdp_handle h, int32_t nranks, int32_t rank, int32_t group_id);
/* The same group with its members in DIFFERENT PROCESSES -- several ranks on one GPU, or one rank per GPU of a node with peer
 * access (replaces nothing in the reference, whose MATLAB path is one process).  `name` is a POSIX shared-memory
 * name ("/..."), the same on every member and fresh for every group.  Rank 0 allocates one fine-grained device block (the slot
 * regions of the cross-rank persistent tCG, one staging slab per rank for the collectives) and EVERY member an exchange buffer
 * for its own rows plus a slot per foreign row it references, on its own device; the blocks are exported with hipIpcGetMemHandle
 * and mapped by the others with hipIpcOpenMemHandle (peer access between devices).  With sparse C the whole
 * tCG of a trust-region iteration then runs as ONE grid-synchronised computation across the members' launches -- no collective
 * per trip; a member whose launch never arrives turns the others' bounded spins into MSDP_ECOMM.  Collectives outside the tCG
 * (cost / gradient at the proposal, AL bookkeeping) go through the staging slabs behind a barrier on the shared segment. */
int msdp_comm_init_ipc(msdp_handle h, int32_t nranks, int32_t rank, const char* name);
/* Local row range [row0, row1) of this rank. */
int msdp_local_rows(msdp_handle h, int64_t* row0, int64_t* row1);

/* ------------------------------------------- AL bookkeeping */

/* Affine handles, at the resident point (x = vec(YY')): obj = c'x and Ax = A x (m doubles;
 * the caller forms Axb = Ax - b) -- ManiSDP_unitdiag.m:59-62, ManiSDP_unittrace.m:59-62,
 * ManiSDP.m:58-63 without the n x n X on the host. */
int msdp_al_primal(msdp_handle h, double* obj, double* Ax);
/* Dual slack for the multipliers y: eS = reshape(c - At*y, n, n); unit diagonal:
 * z = sum(X.*eS) (n values), S = eS - diag(z) (ManiSDP_unitdiag.m:65-67); unit trace:
 * z = sum(eS.*X,'all') (1 value), S = eS - z*I (ManiSDP_unittrace.m:65-67); generic: S = eS
 * (ManiSDP.m:64; z untouched, may be NULL).  S stays on the device for msdp_escape_eigs_dual. */
int msdp_al_dual(msdp_handle h, const double* y, double* z);
/* lambda_min, lambda_max and <= k bottom eigenvectors of the S of the last msdp_al_dual call
 * (eig(S) of ManiSDP_unitdiag.m:68 / ManiSDP_unittrace.m:68 / ManiSDP.m:65). */
int msdp_escape_eigs_dual(msdp_handle h, int32_t k, double tol, int32_t maxit,
                          double* lam_min, double* V, double* lam_max, int32_t* iters);
/* The dense S (n x n, symmetric: column-major = row-major) of the last msdp_al_dual call: the reference's
 * data.S (ManiSDP_unitdiag.m:116, ManiSDP_unittrace.m:121) and the input of a host eig(S) for small n. */
int msdp_get_dual_slack(msdp_handle h, double* S);
/* The nb x nb diagonal block of the same matrix that starts at row / column row0 (column-major == row-major: symmetric).
 * ManiSDP_multiblock.m:78-88 takes eig(S_i) block by block: with a hundred blocks the whole N x N matrix is 100 times what
 * the host reads (N = 21 100 for example_bqp_sparse.m with t = 100: 3.5 GB per outer iteration against 35 MB). */
int msdp_get_dual_slack_block(msdp_handle h, int64_t row0, int64_t nb, double* S);
/* eig of `nb` diagonal blocks of the same matrix on the device, in one launch: what ManiSDP_multiblock.m:78-88 computes block by
 * block with eig(S{i}) -- all eigenvalues (dinf needs the extremes, the escape the count of negative ones, :129-133) and the
 * eigenvectors of the `k` smallest (:137-147 take at most options.delta of them).  Block b = rows / columns row0[b] .. row0[b] +
 * nblk[b] - 1 (a block of the handle when it stores per block); orders up to 256 (MSDP_EUNSUPPORTED beyond: the host loop over
 * msdp_get_dual_slack_block remains).  w: the eigenvalues, block after block, ascending inside a block (sum nblk values);
 * V: (sum nblk) x k row-major, row = position in the concatenation of the blocks, column c = eigenvector of the block's c-th
 * smallest eigenvalue (zero columns beyond a block's order).  One workgroup per block (msdp_blockjacobi.hip); method 0 = Householder
 * tridiagonalisation + bisection + inverse iteration when k <= 8 (the way of LAPACK's dsyevx), else 1 = cyclic Jacobi; 2 = the former or an error. */
int msdp_block_eigs(msdp_handle h, int32_t nb, const int64_t* row0, const int64_t* nblk, int32_t k, int32_t method, double* w, double* V);

/* Run-time switches of one handle (production = the defaults; the tests and the profiling scripts use them):
 *   "persist"      1/0  persistent single-launch tCG / Lanczos kernels where they fit (default 1; env MSDP_NO_PERSIST=1)
 *   "fused_rtr"    1/0  whole trustregions() loop in one launch for p <= 32          (default 1; env MSDP_NO_FUSED_RTR=1)
 *   "graph"        1/0  chunked tCG trips replayed as hipGraphs                      (default 1; env MSDP_NO_GRAPH=1)
 *   "affine_route" 0 = choose by bytes moved, 1 = SDDMM, 2 = Gram                    (default 0; env MSDP_AFFINE_ROUTE)
 *   "escape_deflate" 1/0 escape: deflate span(Y) at near-stationary points (default 1).  Fast, but only as accurate as
 *                       S*Y is small; a caller about to DECLARE optimality re-checks lambda_min with 0 (see solvers.py)
 *   "escape_warm"  1/0  escape: start from what the previous call found (default 1; 0 = hashed random start vector)
 *   "persist_refresh" k  persistent tCG kernel: every k-th trip exchanges the rows of the new direction itself (one extra
 *                       barrier) so that the product C*mdelta, otherwise assembled by linearity, starts afresh (default 32;
 *                       0 = never: |Heta - Hess(eta)|/|Heta| then grows to 1e-8 over 100 trips on G81)
 *   "persist_early" k  persistent tCG kernel (rows of <= 5 entries, p <= 32): k >= 1 = the neighbours' rows of tangent(r') are
 *                       gathered WHILE the second grid reduction of the trip is in flight -- the exchange buffer's halves hold a NaN
 *                       sentinel until a row is stored, so the rows are their own flags (k - 1 = units of 64 cycles a wave sleeps
 *                       between posting the reduction and its first gather); 0 = at the top of the next trip, behind that
 *                       reduction (the round-4 trip).  Same arithmetic, same decisions; measured slower (default 0)
 *   "persist_pipe" 1/0  persistent tCG kernel (rows of <= 8 entries, or CSR rows with "persist_ep"; p <= 32): ONE grid reduction per trip instead of two -- the
 *                       values of tCG.m:227-241 (model value, <r', r'>) follow from eight inner products formed BEFORE the step length
 *                       is known, the neighbours gather the rows of H*mdelta, and C*tangent(r), C*mdelta follow by linearity.
 *                       Same tests and decisions as tCG.m; <r', r'> and the model value that decide a trip carry a rounding error of
 *                       eps <r, r> / <r', r'> (the directly summed values replace them one trip later).  G81, p = 32: 6.6 -> 4.9 us
 *                       per trip (default 1; 0 = the two-reduction trip; env MSDP_NO_PERSIST_PIPE=1)
 *   "pipe_refresh" k   one-reduction trip: every k-th trip also publishes tangent(r) and mdelta, and the next one forms both
 *                       products from direct gathers (default 16: |Heta - Hess(eta)|/|Heta| <= 1.2e-11 after 100 trips on G81;
 *                       32: 6e-11, 8: 2.5e-12; the recurrences lose accuracy with the SQUARE of k)
 *   "affine_overlap" 1/0  affine kinds: the 2*eS*U contraction of a Hess-vec runs on a second stream beside the A(.) / A'(.)
 *                       chain (default 0: measured slower than one stream; kept for A/B timing; results agree to rounding)
 *   "trip2"        0/1/2  chunked path, sparse C / oblique manifold / one rank: two launches per tCG trip (12 vector passes,
 *                       msdp_trip2.hip) instead of three (17 passes): 1 = where it pays, from 2^21 vector entries on (default);
 *                       2 = always (tests); 0 = never
 *   "escape_method" 0 = block eigen-solver where it applies (sparse C, n >= 512), Lanczos otherwise (default); 1 = Lanczos
 *                       always; 2 = block also for small n (tests)
 *   "be_width" 0/32/64/128, "be_degree", "be_grid", "be_lpr"  block eigen-solver: panel width, filter degree per round,
 *                       workgroups and lanes per row of the filter step (0 = automatic; measurement and tests)
 *   "escape_start_y" 1/0  undeflated cold-start escape runs start from a random combination of the columns of Y plus 5 %
 *                       noise instead of pure noise (default 0; the independent lambda_min check of the host loops sets it)
 *   "lanczos_onesync" 1/0  undeflated persistent Lanczos runs use one grid synchronisation per step (default 1; 0 = the
 *                       two-synchronisation kernel the deflated runs use)
 *   "halo_exchange" 1/0  row-sharded sparse C (after msdp_comm_init*): before S*U every rank receives only the rows of the
 *                       direction its rows of C reference (grouped ncclSend / ncclRecv) instead of all rows (ncclAllGather,
 *                       default 0).  Bit-identical results; msdp_get_point_all and the escape keep the all-gather
 *   "lanczos_qglobal" 1/0  deflated persistent Lanczos runs read the deflation columns in place instead of from their LDS
 *                       copy (default 0: in place only where the copy does not fit, n > ~117 000 with 40 columns; bit-identical)
 *   "block_skip"   1/0  multiblock kind: the dense contraction skips the zero off-diagonal blocks of its operands (default 1;
 *                       0 = stream the whole N x N matrices; same results, for A/B timing)
 *   "dense_pack"   1/0  dense C*U reads the MFMA-fragment-ordered copy of C (default 1; 0 = the row-major one;
 *                       bit-identical results, for A/B timing)
 *   "persist_ep"   1/0  persistent tCG kernel on CSR rows (rows of more than 8 entries): the 64 / lanes-per-row lane groups of a
 *                       wave share ONE row and split its entries where the grid leaves lanes free -- in that form the one-reduction trip
 *                       ("persist_pipe") takes CSR rows too (G1: 12.5 -> 5.5 us per trip at p <= 16; default 1; 0 = one lane group per
 *                       row, two reductions per trip).  Same row arithmetic; the row's products are summed in another order
 *   "dense_sym", "dense_sym_min", "dense_sym_rt", "dense_sym_db", "dense_sym_len", "dense_sym_res"  symmetric dense contraction
 *                       (msdp_densesym.hip): on from dense_sym_min rows (1, default) / always (2) / never (0); workgroup shape 1..4 =
 *                       8 x 16, 8 x 32, 16 x 16, 12 x 32 rows (0 = by p and n); one or two barriers per step; slice length; workgroups
 *                       assumed resident when the slices are cut (A/B timing and tests; results agree to rounding, every shape is
 *                       bit-reproducible run to run)
 *   "timing", "esc_debug"  1/0  diagnostics on stderr                                (env MSDP_TIMING, MSDP_ESC_DEBUG)
 *   "debug_fail_persist"   1    test hook: the next persistent launch reports a synchronisation time-out
 * The environment variables are read once, when the handle is created.  Unknown names -> MSDP_EINVAL. */
int msdp_set_option(msdp_handle h, const char* name, int32_t value);

/* Which implementation msdp_rtr uses for the tCG inner loop at the resident point:
 * 1 = persistent single-launch kernel (working set in registers/LDS, sparse C, oblique,
 * one rank, n and p small enough to stay on chip), 0 = chunked hipGraph of three kernels
 * per trip.  Both follow tCG.m:95-292; the choice is a speed matter only. */
int msdp_tcg_path(msdp_handle h, int32_t* path);
/* (test / diagnostic) The trip form of the persistent kernel at the resident point: 2 = ONE grid reduction per trip (option
 * "persist_pipe", rows of <= 8 entries or shared CSR rows, p <= 32, every vector in registers), 1 = the "persist_early" form, 0 = two reductions per
 * trip (tCG.m:166 and :227-241 separately); -1 = the tCG is not persistent.  All forms follow tCG.m:95-292. */
int msdp_debug_persist_form(msdp_handle h, int32_t* form);

/* ------------------------------------------------------------ measurement */

/* Launch the Hess-vec kernel `reps` times on the library's stream between two
 * HIP events and return the average device time per launch (ms) together with
 * the algorithmic bytes/flops of one launch (SURVEY.md 8d formulas). */
int msdp_bench_hessvec(msdp_handle h, int32_t reps, double* avg_ms,
                       double* algo_bytes, double* algo_flops);
/* One kernel of the tCG trip in isolation: which = 0 Hess-vec, 1 upd1 (tCG.m:166-241),
 * 2 upd2 (tCG.m:249-287); average device time per launch in ms. */
int msdp_bench_kernel(msdp_handle h, int32_t which, int32_t reps, double* avg_ms);
/* Same for one whole tCG trip (Hess-vec + the vector updates), exits disabled. */
int msdp_bench_tcg_trip(msdp_handle h, int32_t reps, double* avg_ms);

const char* msdp_last_error(void);
const char* msdp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MANISDP_HIP_H */
