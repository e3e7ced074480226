// msdp_common.h -- internal structures shared by the HIP translation units of
// libmanisdp_hip.so.  gfx950 only (wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/manisdp_hip.h"

#define MSDP_BLOCK 1024           // threads per workgroup (16 waves: one row step per wave, latency hidden by occupancy)
#define MSDP_WAVES (MSDP_BLOCK / 64)
#define MSDP_MAX_GRID 512         // <= 512 partial sums per reduction (2 per CU)
#define MSDP_NPART 12             // number of partial-sum arrays

void msdp_set_error(const char* fmt, ...);

// Host <-> device copies (msdp_xfer.hip): signatures of hipMemcpy / hipMemcpyAsync / hipMemcpy2D / hipMemcpy2DAsync.  Unpinned caller
// memory goes through a pinned staging buffer of the process -- no msdp_*.hip file calls the hipMemcpy family itself.
hipError_t msdp_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t msdp_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t msdp_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t msdp_memcpy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
void msdp_xfer_release();

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            msdp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                   \
            return MSDP_EHIP;                                                     \
        }                                                                         \
    } while (0)

// RTR-level scalars; written only by single-workgroup kernels that run alone
// between vector kernels (no intra-launch hazards).
struct Ctl {
    double fx, fx_prop, gg, gg_prop, norm_grad, Delta, Delta_bar, Delta0;
    double tolgradnorm, kappa, theta, rho_prime, rho_reg;
    double rho, rhonum, rhoden;
    double z_sphere[2];          // unittrace: store.z per slot
    double sigma;                // AL penalty (affine kinds)
    int maxiter, maxinner, mininner;
    int k, cur, done, stop_reason;
    int hessvecs, accepted, rejected, cost_evals, last_stop_inner;
    int bench_mode;              // 1: tCG exits disabled (throughput measurement)
    int tcg_running;             // mirror of Frame.active for host polling
    int psync8_backoff;          // the eight-value reduction of the one-reduction trip (msdp_pipe.h): units before its first poll where bits 16..23 of psync_backoff are 0
    int psync_backoff;           // grid reductions of the persistent kernels: (s_sleep units of 64 cycles before the first poll) | (units after a failed poll) << 8
    int persist_goff;            // persistent tCG: gather offsets kept in registers (A/B)
    int pipe_local;              // one-reduction trip: 1 = rows of the own workgroup from LDS / registers, 0 = every row through the exchange buffer
    int pipe_refresh;            // one-reduction trip (msdp_pipe.h): every pipe_refresh-th trip the products start afresh from direct gathers
    int persist_early;           // persistent tCG: 0 = gather the neighbours' rows at the top of the trip (round 4), >= 1 = while reduction 2 is in
                                 //   flight, the rows being their own flags (msdp_persist.hip EARLY); value - 1 = s_sleep units before the first gather
    int persist_refresh;         // persistent tCG (two-synchronisation trips): every this-many trips the product C*mdelta is
                                 //   gathered directly instead of assembled by linearity (0: never); msdp_persist.hip
};

// tCG scalars (tCG.m:102-157, 286-287); two frames, each kernel reads one frame and
// writes the other so no workgroup ever reads a word another workgroup writes in
// the same launch.
struct Frame {
    double z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta;
    int active, j, stop, eta_idx;
    int md_idx;       // which of md / md2 holds the current direction (fused trips ping-pong it)
    int fresh;        // 1 right after k_tcg_init: the direction is the gradient, no update step yet
};

// Partial-sum array ids
// P_AXB: |A x - b - y/sigma|^2 of the affine kinds, summed over the constraints by MSDP_MAX_GRID workgroups (every other
// array is filled by the d.G workgroups of a row-parallel launch)
enum { P_F = 0, P_GG = 1, P_DHD = 2, P_S1 = 3, P_S2 = 4, P_S3 = 5, P_RD = 6, P_AUX = 7, P_AXB = 8,
       // sphere Hess-vec, fused form (msdp_affine.hip, k_sddmm1 mode 2): <U, G>, <U, Y>, sum_k w_k (A x)_k
       P_T1 = 9, P_T2 = 10, P_T3 = 11 };

enum { MANI_OBLIQUE = 0, MANI_SPHERE = 1, MANI_EUCLID = 2 };
enum { COST_SPARSE = 0, COST_DENSE = 1, COST_AFFINE = 2 };

// Everything a kernel needs, passed by value.
struct Dev {
    int n;            // global number of points (matrix order)
    int n_loc;        // rows owned by this rank
    int row0;         // global index of local row 0
    int p;            // current factor width
    int ld;           // row stride in doubles (even, >= p; pad columns are zero)
    int G;            // workgroups per row-parallel launch (multiple of 8, <= MSDP_MAX_GRID)
    int manifold;     // MANI_*
    int costkind;     // COST_*
    Ctl* ctl;
    Frame* F;         // F[0], F[1]
    double* P;        // MSDP_NPART arrays of MSDP_MAX_GRID doubles
    double* Y[2];     // point per slot            (n_loc x ld)
    double* Gr[2];    // Riemannian gradient / slot (n_loc x ld)
    double* eG[2];    // per-row scalar / slot: eG (onlyunitdiag), YeG (unitdiag)  (n_loc)
    double* eta[2];
    double* Heta[2];
    double* r;
    double* r2;       // second residual buffer (two-launch trips, msdp_trip2.hip: eta and r ping-pong together)
    double* md;       // mdelta, local rows
    double* md2;      // second direction buffer (fused two-launch trips)
    double* mdx;      // persistent tCG: exchange buffer of the direction rows (uncached memory, sc1 accesses only)
    double* Hmd;
    double* full;     // gather source of n x ld (== local buffer when nranks == 1)
    double* W0;       // scratch n_loc x ld
    double* W1;
    // msdp_trip1.hip: this rank's three sums of a trip (xs[0..2]), all ranks' (xs_all[4*q + 0..2], filled by the exchange), the
    // arrival counter of the launch that forms xs, the number of ranks
    double* xs; double* xs_all; unsigned* xcount; int xn;
    int sweep;        // gather kernels walk the rows window by window (msdp_sweep_rows, msdp_device.h) instead of chunk by chunk
    // sparse C (local rows, global column indices)
    const int* rowptr;
    const int* colind;
    const double* cval;
    int64_t nnz;
    // optional ELL copy of the same rows (max row length <= 8): [w][row] slices
    int ellW;
    int64_t ell_stride;
    const int* ellc;
    const double* ellv;
    // dense C / eS (n_loc x n row-major) per slot for affine kinds
    double* Cd;       // dense cost matrix rows (COST_DENSE) or c reshaped (COST_AFFINE)
    double* Cpk;      // COST_DENSE: the same rows in MFMA-fragment order (msdp_dense.hip, k_pack_fragments)
    double* eS[2];    // affine kinds: eS per slot (n_loc x n)
    double* AyU;      // affine kinds: A'(A(.)) scratch (n_loc x n)
    double* Sdual;    // affine kinds: dual slack S of msdp_al_dual (= AyU unless the restricted adjoint is in use)
    // affine operator: At in CSC (by constraint) and CSR (by matrix entry)
    int64_t m;
    const int64_t* at_jc; const int64_t* at_ir; const double* at_pr;     // CSC n^2 x m
    const int64_t* a_rp;  const int* a_ci;  const double* a_v;           // CSR by entry (n^2 rows)
    const double* b; double* yv; double* Axb[2]; double* w;              // length m
    double* Pm;       // partial sums for m-length reductions
    unsigned long long* status;   // host-mapped progress word: (TR iteration+1) << 32 | tCG j << 1 | active
    // multiblock kind: rowfree[i] != 0 marks the rows of the blocks that carry no unit-diagonal constraint (Euclidean
    // factor of the product manifold: no projection term, no normalisation); nullptr = every row is oblique
    const unsigned char* rowfree;
    // multiblock kind: [blk_lo[i], blk_hi[i]) = the rows (= columns) of the diagonal block row i belongs to.  Every dense
    // n x n operand of that kind (cost, eS, AyU, S) is block diagonal: the contraction skips the k range outside a row
    // tile's blocks (exact zeros there).  nullptr for every other kind.
    const int* blk_lo; const int* blk_hi;
    // msdp_debug_persist_trace: s_memtime stamps of the phases of the persistent tCG trip (traced kernel instance only), else null
    unsigned long long* trace;
    // cross-rank persistent tCG (msdp_persist.hip, XR): this rank's first workgroup index among all ranks' workgroups, their total,
    // the members' exchange buffers (sc1 accesses only)
    int xr_gid0, xr_gtot;
    // round 5: every member keeps the exchanged rows of ITS rows in its OWN buffer (local HBM on its device; the others map it through
    // HIP IPC / peer access, or simply share the address space): xr_rows[q] = member q's buffer as this process sees it, xr_cap rows each
    // (the row capacity of a rank), xr_me = this member.  A gather of row c goes to member c / xr_cap -- its own for most rows of a
    // partition with locality.
    double* xr_rows[4];
    int xr_cap, xr_me, xr_halo;
    // "push" exchange (msdp_api.hip halo_setup): a member's buffer = [its rows (xr_cap)] [one slot per foreign row its rows of C
    // reference]; buffer-local column indices of its rows (ELL copy [w][xr_cap] and CSR), and for every local row up to two
    // (member, position there) pairs it has to be stored to as well ([2][n_loc], member -1 = none)
    const int* xr_ellc; const int* xr_colind; const int* xr_pq; const int* xr_pidx;
    const unsigned long long* xr_paddr;   // [2][n_loc] address of local row i's slot in the buffer of a member that references it (0: none); built by msdp_xpersist_member
    // round 6, two-level grid reductions between members that own a device each (msdp_psync.h psync2): xr2_on, this member's block
    // (fine-grained memory of its device), the members' blocks as this process maps them (device array of xr2_n pointers), this member's
    // index among the xr2_n members; xr_sys: the pushed rows cross devices (system-scope stores)
    int xr_pipe;          // the members' launches run the one-reduction trip (msdp_pipe.h XRM) -- process ranks, rows of <= 5 entries, <16,5,3> / <8,5,2> plans
    int xr2_on, xr2_n, xr2_me, xr_sys, xr2_skip;   // (xr2_skip: test hook -- the leader waits for this many local workgroups that do not exist)
    unsigned long long* xr2_blk;
    unsigned long long* const* xr2_peers;
    int persist_slots;    // A/B: row slots of the persistent tCG plan at p = 17..32 (0: planned)
    int persist_ep;       // CSR rows: the lane groups of a wave share a row and split its entries where the grid allows (1, default) or never (0)
};

// Run-time switches of a handle (msdp_set_option; the environment variables of the same meaning are read ONCE, when
// the handle is created).  Production = the defaults.
struct Tuning {
    int persist = 1;       // persistent single-launch tCG / Lanczos kernels where they fit        (MSDP_NO_PERSIST=1 -> 0)
    int fused_rtr = 1;     // whole trustregions() loop in one launch for p <= 32                  (MSDP_NO_FUSED_RTR=1 -> 0)
    int graph = 1;         // chunked tCG trips replayed as hipGraphs                              (MSDP_NO_GRAPH=1 -> 0;
                           //   rocprofv3 7.2 crashes on graph replay)
    int affine_route = 0;  // A(Ya Yb'): 0 = by bytes moved, 1 = SDDMM, 2 = Gram                   (MSDP_AFFINE_ROUTE=sddmm|gram)
    int timing = 0;        // per-call timing lines on stderr                                       (MSDP_TIMING=1)
    int esc_debug = 0;     // per-run Lanczos statistics on stderr                                  (MSDP_ESC_DEBUG=1)
    int escape_deflate = 1;  // escape: deflate span(Y) at near-stationary points (fast, approximate when S*Y != 0)
    int escape_start_y = 0;   // undeflated cold-start runs begin in span(Y) + 5 % noise (the independent lambda_min check sets it)
    int halo_exchange = 0;    // row-sharded sparse C: exchange only the referenced rows before S*U (0: the all-gather north_star prescribes)
    int block_skip = 1;       // multiblock kind: the dense contraction skips the k range outside a row tile's diagonal blocks (0: A/B, tests)
    int lanczos_qglobal = 0;  // tests: deflated persistent Lanczos reads the deflation columns in place even where they fit the LDS
    int lanczos_onesync = 1;  // undeflated persistent Lanczos runs: one grid synchronisation per step (0: two)
    int dense_pack = 1;    // dense C*U reads the fragment-ordered copy of C (0: the row-major one; same results)
    int affine_fuse = 1;   // affine kinds, SDDMM route, one rank: A(Ya Yb') and its finish in ONE launch (k_sddmm1); sphere Hess-vec: the
                           //   sparse A'(w)*Y product, the slab sum and the projection in ONE launch (k_sph_hess_fused) (0: A/B, tests)
    int affine_side = 1;   // sphere / Euclidean Hess-vec, SDDMM route, few touched entries: the SDDMM rides in the contraction launch as a side job
                           //   (msdp_dense_gemm_side) instead of a launch of its own (0: A/B, tests)
    int affine_broute = 1; // affine Hess-vec, Gram route, symmetric data with short constraints: A'(A(.)) as one sparse matrix on the Gram matrix
                           //   (k_adjoint_gram) instead of k_gram_apply + k_adjoint_tiled (0: A/B, tests)
    int dense_sym = 1;     // symmetric dense operands (p <= 32, one rank): the contraction reads the upper triangle only
                           //   (msdp_densesym.hip); 1 = from dense_sym_min rows on, 2 = always, 0 = never
    int dense_sym_min = 8192;   // see dense_sym (below, the partial slabs of the transposed products cost what the halved matrix saves: measured)
    int dense_sym_rt = 0;       // A/B: 16-row tiles per wave of k_dense_sym (0: by n; 1 or 2)
    int dense_sym_db = 0;       // A/B: double-buffered reduction, one barrier per step (0: by shape, 1: never, 2: always)
    int dense_sym_len = 0;      // A/B: slice length of a work item in 16-column steps (0: planned)
    int dense_sym_res = 0;         // workgroups of the symmetric contraction assumed resident at a time when its slices are cut (0: by shape)
    int escape_warm = 1;     // escape: start the Lanczos runs from what the previous call found (0: hashed random vector)
    int xpersist = 1;        // in-process ranks (msdp_comm_init_local), sparse C, oblique: ONE persistent tCG spanning the ranks' launches -- grid
                             //   reductions and row exchange through shared uncached memory, no collective per trip (msdp_persist.hip XR); 0: lock-step chunks
    int xtail = 1;           // process ranks (msdp_comm_init_ipc): the rest of a TR iteration behind the cross-rank tCG is one launch per member as well
                             //   (retraction, cost / gradient at the proposal, decision; msdp_trtail.hip XR); 0: sharded kernels with their collectives
    int psync8_backoff = 16;   // eight-value reduction of the one-reduction trip: units slept before the first poll (round 6: 16 against 19, 0.8 % of a G81 call)
    int psync_backoff = 19;  // grid reductions of the persistent kernels: s_sleep units (64 cycles) before the first poll | units after a
                             //   failed poll << 8.  Round 4: nothing can be visible for the first half microsecond after the posts, and
                             //   the polls of 216 workgroups are the traffic the posts compete with -- G81, p = 32: 8.03 us per trip
                             //   with 0, 7.06 (14), 6.85 (18), 6.87 (20), 7.05 (24), 7.44 (32), 7.86 (40); p = 16: 6.87 -> 5.84
                             //   (tools/psync_backoff_probe.py); sleeping between failed polls gains nothing
    int window = 1;            // stand-alone sparse S*U with the gathered rows staged once per workgroup in LDS (msdp_window.hip): 1 = for vectors
                               //   far beyond the L2s (from 3 * 2^22 entries on), 2 = always (tests), 0 = never
    int window_lds = 144;      // ... KB of LDS a window may take (A/B; one 1024-thread workgroup per CU)
    int persist_refresh = 32;  // persistent tCG: direct (three-synchronisation) trip every this-many trips, bounds the drift of C*mdelta
    int persist_slots = 0;     // A/B: row slots per lane group of the persistent tCG at p = 17..32 (0: planned; 3 or 4)
    int persist_ep = 1;        // CSR rows of the persistent tCG: entry-parallel lanes where the rows leave lanes free (A/B: 0 never)
    int persist_goff = 1;      // persistent tCG: the byte offsets of the R x EW gathers of a trip live in registers (0: recomputed per trip from the LDS copy of the column indices)
    int pipe_local = 1;        // one-reduction trip: neighbours that belong to the same workgroup are read from LDS, the diagonal from registers
    int xr_twolevel = 0;       // process ranks: 1 = two-level grid reductions (msdp_psync.h psync2) also where the flat ones would do (one device, <= 4 members); A/B, tests
    int pipe_refresh = 16;     // one-reduction trip: trips between two direct exchanges (its recurrences drift with the square of this)
    int persist_pipe = 1;      // persistent tCG: ONE grid reduction per trip (msdp_pipe.h) where an instance exists (rows of <= 8 entries, p <= 32)
    int persist_early = 0;     // persistent tCG: the neighbours' rows are gathered while reduction 2 is in flight -- sentinel-initialised exchange
                               //   halves, the rows are their own flags (0: at the top of the next trip, behind reduction 2 -- the round-4 trip;
                               //   n >= 1: n - 1 s_sleep units between the post of reduction 2 and the first gather)
    int affine_overlap = 0;  // affine Hess-vec: 2*eS*U on a second stream beside the A(.) / A'(.) chain.  Measured SLOWER (round 3: BQP d = 60
                             //   85 against 73 us, theta n = 5000 83 against 74 us per Hess-vec inside graph replays): every launch of the chain
                             //   already fills the chip, the fork / join only adds dependencies.  Kept as an A/B switch, default off.
    int dense_sk = 0;        // A/B switch: number of k slices of the dense contraction (0 = the plan's choice)
    int sweep_k = 2;         // 64-row steps a workgroup takes per window of the traversal (window = 32 * 64 * sweep_k rows per XCD)
    int sweep = 1;           // windowed row traversal of the large-vector gather kernels: 1 = from 2^21 vector entries per rank on (streaming accesses from 3 * 2^22 on), 2 always / without streaming accesses, 3 always / with them, 0 never
    int trip1 = 1;           // msdp_trip1.hip (sparse C / oblique): row-sharded handles -- one exchange + one all-reduce per tCG trip instead of one + two;
                             // one rank, chunked path -- the same two launches (14 vector passes, one gathered vector); 0: never
    int trip2 = 1;           // chunked path, sparse C / oblique / one rank: two launches per tCG trip (msdp_trip2.hip) instead of three
    int escape_method = 0;   // 0: block Chebyshev-filtered subspace iteration where it applies (msdp_blockeig.hip), else Lanczos;
                             //   1: Lanczos always (msdp_escape.hip); 2: block also below its size threshold (tests)
    int be_width = 0;        // block eigen-solver: panel width 32 / 64 / 128 (0: 64, or 128 when the start block needs it)
    int be_degree = 0;       // ... filter degree per round (0: 200 warm, 400 cold)
    int be_grid = 0;         // ... workgroups of the filter step (0: by rows)
    int be_lpr = 0;          // ... lanes per row of the filter step (0: by rows per workgroup)
    int grid = 0;          // workgroups of the row-parallel launches (0: choose_grid; A/B switch)
    int fail_block = 0;    // test hook (msdp_set_option "debug_fail_block"): the next block eigen-solver call reports itself unconverged
    int fail_xr = 0;       // test hook ("debug_xr_skip"): this member skips its next cross-rank persistent launch
    int fail_persist = 0;  // test hook (msdp_set_option "debug_fail_persist"): the next persistent launch reports a
                           //   grid-synchronisation time-out without running, to exercise the recovery path
};

struct msdp_handle_s {
    int kind = 0;
    Tuning tune{};
    bool persist_failed = false;   // a persistent launch timed out on this handle: stay on the chunked path
    double* rtr_start = nullptr;   // copy of the start point of the running msdp_rtr call (persistent path: recovery)
    size_t rtr_start_cap = 0;
    Dev d{};
    int pcap = 0;
    int ldcap = 0;
    bool have_point = false;
    bool state_valid = false;      // cost/grad state computed at the resident point
    bool gradnorm_valid = false;   // h_ctl->norm_grad / fx describe the resident point
    bool dual_valid = false;       // d.Sdual holds the dual slack S of the last msdp_al_dual call
    double* vec_pool = nullptr;    // the one allocation behind Y, Gr, eta, Heta, r, md, Hmd, W0, W1 (msdp_alloc_vectors)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    Ctl* h_ctl = nullptr;          // pinned host mirror
    Frame* h_frame = nullptr;
    std::vector<void*> allocs;     // everything to hipFree
    // host copies needed for re-allocation / sharding
    int nranks = 1, rank = 0;
    void* comm = nullptr;          // ncclComm_t
    bool presharded = false;       // created per shard (dense synthetic): row0/n_loc fixed at creation
    double* full_buf = nullptr;    // gather buffer (nranks x cap rows) when the rows are sharded
    // halo exchange (sparse C, option "halo_exchange"): instead of all rows of the direction every rank receives only the rows
    // its rows of C reference (msdp_api.hip, "Halo exchange")
    struct Halo* halo = nullptr;
    struct LocalGroup* lgroup = nullptr;   // in-process stand-in for the RCCL communicator (msdp_comm_init_local)
    unsigned long long* xr_paddr = nullptr; size_t xr_paddr_cap = 0; double* xr_paddr_key[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; int xr_paddr_ld = 0, xr_paddr_n = 0; const int* xr_paddr_pq = nullptr;   // cross-rank push addresses and what they were built from
    bool lgroup_is_ipc = false;                  // the handle's group is a process group (msdp_comm_init_ipc)
    unsigned long long* xr2_blk = nullptr;       // this member's two-level block (its own allocation), the members' blocks as mapped here (device array)
    unsigned long long** xr2_peers = nullptr;
    int xr2_share = 1; bool xr2_multi = false;   // process ranks (msdp_comm_init_ipc): the most members on one device, members on different devices
    bool xr_ok = false; int xr_halo_rows = 0;   // push exchange of the cross-rank kernels: usable (no row needed by more than two members), foreign rows referenced
    struct WinCache* win = nullptr;        // patch plans of the LDS-staged S*U (msdp_window.hip), one per lanes-per-row
    void* blk_ws = nullptr; size_t blk_ws_cap = 0;   // workspace of msdp_block_eigs (msdp_blockjacobi.hip)
    double* lc_tmp = nullptr; size_t lc_tmp_cap = 0;   // its reduction scratch
    // row-sharded onlyunitdiag (sparse C): the escape runs replicated on full copies of C's CSR arrays and of z
    int* esc_rp = nullptr; int* esc_ci = nullptr; double* esc_cv = nullptr; double* esc_z = nullptr;
    double* yfull[2] = {nullptr, nullptr};   // row-sharded affine kinds: all rows of Y[slot] (the A(.) operator and the dense
                                             //   contraction read every row; msdp_affine.hip, "Row sharding")
    bool use_comm = false;         // route the exchanges through RCCL (set by msdp_comm_init, any nranks)
    std::vector<int> h_rowptr; std::vector<int> h_colind; std::vector<double> h_cval;
    // device-side sparse arrays owned by the handle
    int* d_rowptr = nullptr; int* d_colind = nullptr; double* d_cval = nullptr;
    int* d_ellc = nullptr; double* d_ellv = nullptr;
    msdp_rtr_opts last_opts{};
    double last_rtr_device_ms = -1.0; bool last_rtr_fused = false;   // msdp_debug_last_rtr_device_ms
    // tCG chunk graph (CH x {hess, upd1, upd2}) and its validity signature
    hipGraphExec_t chunk_exec = nullptr;          // the one to launch now (alias into chunk_execs)
    hipGraphExec_t chunk_execs[2] = {nullptr, nullptr};
    Dev chunk_sig{};
    int chunk_len = 0;
    volatile unsigned long long* h_status = nullptr;   // host view of Dev::status
    volatile int* h_flags = nullptr;                   // pinned copies of ctl->tcg_running, one per chunk in flight (lock-step path)
    hipEvent_t ev_flag[2] = {nullptr, nullptr};
    double* snap = nullptr;        // msdp_point_snapshot copy of the resident point
    size_t snap_cap = 0;
    int snap_p = 0;
    double* slab = nullptr;        // split-K partial slabs of the dense MFMA path
    bool blocked = false;          // multiblock kind with per-block storage: no N x N operand exists (msdp_affine.hip)
    bool dense_symmetric = false;  // every dense operand of the contraction (C, eS, AyU) is symmetric (checked at set-up)
    unsigned long long* trace_buf = nullptr;   // device buffer behind Dev::trace (msdp_debug_persist_trace)
    void* symplans = nullptr;      // work-item plans of the symmetric contraction (msdp_densesym.hip)
    size_t slab_cap = 0;
    // escape workspace, kept between calls (freeing 3 GB after every call stalled the NEXT kernels on the stream
    // for ~60 ms while the driver unmapped it: gaps seen in the kernel trace of the G81 solve)
    double* esc_mem = nullptr;
    size_t esc_cap = 0;               // doubles
    unsigned long long* lz_slots = nullptr;   // grid-sync slots of the persistent Lanczos kernel (uncached device memory)
    double* esc_prev = nullptr;       // sum of the bottom eigenvectors found by the previous escape call (warm start)
    int esc_prev_n = 0;
    // outcome of the last escape call (msdp_escape_info): real pairs returned, every Lanczos run passed a stop test,
    // largest relative residual |S x - theta x| / max(|theta|, |lam_max|) among the runs that hit maxit instead
    int esc_nvalid = 0, esc_converged = 0;
    double esc_maxres = 0.0;
    double esc_lower = 0.0;           // lower estimate of lambda_min(S) from the last call (-inf: none), see msdp_escape_lower_bound
    void* be = nullptr;               // workspace and warm-start state of the block eigen-solver (msdp_blockeig.hip)
    double* esc_top = nullptr;        // top eigenvector of the previous escape call (warm start of the lambda_max run), esc_top_n entries
    int esc_top_n = 0;
    long long coll_calls = 0;         // collective calls issued so far (exchange, all-reduce, all-gather; a grouped call counts once)
    hipEvent_t xr_ev = nullptr;       // marks this member's stream in front of the combined cross-rank launch
    bool xpersist_last = false;       // the last trustregions() call ran the cross-rank persistent tCG (msdp_tcg_path reports 2)
    bool trip1_capture = false;       // enqueue_trips runs inside a graph capture (no count-dependent launches)
    int trip1_count = 0;              // msdp_trip1.hip: trips enqueued since the tCG began (refresh schedule)
    int esc_method_last = 0;          // what the last escape call ran: 0 Lanczos, 1 block
    // persistent tCG kernel (msdp_persist.hip): grid-sync slots, error flag, cached eligibility
    unsigned long long* psync_slots = nullptr;
    int* psync_err = nullptr;
    int persist_sig_lpr = 0, persist_sig_ew = 0, persist_sig_r = 0, persist_sig_G = 0, persist_sig_ok = 0;
    const void* persist_sig_fn = nullptr; const void* fused_sig_fn = nullptr;     // the kernel instance the cached answers are for
    int fused_sig_lpr = 0, fused_sig_ew = 0, fused_sig_G = 0, fused_sig_ok = 0;
};

// --- launchers implemented in the .hip units (all asynchronous on h->stream) ---
int msdp_launch_costgrad(msdp_handle h, int slot);            // Y[slot] -> Gr[slot], eG[slot], P_F, P_GG
int msdp_launch_hess(msdp_handle h);
int msdp_launch_tcg_init(msdp_handle h);
int msdp_launch_upd1(msdp_handle h);
int msdp_launch_upd2(msdp_handle h);
#define MSDP_XS_MAX_RANKS 64
int msdp_exchange_rows_sums(msdp_handle h, const double* local_rows);    // msdp_api.hip
int msdp_trip1_ok(msdp_handle h);                             // msdp_trip1.hip: sharded trip with one all-reduce applies to this handle
int msdp_launch_trip1_init(msdp_handle h);
int msdp_launch_trip1_head(msdp_handle h, bool direct);
int msdp_launch_trip1_upd(msdp_handle h);
int msdp_trip2_ok(msdp_handle h);                             // msdp_trip2.hip: two-launch trip applies to this handle
int msdp_launch_trip2_init(msdp_handle h);
int msdp_launch_trip2_head(msdp_handle h);
int msdp_launch_trip2_upd(msdp_handle h);
int msdp_launch_retract(msdp_handle h);                       // Y[cur]+eta -> Y[1-cur], P_RD
int msdp_launch_rtr_begin(msdp_handle h);
int msdp_launch_rtr_decide(msdp_handle h);
int msdp_alloc_vectors(msdp_handle h, int pcap);
int msdp_persist_eligible(msdp_handle h);                     // msdp_persist.hip
int msdp_launch_tcg_persist(msdp_handle h, int reset_slots = 1);   // whole tCG of the current TR iteration, one launch
int msdp_launch_tr_tail(msdp_handle h);                       // retract + cost/grad at the proposal + accept/reject, one launch
int msdp_persist_fused_ok(msdp_handle h);                     // whole trustregions() loop in one launch possible?
int msdp_launch_rtr_fused(msdp_handle h);
size_t msdp_psync_bytes();
int msdp_allreduce_partials(msdp_handle h, int first, int count);   // no-op when nranks == 1
int msdp_allgather_rows(msdp_handle h, const double* local_rows);   // local -> d.full
int msdp_exchange_rows(msdp_handle h, const double* local_rows);    // sparse C: halo rows only when option halo_exchange is set, else the all-gather
