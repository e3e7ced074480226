// msdp_persist.hip -- the whole truncated-CG solve of one trust-region iteration in ONE launch
// (sparse C, oblique manifold, single rank): tCG.m:95-292 with the working set on chip.
//
// Why: at n = 20000, p = 32 every tCG vector is 5 MB and the whole working set (eta, Heta, r, mdelta,
// Hmdelta, Y, grad: 36 MB) fits in the register files + LDS of the 256 CUs (128 MB + 40 MB).  The
// three-launch trip (k_hess / k_tcg_upd1 / k_tcg_upd2) streams ~83 MB through HBM/MALL per trip and pays
// three launch gaps; here each workgroup keeps ITS rows of every vector in registers (eta, r, mdelta,
// Hmdelta) and LDS (Y, grad, eG and the ELL rows of C) for the whole solve.  The only global traffic per
// trip is the new direction (written once, gathered by the neighbours' S*U) and the partial sums of the
// two reductions (d_Hd; model value + r_r) plus one value-less barrier before the gathers.
// Variants: p = 33..64 keeps mdelta / Hmdelta in LDS and re-reads Y / grad from L2 (LOWREG); rows longer than
// 8 entries are walked in CSR form (EW = 0); FUSE = true (opt-in) runs the whole trustregions() loop here.
// The rest of a TR iteration (retraction, cost/gradient at the proposal, accept/reject) is msdp_trtail.hip.
//
// Grid synchronisation = deterministic all-to-all reduction: every workgroup stores its partial sums
// into its own slot of a generation buffer (agent-scope atomic store, 8 per-XCD replicas), then wave 0 of
// every workgroup polls the G slots of its replica until none holds the sentinel and sums them in the same fixed order (the order of
// msdp_sum_partials), so all workgroups take bit-identical decisions.  Three generation buffers rotate;
// a workgroup resets its slot of the previous generation after it has passed the current one (at that
// point every workgroup has finished reading it).  All G workgroups must be co-resident: G <= number of
// CUs and one 512-thread workgroup per CU (checked on the host with the occupancy API); a bounded spin
// turns a would-be hang into an error flag.
//
// Reference lines are quoted next to the statements (manopt7.0/manopt/solvers/trustregions/tCG.m);
// cost/projection expressions are those of ManiSDP_onlyunitdiag.m:127-156.
#include "msdp_device.h"
#include <math.h>
#include <cstdlib>

#include "msdp_psync.h"

size_t msdp_psync_bytes() { return 2 * PSYNC_REGION * sizeof(double); }       // regions A (tCG) and B (TR tail)

__global__ void k_psync_reset(unsigned long long* slots, int* err) {
    const int tot = (int)PSYNC_CNT_OFF;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x) { slots[i] = PSYNC_SENT; slots[PSYNC_REGION + i] = PSYNC_SENT; }
    if (blockIdx.x == 0 && threadIdx.x < 64) { slots[PSYNC_CNT_OFF + threadIdx.x] = 0ULL; slots[PSYNC_REGION + PSYNC_CNT_OFF + threadIdx.x] = 0ULL; }
    if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0;
}

// LPR lanes per row (one double2 per lane), EW = stored ELL width, R = row slots per lane group.
// TRACE (one instance, msdp_debug_persist_trace): thread 0 of every workgroup stamps s_memtime at the phase boundaries of the
// trips MSDP_TRACE_J0 .. J0 + MSDP_TRACE_NJ - 1 into d.trace[((workgroup * NJ + trip) * 8 + phase)]:
//   0 top of the trip (gathers about to be issued)   1 gathers + row arithmetic done   2 first grid reduction returned (d_Hd)
//   3 trial step formed, residual rows stored        4 those stores performed          5 second grid reduction returned
//   6 new direction formed (end of the trip)
// EARLY trips (round 5): 3 = trial step formed, rows stored, reduction 2 POSTED; 4 = last trip's half back to the sentinel, back-off
//   slept; 5 = the neighbours' rows gathered, none holds the sentinel (C*tangent(r') formed); 6 = second grid reduction returned;
//   7 = new direction formed (end of the trip)
#define MSDP_TRACE_J0 16
#define MSDP_TRACE_NJ 32
#define TSTAMP(ph) do { if (TRACE && threadIdx.x == 0 && j >= MSDP_TRACE_J0 && j < MSDP_TRACE_J0 + MSDP_TRACE_NJ) \
        d.trace[((size_t)bx * MSDP_TRACE_NJ + (j - MSDP_TRACE_J0)) * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
// XR (cross-rank, round 4): the launches of N ranks -- N x d.G workgroups, all co-resident -- run ONE tCG together: the grid
// reductions span the ranks (shared slot regions, workgroup index d.xr_gid0 + blockIdx.x of d.xr_gtot), the residual / direction rows
// travel through the members' exchange buffers (d.xr_rows[q]: member q's rows, in q's own memory), and no collective is issued per trip.  Same
// arithmetic per row as the one-rank kernel; the sums are formed over d.xr_gtot partials in index order on every rank (same bits on
// every rank -> same decisions).  Two slot regions alternate with the TR iteration; each launch clears the other one at its start.
// EP (round 6, CSR rows only): EP lane groups share a row.  The group with epi = 0 owns the row (its registers, its sums, its stores); all EP
// groups walk the row's entries with stride EP and their partial products are added across the groups -- a row of 48 entries is ONE
// round trip of six gathers per lane instead of six round trips of eight.  For the sizes where the rows do not fill the chip's lanes
// anyway (G1: 800 rows on 256 CUs).
template <int LPR, int EW, int R, bool FUSE, bool TRACE, bool XR, bool EARLYP, bool XR2 = false, int EP = 1>
__device__ __forceinline__ void tcg_persist_body(const Dev& d, unsigned long long* slots, int* err, const int bx) {
    static_assert(EP == 1 || (EW == 0 && !XR), "entry-parallel lanes: CSR rows of one rank");
    extern __shared__ double lds[];
    __shared__ double sh[3 * PWAVES];
    __shared__ double shb[8];
    constexpr int RPW = 64 / (LPR * EP);
    constexpr int RSTEP = PWAVES * RPW;       // rows per pass of the workgroup
    constexpr int ROWS = R * RSTEP;               // row slots of the workgroup
    double2* Ys = reinterpret_cast<double2*>(lds);                 // [R][PB]
    double2* Gs = Ys + R * PB;                             // [R][PB]
    double* eGs = reinterpret_cast<double*>(Gs + R * PB);  // [ROWS]
    double* vs = eGs + ROWS;                                       // [EW][ROWS]
    int* cs = reinterpret_cast<int*>(vs + EW * ROWS);              // [EW][ROWS]
    // FUSE only: the proposal point, its gradient and eG (the cost/gradient phase runs rolled, out of LDS)
    double2* YPs = reinterpret_cast<double2*>(cs + ((EW * ROWS + 3) & ~3));   // [R][PB]
    double2* GPs = YPs + R * PB;                                   // [R][PB]
    double* EGPs = reinterpret_cast<double*>(GPs + R * PB);        // [ROWS]

    const Ctl* c = d.ctl;
    const bool lead = bx == 0 && threadIdx.x == 0;
    const int k_tr = c->k;
    if (c->done) {                                                 // same as k_tcg_init on a finished solve
        if (lead) {
            frame_store(&d.F[0], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, 0, 0, 5, 0, 0, 1);
            d.ctl->tcg_running = 0;
            msdp_publish(d, k_tr, 0, 0);
        }
        return;
    }
    if (lead && !FUSE) msdp_publish(d, k_tr, 0, 1);                // "TR iteration k_tr has started" (host pipelining)
    // XR, two-level form (round 6, d.xr2_on: members that own a device each, up to 8): the reductions run over THIS member's workgroups
    // in its own block and over the members' sums (psync2, msdp_psync.h) -- `slots` is not used, bid / GS are local
    // (a template parameter: as a run-time flag the second form cost the flat <16, 5, 5> instance 16 scratch loads per trip, 10.6 -> 12.4 us)
    constexpr bool two = XR && XR2;
    __shared__ unsigned long long* shpeer[two ? 8 : 1];
    const int bid = (XR && !two) ? d.xr_gid0 + bx : bx;
    const int GS = (XR && !two) ? d.xr_gtot : d.G;                 // workgroups that synchronise
    const int xri = k_tr & 1;                                      // two-level: the region of this launch
    if (two) {
        if (threadIdx.x < 8) shpeer[threadIdx.x] = threadIdx.x < d.xr2_n ? d.xr2_peers[threadIdx.x] : d.xr2_blk;
        psync2_reset_other(d.xr2_blk, xri ^ 1, bid, GS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (XR) {
        unsigned long long* other = slots + (size_t)((k_tr & 1) ^ 1) * PSYNC_REGION;
        slots += (size_t)(k_tr & 1) * PSYNC_REGION;
        psync_reset_other(other, bid, GS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // performed before this workgroup's first post of this launch
    } else if (!FUSE) psync_reset_other(slots + PSYNC_REGION, bid, GS);     // region B belongs to the TR-iteration tail kernel
    // the grid reduction / barrier of this launch, in whichever form
#define XSYNC(nv, a, b, c, dr) (two ? psync2(d.xr2_blk, shpeer, d.xr2_n, d.xr2_me, xri, gen++, GS + d.xr2_skip, (nv), (a), (b), (c), sh, shb, err, bid, backoff, (dr)) \
                                    : psync(slots, gen++, GS, (nv), (a), (b), (c), sh, shb, err, bid, backoff, (dr)))
#define XBAR() (two ? xbar2() : pbarrier(slots, nbar++, GS, shb, err, bid))
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi, 0, bx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPR - 1), rsub = lane / (LPR * EP);
    const int epi = (lane / LPR) & (EP - 1);                       // EP > 1: which of the row's lane groups (0 owns the row)
    const bool colok_g = 2 * sub < d.ld;                           // the lane has columns (gathers)
    const bool colok = colok_g && epi == 0;                        // ... and owns them (registers, sums, stores)
    int cur = c->cur;
    const bool bench = c->bench_mode != 0;
    double Delta = c->Delta;
    const double kappa = c->kappa, theta = c->theta;
    const int mininner = c->mininner, maxinner = c->maxinner;
    double gg = c->gg;
    // trust-region level state (FUSE: the whole trustregions() loop runs in this launch; trustregions.m:441-767)
    double fx = c->fx;
    const double Delta_bar = c->Delta_bar, tolgradnorm = c->tolgradnorm, rho_prime = c->rho_prime, rho_reg_opt = c->rho_reg;
    const int maxiter = c->maxiter;
    int k_it = c->k;            // the statistics (counts, rho, ...) are kept in d.ctl by the lead thread, not in registers
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ gl = cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ eGl = cur ? d.eG[1] : d.eG[0];

    // row slot r of this lane group: slot = r*RSTEP + slot0, row = lo + slot (recomputed where needed: arrays
    // of R row indices / flags would sit in registers next to the resident rows)
    const int slot0 = wave * RPW + rsub;
#define SLOT(r) ((r) * RSTEP + slot0)
#define ROW(r) (lo + SLOT(r))
#define ROK(r) (ROW(r) < hi)
#define OK(r) (ROK(r) && colok)
    // p = 33..64 (LPR = 32).  R = 5 row slots (80 rows per workgroup: n <= 20480 on 256 CUs): every vector stays in
    // registers, 256 VGPRs with 7 dwords of scratch, two-synchronisation trip -- 9.9 us per trip on G81 at p = 40.
    // R = 8 (larger n): four resident vectors = 128 registers + temporaries spill; there mdelta and Hmdelta live
    // in LDS (the regions that hold Y and grad otherwise) and Y / grad are re-read from global memory (static
    // during the launch: plain cached loads, L2 resident); only eta and r stay in registers (LOWREG, three
    // synchronisations: 17.6 us per trip).
    constexpr bool LOWREG = (R > 5);
    // TWOSYNC (everything but LOWREG; measured with LOWREG: the extra C*mdelta registers spill, 21 us): two grid synchronisations per trip instead of three.  The workgroups publish
    // the rows of the new RESIDUAL (known before the second reduction) instead of the new direction (known only after
    // it), the stores complete before the workgroup posts its partial sums, so the second reduction doubles as the
    // barrier in front of the gathers; the product with the new direction follows from linearity,
    //   C*mdelta_new = C*tangent(r_new) + beta * C*mdelta_old      (mdelta_new = tangent(r_new + beta*mdelta_old), tCG.m:273,283;
    //                                                                mdelta_old is tangent to rounding, r_new is not quite)
    // with C*mdelta_old kept in registers (cmd).  Each workgroup's own rows of mdelta are still formed and
    // re-projected exactly as tCG.m:273,283 do.
    constexpr bool TWOSYNC = !LOWREG;
    // Round 2 published the rows of r_new themselves: the normal component r accumulates (rounding of :238, never projected)
    // stayed in cmd, and with it |Heta - Hess(eta)| / |Heta| reached 3e-9 after 100 trips on G81 where the direct products of the
    // chunked path stay at 5e-14 (tools/archive/tcg_invariant_probe.py).  Publishing tangent(r_new) removes the
    // source; what the re-projection of mdelta_old + beta-scaling still leaves (1e-16 per trip) is reset by a direct exchange
    // every `refresh`-th trip, which ends like a three-synchronisation trip: the workgroups
    // publish the rows of the NEW DIRECTION after beta is known, a value-less barrier follows, and the next product is gathered
    // directly (cmd starts afresh).  One extra barrier (1.5 us) per `refresh` trips; the schedule depends on the trip count only,
    // so every workgroup takes the same branch.
    const int refresh = c->persist_refresh;
    const int backoff = c->psync_backoff;
    // EARLY (round 5): the gather of tangent(r') leaves the critical path.  Its rows need only the neighbours' stores, not beta.
    // First form (kept in the history: per-wave row flags raised behind the drain of the stores, watched by the consumers):
    // 8.6-8.7 us per trip against 7.8 -- drain (0.5 us) + flag store becoming visible and polled (1.4 us median, 2.2 us on three of
    // the XCDs) + the gather (0.5 us) add up to MORE than the reduction they were to hide under, and the flag polls of 1700 waves
    // slowed the reduction itself (profiles/r5_persist_timeline_p32_flags.md).  This form has no flags and no drain: THE ROWS ARE
    // THEIR OWN FLAGS.  The exchange buffer has two halves that alternate trip by trip, and a half holds a NaN sentinel in every
    // 16-byte element until its owner stores the row: a workgroup (1) stores its rows of tangent(r') and posts its partial sums of
    // reduction 2 at once, (2) gathers the rows its rows of C reference and looks at what came back -- an element that still holds
    // the sentinel means "not stored yet": the gather is repeated (a 16-byte element is written by one lane of one store, whole),
    // (3) forms C*tangent(r') while reduction 2 is in flight, picks the reduction up (waves 0..2 poll it under the gather's wait),
    // then beta, the new direction and C*mdelta' = C*tangent(r') + beta*C*mdelta as before.  The half filled at trip j is put back
    // to the sentinel by its owner during trip j+1, behind reduction 1 of that trip (every gather of trip j was consumed before its
    // workgroup posted that reduction), and filled again at trip j+2.  Same row arithmetic, same summation orders, same decisions
    // as the trip above: the oracle / drift tests are unchanged.
    constexpr bool ALLG = !LOWREG && EW > 0 && R * EW <= 25;
    constexpr bool EARLY = EARLYP && TWOSYNC && ALLG && !XR;
    const int fbackoff = c->persist_early > 1 ? c->persist_early - 1 : 0;   // s_sleep units between the post of reduction 2 and the first gather
    // EARLY: eta lives in LDS (read by the trial step, updated by the commit, nothing else touches it): R x 4 registers less next to the
    // R x EW row registers of the early gather
    constexpr bool ELDS = EARLY;
    double2* Es = FUSE ? reinterpret_cast<double2*>(EGPs + ROWS) : YPs;   // [R][PB], behind everything else
    double2 eta[ELDS ? 1 : R], rr[R], md[LOWREG ? 1 : R], hmd[LOWREG ? 1 : R], cmd[TWOSYNC ? R : 1];
#define ETA_GET(r) (ELDS ? Es[(r) * PB + threadIdx.x] : eta[ELDS ? 0 : (r)])
#define ETA_SET(r, val) do { if (ELDS) Es[(r) * PB + threadIdx.x] = (val); else eta[ELDS ? 0 : (r)] = (val); } while (0)
#define VOFF(r) ((int64_t)ROW(r) * d.ld + 2 * sub)
#define Y_GET(r) (LOWREG ? (OK(r) ? ld2(Yl + VOFF(r)) : zz) : Ys[(r) * PB + threadIdx.x])
#define G_GET(r) (LOWREG ? (OK(r) ? ld2(gl + VOFF(r)) : zz) : Gs[(r) * PB + threadIdx.x])
#define MD_GET(r) (LOWREG ? Ys[(r) * PB + threadIdx.x] : md[LOWREG ? 0 : (r)])
#define MD_SET(r, val) do { if (LOWREG) Ys[(r) * PB + threadIdx.x] = (val); else md[LOWREG ? 0 : (r)] = (val); } while (0)
#define HMD_GET(r) (LOWREG ? Gs[(r) * PB + threadIdx.x] : hmd[LOWREG ? 0 : (r)])
#define HMD_SET(r, val) do { if (LOWREG) Gs[(r) * PB + threadIdx.x] = (val); else hmd[LOWREG ? 0 : (r)] = (val); } while (0)
    const double2 zz = make_double2(0.0, 0.0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // unconditional loads from a clamped row (then masked): branches around the loads serialise them
        const bool rok = ROK(r);
        const int rc = rok ? ROW(r) : lo;
        const int64_t o = (int64_t)rc * d.ld + (colok ? 2 * sub : 0);
        double2 y = ld2(Yl + o), g = ld2(gl + o);
        if (!OK(r)) { y = zz; g = zz; }
        if (!LOWREG) { Ys[r * PB + threadIdx.x] = y; Gs[r * PB + threadIdx.x] = g; }
        ETA_SET(r, zz); rr[r] = g; MD_SET(r, g); HMD_SET(r, zz);      // tCG.m:102-157
        const double egv = eGl[rc];
        int cw[EW > 0 ? EW : 1];
        double vw[EW > 0 ? EW : 1];
#pragma unroll
        for (int w = 0; w < EW; ++w) {
            cw[w] = XR ? d.xr_ellc[(int64_t)w * d.ell_stride + rc] : d.ellc[(int64_t)w * d.ell_stride + rc];
            vw[w] = d.ellv[(int64_t)w * d.ell_stride + rc];
        }
        if (XR && sub == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)                            // push targets of this row (slot addresses, 0: none)
                reinterpret_cast<unsigned long long*>(YPs)[t * ROWS + SLOT(r)] = rok ? d.xr_paddr[(int64_t)t * d.n_loc + rc] : 0ULL;
        }
        if (sub == 0) {
            eGs[SLOT(r)] = rok ? egv : 0.0;
#pragma unroll
            for (int w = 0; w < EW; ++w) {
                cs[w * ROWS + SLOT(r)] = cw[w];
                vs[w * ROWS + SLOT(r)] = rok ? vw[w] : 0.0;
            }
        }
    }
    if (threadIdx.x == 0) shb[7] = 0.0;
    __syncthreads();

    unsigned gen = 0, nbar = 0;
    auto xbar2 = [&]() { double z0 = 0.0, z1 = 0.0, z2 = 0.0; return psync2(d.xr2_blk, shpeer, d.xr2_n, d.xr2_me, xri, gen++, GS + d.xr2_skip, 0, z0, z1, z2, sh, shb, err, bid, backoff, false); };
    const unsigned xrow0 = 0u;                                     // (stores into the exchange buffer use local row numbers on every path)
    const unsigned xglob0 = 0u;                                    // (XR: the column indices are buffer-local, d.xr_ellc / d.xr_colind)
    const unsigned vec_bytes = (unsigned)((size_t)d.n_loc * d.ld * sizeof(double));
    // (EARLY: two halves of n_loc x ld doubles each)
    const unsigned half_bytes = (unsigned)((size_t)d.n_loc * d.ld * sizeof(double));
    // XR ("push" exchange, round 5): rs_md = this member's own exchange buffer -- its rows followed by a slot for every foreign row its
    // rows of C reference; ALL gathers are local loads with buffer-local indices (d.xr_ellc / d.xr_colind).  The owner of a row stores
    // it into its own buffer and into the halo slots of the members that reference it (d.xr_paddr: the slots' addresses).
    const unsigned xr_bytes = XR ? (unsigned)(((size_t)d.xr_cap + (size_t)d.xr_halo) * d.ld * sizeof(double)) : 0u;
    double* xr_own = d.xr_rows[0];                                 // (selects, not a dynamic index: that would put the kernel-argument array into scratch)
    if (XR) { if (d.xr_me == 1) xr_own = d.xr_rows[1]; if (d.xr_me == 2) xr_own = d.xr_rows[2]; if (d.xr_me == 3) xr_own = d.xr_rows[3]; }
    __amdgpu_buffer_rsrc_t rs_md = XR ? __builtin_amdgcn_make_buffer_rsrc(xr_own, 0, xr_bytes, 0x00020000)
                                      : __builtin_amdgcn_make_buffer_rsrc(d.mdx, 0, (EARLY ? 2u : 1u) * half_bytes, 0x00020000);
    // XR: this lane's 16 bytes of local row slot r also go to the members that reference the row: plain address arithmetic on a
    // descriptor held in LDS, no branch per member (the boundary waves are on the critical path of the reduction that follows)
    unsigned long long* pds = reinterpret_cast<unsigned long long*>(YPs);   // [2][ROWS] slot address (0: none)   (XR is never FUSE: the region is free)
    bool xr_has_push = false, xr_has_push2 = false;                // wave-uniform: some row of this wave is referenced by another member / by two (set below)
    auto xr_push = [&](int r, double2 v) {
        if (!xr_has_push) return;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !xr_has_push2) continue;
            const unsigned long long a = pds[t * ROWS + SLOT(r)];
            if (a != 0ULL && OK(r)) {
                double* ptr = reinterpret_cast<double*>(a) + 2 * sub;
                if (two) {                                         // the slot may live on another device: system scope
                    __hip_atomic_store(ptr, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(ptr + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __hip_atomic_store(ptr, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ptr + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    const double2 sent2 = make_double2(__longlong_as_double((long long)PSYNC_SENT), __longlong_as_double((long long)PSYNC_SENT));
    // my rows of half q back to the sentinel
    auto reset_half = [&](int q) {
#pragma unroll
        for (int r = 0; r < R; ++r) if (OK(r)) st2_sc1(rs_md, (unsigned)q * half_bytes + ((unsigned)ROW(r) * (unsigned)d.ld + 2 * sub) * 8u, sent2);
    };
    int xq = 0;                      // EARLY: the half the next publication goes to
    int pend = -1;                   // EARLY: the half to put back to the sentinel behind the next reduction 1 (-1: none)
    if (EARLY) { reset_half(0); reset_half(1); }                   // (whatever an earlier launch left; the first look at them is behind reduction 1 of trip 1)
    if (XR) {
        bool any = false, any2 = false;
#pragma unroll
        for (int r = 0; r < R; ++r) { any = any || pds[SLOT(r)] != 0ULL; any2 = any2 || pds[ROWS + SLOT(r)] != 0ULL; }
        xr_has_push2 = __builtin_amdgcn_ballot_w64(any2) != 0ULL;
        xr_has_push = xr_has_push2 || __builtin_amdgcn_ballot_w64(any) != 0ULL;
    }
    // byte offsets of the R x EW gathers of a trip (the same in the gradient buffer and in the exchange buffer: both have row stride ld)
    constexpr bool GOFF = ALLG && !EARLY && !XR && R * EW <= 15;        // (more row slots: the extra registers spill)
    const bool use_goff = GOFF && c->persist_goff != 0;
    unsigned goff[GOFF ? R : 1][GOFF ? EW : 1];
    if (GOFF) {
#pragma unroll
        for (int r = 0; r < (GOFF ? R : 0); ++r)
#pragma unroll
            for (int w = 0; w < (GOFF ? EW : 0); ++w)
                goff[GOFF ? r : 0][GOFF ? w : 0] = ((unsigned)cs[w * ROWS + SLOT(r)] * (unsigned)d.ld + (colok ? 2 * sub : 0)) * 8u;
    }
    bool failed = false;
    bool first_tr = true;
  for (;;) {   // ---- trust-region iterations (exactly one pass when !FUSE)
    if (FUSE && !first_tr) {
        // tCG.m:102-157 at the (possibly new) current point: eta = 0, r = mdelta = grad
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 g = Gs[r * PB + threadIdx.x];
            ETA_SET(r, zz); rr[r] = g; MD_SET(r, g); HMD_SET(r, zz);
        }
    }
    first_tr = false;
    double z_r = gg, d_Pd = gg, e_Pd = 0.0, e_Pe = 0.0, model_value = 0.0, alpha = 0.0, beta = 0.0;
    const double norm_r0 = sqrt(gg);
    int j = 0, stop = 5;
    // first direction = gradient: already in global memory (written by an earlier launch, or with sc1 stores by
    // the cost/gradient phase of the previous TR iteration); later trips gather the rows the other workgroups
    // stored with sc1 during this launch
    __amdgpu_buffer_rsrc_t rs_g = XR ? rs_md : __builtin_amdgcn_make_buffer_rsrc(cur ? d.Gr[1] : d.Gr[0], 0, vec_bytes, 0x00020000);
    if (XR) {
        // the first direction = the gradient, whose rows live in every rank's own buffer: hand them to the other ranks first
#pragma unroll
        for (int r = 0; r < R; ++r) { if (OK(r)) st2_sc1(rs_md, ((xrow0 + (unsigned)ROW(r)) * (unsigned)d.ld + 2 * sub) * 8u, MD_GET(r)); xr_push(r, MD_GET(r)); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!XBAR()) return;
    }
    bool first = true;
    bool have_early = false;         // EARLY: acc_e holds C*tangent(r') of my rows for the trip that starts
    int lastq = 0;                   // EARLY: the half the last direct exchange (refresh trip) went to
    double2 acc_e[EARLY ? R : 1];
    bool direct = false;             // TWOSYNC: the exchange buffer holds the rows of mdelta itself (a refresh trip preceded)
    // acc = sum_k C[row,k] * X[k, my columns] with X read through the agent-coherent resource rs
    constexpr int CB = 8;                                          // gathers of a CSR row in flight together (16 at eight lanes per row: measured, no gain)
    auto gather_row = [&](int r, __amdgpu_buffer_rsrc_t rs, unsigned base) -> double2 {
            double2 acc = zz;
            const unsigned gld = (unsigned)d.ld, gcol = colok ? 2 * sub : 0;
            if (EW > 0) {
                double2 x[EW > 0 ? EW : 1];
                double v[EW > 0 ? EW : 1];
#pragma unroll
                for (int w = 0; w < EW; ++w) {
                    const int cidx = cs[w * ROWS + SLOT(r)];
                    v[w] = vs[w * ROWS + SLOT(r)];
                    const unsigned off = base + ((unsigned)cidx * gld + gcol) * 8u;
                    x[w] = ld2_sc1(rs, off);
                }
#pragma unroll
                for (int w = 0; w < EW; ++w) {
                    acc.x = fma(v[w], x[w].x, acc.x);
                    acc.y = fma(v[w], x[w].y, acc.y);
                }
            } else if (EP > 1) {
                // entry-parallel: lane group epi takes the entries s0 + epi, s0 + epi + EP, ... -- all of them in flight together for rows of
                // up to CBE * EP entries -- and the groups' partial products are added (every group ends with the row's sum)
                constexpr int CBE = CB;                                // (sixteen in flight: 7.9 against 7.8 us per trip on G1 -- no gain)
                const bool rok = ROK(r);
                const int s0 = rok ? d.rowptr[ROW(r)] : 0, s1 = rok ? d.rowptr[ROW(r) + 1] : 0;
                const unsigned gcg = colok_g ? 2 * sub : 0;
                for (int k0 = s0 + epi; __builtin_amdgcn_ballot_w64(k0 < s1) != 0ULL; k0 += CBE * EP) {
                    double2 x[CBE];
                    double cvk[CBE];
#pragma unroll
                    for (int u = 0; u < CBE; ++u) {
                        const int k = k0 + u * EP;
                        const bool in = k < s1;
                        const int cidx = in ? d.colind[k] : (int)xglob0 + (rok ? ROW(r) : lo);
                        cvk[u] = in ? d.cval[k] : 0.0;
                        x[u] = ld2_sc1(rs, base + ((unsigned)cidx * gld + gcg) * 8u);
                    }
#pragma unroll
                    for (int u = 0; u < CBE; ++u) {
                        acc.x = fma(cvk[u], x[u].x, acc.x);
                        acc.y = fma(cvk[u], x[u].y, acc.y);
                    }
                }
#pragma unroll
                for (int m = LPR; m < LPR * EP; m <<= 1) {
                    acc.x += __shfl_xor(acc.x, m);
                    acc.y += __shfl_xor(acc.y, m);
                }
            } else if (ROK(r)) {
                // CSR rows of any length: (col, val) are static (plain loads, L2 resident), the direction rows are not
                const int s0 = d.rowptr[ROW(r)], s1 = d.rowptr[ROW(r) + 1];
                // batches of CB gathers in flight, software pipelined: the (col, val) pairs of batch b+1 are loaded
                // while the gathers of batch b are outstanding; the tail is clamped (weight 0), not peeled
                int cn[CB];
                double vn[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    const bool in = s0 + u < s1;
                    const int k = in ? s0 + u : (s1 > s0 ? s1 - 1 : 0);
                    cn[u] = (s1 > s0) ? (XR ? d.xr_colind[k] : d.colind[k]) : (XR ? ROW(r) : (int)xglob0 + ROW(r));
                    vn[u] = in ? d.cval[k] : 0.0;
                }
                for (int k0 = s0; k0 < s1; k0 += CB) {
                    double2 x[CB];
                    double cvk[CB];
#pragma unroll
                    for (int u = 0; u < CB; ++u) {
                        const unsigned off = base + ((unsigned)cn[u] * gld + gcol) * 8u;
                        cvk[u] = vn[u];
                        x[u] = ld2_sc1(rs, off);
                    }
                    if (k0 + CB < s1) {
#pragma unroll
                        for (int u = 0; u < CB; ++u) {
                            const bool in = k0 + CB + u < s1;
                            const int k = in ? k0 + CB + u : s1 - 1;
                            cn[u] = XR ? d.xr_colind[k] : d.colind[k];
                            vn[u] = in ? d.cval[k] : 0.0;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < CB; ++u) {
                        acc.x = fma(cvk[u], x[u].x, acc.x);
                        acc.y = fma(cvk[u], x[u].y, acc.y);
                    }
                }
            }
            if (!colok) acc = zz;
            return acc;
    };
    for (;;) {
        // ---- Hmdelta = proj(C*mdelta) - mdelta.*eG   (tCG.m:163, ManiSDP_onlyunitdiag.m:127-130)
        double pd = 0.0, u1 = 0.0, u2 = 0.0;
        TSTAMP(0);
        // all R * EW gathers of the trip are requested before the first one is consumed (the compiler left to itself requests the
        // gathers of one row slot, waits, does the row's arithmetic and only then turns to the next slot: R round trips)
        // (EARLY instances gather here on the first trip of a tCG and after a refresh trip only: row slot by row slot, so that the
        // R x EW row registers exist at ONE place of the loop -- the early gather below)
        constexpr bool ALLGT = ALLG && !EARLY;
        double2 X[ALLGT ? R : 1][ALLGT ? EW : 1];
        if (ALLGT) {
            const __amdgpu_buffer_rsrc_t rs = first ? rs_g : rs_md;
            const unsigned gld = (unsigned)d.ld, gcol = colok ? 2 * sub : 0;
#pragma unroll
            for (int r = 0; r < (ALLGT ? R : 0); ++r)
#pragma unroll
                for (int w = 0; w < (ALLGT ? EW : 0); ++w) {
                    // (round 5, option persist_goff: the byte offset of every gather kept in a register from kernel start instead of an
                    // LDS read + a quarter-rate 32-bit multiply + an add per gather and trip)
                    unsigned off;
                    if (GOFF && use_goff) off = goff[GOFF ? r : 0][GOFF ? w : 0];
                    else { const int cidx = cs[w * ROWS + SLOT(r)]; off = ((unsigned)cidx * gld + gcol) * 8u; }
                    X[ALLGT ? r : 0][ALLGT ? w : 0] = ld2_sc1(rs, off);
                }
        }
        auto hrow = [&](int r) {
            double2 acc;
            if (EARLY && have_early) acc = acc_e[EARLY ? r : 0];          // C*tangent(r') of my rows, gathered during reduction 2 of the previous trip
            else if (ALLGT) {
                acc = zz;
#pragma unroll
                for (int w = 0; w < (ALLGT ? EW : 0); ++w) {
                    const double v = vs[w * ROWS + SLOT(r)];
                    acc.x = fma(v, X[ALLGT ? r : 0][ALLGT ? w : 0].x, acc.x);
                    acc.y = fma(v, X[ALLGT ? r : 0][ALLGT ? w : 0].y, acc.y);
                }
                if (!colok) acc = zz;
            } else acc = gather_row(r, first ? rs_g : rs_md, (!first && EARLY) ? (unsigned)lastq * half_bytes : 0u);
            if (TWOSYNC) {
                // the gathered rows are those of r_new (first trip: of the gradient = mdelta; after a refresh: of mdelta)
                if (!first && !direct) { acc.x = fma(beta, cmd[TWOSYNC ? r : 0].x, acc.x); acc.y = fma(beta, cmd[TWOSYNC ? r : 0].y, acc.y); }
                cmd[TWOSYNC ? r : 0] = acc;
            }
            const double2 y = Y_GET(r), mdr = MD_GET(r);
            double dot = acc.x * y.x + acc.y * y.y;
            dot = msdp_group_sum<LPR>(dot);
            const double eg = eGs[SLOT(r)];
            double2 hq = make_double2(acc.x - y.x * dot - mdr.x * eg, acc.y - y.y * dot - mdr.y * eg);
            if (!OK(r)) hq = zz;
            HMD_SET(r, hq);
            pd += mdr.x * hq.x + mdr.y * hq.y;
        };
        if (LOWREG) {
            // rolled: in this mode the loop touches LDS / global memory only (no register arrays), and unrolling it
            // lets the compiler hoist the loads of all 8 rows (spills)
#pragma unroll 1
            for (int r0 = 0; r0 < R; r0 += 4) { hrow(r0); hrow(r0 + 1); hrow(r0 + 2); hrow(r0 + 3); }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) hrow(r);
        }
        // EARLY: the sentinel stores of the last trip (issued a reduction ago) are performed before anything of this trip is stored to the
        // same half -- explicit, and free at this point (a wait behind reduction 1 would sit on that reduction's own slot-reset store)
        if (EARLY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(1);
        if (!XSYNC(1, pd, u1, u2, false)) { failed = true; break; }
        TSTAMP(2);
        const double d_Hd = pd;                                                        // :166
        alpha = z_r / d_Hd;                                                            // :170
        const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;      // :173
        if (!bench && (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {                    // :183
            const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;   // :188
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double2 mdr = MD_GET(r), hq = HMD_GET(r);
                const double2 e0 = ETA_GET(r);
                ETA_SET(r, make_double2(e0.x - tau * mdr.x, e0.y - tau * mdr.y));      // :192
                rr[r].x -= tau * hq.x; rr[r].y -= tau * hq.y;                          // :198 (Heta = r - grad)
            }
            stop = (d_Hd <= 0.0) ? 1 : 2;
            ++j;
            break;
        }
        // ---- trial step and its three inner products (tCG.m:215-241)
        const bool refresh_now = TWOSYNC && refresh > 0 && ((j + 1) % refresh) == 0;   // this trip ends with a direct exchange
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        // EARLY: this trip's rows go to half xq; every wave's earlier stores to it (the sentinel, one trip ago) were waited for at the
        // end of the top phase of this trip
        const unsigned qoff = EARLY ? (unsigned)xq * half_bytes : 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 g = G_GET(r), mdr = MD_GET(r), hq = HMD_GET(r);
            const double2 e0 = ETA_GET(r);
            const double2 ne = make_double2(e0.x - alpha * mdr.x, e0.y - alpha * mdr.y);                 // :215
            const double2 nr = make_double2(rr[r].x - alpha * hq.x, rr[r].y - alpha * hq.y);             // :238
            const double2 nh = make_double2(nr.x - g.x, nr.y - g.y);                                     // new_Heta (:220)
            s1 += ne.x * g.x + ne.y * g.y;
            s2 += ne.x * nh.x + ne.y * nh.y;
            s3 += nr.x * nr.x + nr.y * nr.y;
            if (TWOSYNC && !refresh_now) {
                // what the neighbours gather is the PROJECTED residual row: C*tangent(r_new) = C*r_new - C*(Y.*rowdot(Y, r_new)).
                // r itself keeps the bits tCG.m gives it (:238 does not project); its rounding-level normal component is what
                // the re-projection of mdelta removes (:283) and what the assembled product would otherwise keep and amplify
                const double2 y = Y_GET(r);
                const double dn = msdp_group_sum<LPR>(nr.x * y.x + nr.y * y.y);
                const double2 tr = make_double2(nr.x - y.x * dn, nr.y - y.y * dn);
                if (OK(r)) st2_sc1(rs_md, qoff + ((xrow0 + (unsigned)ROW(r)) * (unsigned)d.ld + 2 * sub) * 8u, tr);
                if (XR) xr_push(r, tr);
            }
        }
        if (EARLY && !refresh_now) {
            // (1) reduction 2 goes out at once -- nothing waits for the row stores any more
            psync_post3(slots, gen, s1, s2, s3, sh, bid);
            TSTAMP(3);
            // the half of the PREVIOUS trip goes back to the sentinel now: reduction 1 of this trip has returned, so every workgroup has
            // consumed its gather of it; these stores are performed long before the half is filled again (next trip, behind a reduction)
            if (pend >= 0) reset_half(pend);
            pend = xq;
            for (int q = 0; q < fbackoff; ++q) __builtin_amdgcn_s_sleep(1);
            TSTAMP(4);
            // (2) + (3): gather the rows of tangent(r') my rows of C reference until no element holds the sentinel; waves 0..2 look at
            // their value array of reduction 2 under the same wait.  C*tangent(r') stays in acc_e until the top of the next trip.
            const unsigned long long* p0 = psync_poll_base(slots, gen, bid);
            bool r2ok = wave >= 3, fail = false;
            double r2t = 0.0;
            int spins = 0;
            {
                // One full gather; afterwards only the elements that still held the sentinel are asked for again, by the lanes that
                // miss them (a retry of everything would put the whole 5 x n x ld x 8 bytes on the fabric again -- the gather is
                // bandwidth-bound there: two full attempts cost 1.5 us, profiles/r5_persist_timeline_p32_sentinel_full_retry.md)
                double2 XE[EARLY ? R : 1][EARLY ? EW : 1];
#pragma unroll
                for (int r = 0; r < (EARLY ? R : 0); ++r)
#pragma unroll
                    for (int w = 0; w < (EARLY ? EW : 0); ++w) {
                        const int cidx = cs[w * ROWS + SLOT(r)];
                        XE[EARLY ? r : 0][EARLY ? w : 0] = ld2_sc1(rs_md, qoff + ((unsigned)cidx * (unsigned)d.ld + (colok ? 2 * sub : 0)) * 8u);
                    }
                if (!r2ok) r2ok = psync_poll_once(p0, GS, r2t);
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int r = 0; r < (EARLY ? R : 0); ++r)
#pragma unroll
                        for (int w = 0; w < (EARLY ? EW : 0); ++w) {
                            const double2 x = XE[EARLY ? r : 0][EARLY ? w : 0];
                            if ((unsigned long long)__double_as_longlong(x.x) == PSYNC_SENT || (unsigned long long)__double_as_longlong(x.y) == PSYNC_SENT) {
                                const int cidx = cs[w * ROWS + SLOT(r)];
                                XE[EARLY ? r : 0][EARLY ? w : 0] = ld2_sc1(rs_md, qoff + ((unsigned)cidx * (unsigned)d.ld + (colok ? 2 * sub : 0)) * 8u);
                                ready = false;
                            }
                        }
                    if (__builtin_amdgcn_ballot_w64(!ready) == 0ULL) break;
                    if (!r2ok) r2ok = psync_poll_once(p0, GS, r2t);
                    ++spins;
                    if (spins > PSYNC_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
                }
#pragma unroll
                for (int r = 0; r < (EARLY ? R : 0); ++r) {
                    double2 acc = zz;
#pragma unroll
                    for (int w = 0; w < (EARLY ? EW : 0); ++w) {
                        const double v = vs[w * ROWS + SLOT(r)];
                        acc.x = fma(v, XE[EARLY ? r : 0][EARLY ? w : 0].x, acc.x);
                        acc.y = fma(v, XE[EARLY ? r : 0][EARLY ? w : 0].y, acc.y);
                    }
                    if (!colok) acc = zz;
                    acc_e[EARLY ? r : 0] = acc;
                }
            }
            TSTAMP(5);
            // (4) reduction 2
            while (!r2ok && !fail) {
                r2ok = psync_poll_once(p0, GS, r2t);
                ++spins;
                if (spins > PSYNC_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) fail = true;
            }
            if (!psync_finish3(slots, gen++, r2t, fail, s1, s2, s3, shb, err, bid)) { failed = true; break; }
            have_early = true;
            xq ^= 1;
            TSTAMP(6);
        } else {
            TSTAMP(3);
            // (TWOSYNC: my residual rows are performed before I post -- the wait sits inside psync, behind the wave sums)
            TSTAMP(4);
            if (!XSYNC(3, s1, s2, s3, TWOSYNC)) { failed = true; break; }
            have_early = false;
            TSTAMP(5);
            // (EARLY instance, refresh trip: reduction 1 of this trip has returned -- the pending half can go back to the sentinel)
            if (EARLY && pend >= 0) { reset_half(pend); pend = -1; }
        }
        e_Pe = e_Pe_new;
        const double new_model = s1 + 0.5 * s2;                                        // :227
        const double r_r = s3;
        if (!bench && new_model >= model_value) { stop = 6; ++j; break; }              // :228 (eta, Heta stay)
        // (EARLY: the step length through an opaque copy, so that the compiler recomputes the commit instead of carrying the trial's
        // 2 x R row values across the gather of the neighbours' rows -- they were the spills of this instance)
        double alpha_c = alpha;
        if (EARLY) asm volatile("" : "+v"(alpha_c));
#pragma unroll
        for (int r = 0; r < R; ++r) {                                                  // :233-238 commit (same bits as the trial)
            const double2 mdr = MD_GET(r), hq = HMD_GET(r);
            const double2 e0 = ETA_GET(r);
            ETA_SET(r, make_double2(e0.x - alpha_c * mdr.x, e0.y - alpha_c * mdr.y));
            rr[r].x -= alpha_c * hq.x; rr[r].y -= alpha_c * hq.y;
        }
        model_value = new_model;
        ++j;
        const double norm_r = sqrt(r_r);
        const double nr0t = (theta == 1.0) ? norm_r0 : pow(norm_r0, theta);
        if (!bench && j >= mininner && norm_r <= norm_r0 * fmin(nr0t, kappa)) {       // :249
            stop = (kappa < nr0t) ? 3 : 4;
            break;
        }
        if (j >= maxinner) break;                                                      // :160 (stop stays 5)
        beta = r_r / z_r;                                                              // :272
        e_Pd = beta * (e_Pd + alpha * d_Pd);                                           // :286
        d_Pd = r_r + beta * beta * d_Pd;                                               // :287
        z_r = r_r;
        // ---- mdelta = tangent(r + beta*mdelta)  (:273,283) and hand the new rows to the neighbours
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 y = Y_GET(r), mdr = MD_GET(r);
            const double2 v = make_double2(rr[r].x + beta * mdr.x, rr[r].y + beta * mdr.y);
            double dot = v.x * y.x + v.y * y.y;
            dot = msdp_group_sum<LPR>(dot);
            const double2 mnew = make_double2(v.x - y.x * dot, v.y - y.y * dot);
            MD_SET(r, mnew);
            if ((!TWOSYNC || refresh_now) && OK(r)) st2_sc1(rs_md, qoff + ((xrow0 + (unsigned)ROW(r)) * (unsigned)d.ld + 2 * sub) * 8u, mnew);
            if (XR && (!TWOSYNC || refresh_now)) xr_push(r, mnew);
        }
        if (!TWOSYNC || refresh_now) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my rows are performed before my workgroup posts
            if (!XBAR()) { failed = true; break; }
            if (EARLY) { lastq = xq; pend = xq; xq ^= 1; }         // the direction rows sit in half lastq until the next trip has gathered them
        }
        direct = refresh_now;
        first = false;
        { --j; TSTAMP(EARLY && have_early ? 7 : 6); ++j; }          // (j was advanced above: stamp under the trip's own index)
    }
    if (failed) return;
    if (!FUSE) {
        // ---- hand eta, Heta = r - grad and the final scalars to the RTR kernels (trustregions.m:540-550)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (OK(r)) {
                const int64_t o = (int64_t)ROW(r) * d.ld + 2 * sub;
                const double2 g = G_GET(r);
                st2(d.eta[0] + o, ETA_GET(r));
                st2(d.Heta[0] + o, make_double2(rr[r].x - g.x, rr[r].y - g.y));
            }
        }
        if (lead) {
            frame_store(&d.F[0], z_r, d_Pd, e_Pd, e_Pe, model_value, norm_r0, alpha, beta, 0, j, stop, 0, 0, 0);
            d.ctl->tcg_running = 0;
            msdp_publish(d, k_tr, j, 0);
        }
        return;
    }
    // ================= rest of the TR iteration (FUSE): trustregions.m:540-729 =================
    // x_prop = retr(x, eta) (ManiSDP_onlyunitdiag.m:142-145), <eta, grad + .5*Heta> (trustregions.m:549-550)
    double prd = 0.0, pf = 0.0, pgg = 0.0;
    __amdgpu_buffer_rsrc_t rs_yp = __builtin_amdgcn_make_buffer_rsrc(cur ? d.Y[0] : d.Y[1], 0, vec_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_gp = __builtin_amdgcn_make_buffer_rsrc(cur ? d.Gr[0] : d.Gr[1], 0, vec_bytes, 0x00020000);
    double* eGp = cur ? d.eG[0] : d.eG[1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double2 y = Ys[r * PB + threadIdx.x], g = Gs[r * PB + threadIdx.x];
        const double2 he = make_double2(rr[r].x - g.x, rr[r].y - g.y);
        const double2 e0 = ETA_GET(r);
        prd += e0.x * (g.x + 0.5 * he.x) + e0.y * (g.y + 0.5 * he.y);
        const double2 x = make_double2(y.x + e0.x, y.y + e0.y);
        double nn = sqrt(msdp_group_sum<LPR>(x.x * x.x + x.y * x.y));
        if (!(nn > 0.0)) nn = 1.0;                                  // empty row slot
        const double2 ypr = OK(r) ? make_double2(x.x / nn, x.y / nn) : zz;
        YPs[r * PB + threadIdx.x] = ypr;
        if (OK(r)) st2_sc1(rs_yp, ((unsigned)ROW(r) * (unsigned)d.ld + 2 * sub) * 8u, ypr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!pbarrier(slots, nbar++, GS, shb, err, bid)) return;
    // (EARLY: every workgroup has left the tCG -- the half its last trip published can go back to the sentinel)
    if (EARLY && pend >= 0) { reset_half(pend); pend = -1; }
    // cost and gradient at the proposal (ManiSDP_onlyunitdiag.m:117-125): YC = Y*C, eG = sum(YC.*Y), G = YC - Y.*eG.
    // Rolled loop over the row slots (LDS in, LDS out): this phase runs once per TR iteration and must not add
    // register pressure to the tCG loop above.
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        const double2 acc = gather_row(r, rs_yp, 0u);
        const double2 ypr = YPs[r * PB + threadIdx.x];
        const double dot = msdp_group_sum<LPR>(acc.x * ypr.x + acc.y * ypr.y);
        const double2 gpr = OK(r) ? make_double2(acc.x - ypr.x * dot, acc.y - ypr.y * dot) : zz;
        GPs[r * PB + threadIdx.x] = gpr;
        pgg += gpr.x * gpr.x + gpr.y * gpr.y;
        if (sub == 0 && epi == 0) {                               // (EP > 1: the lane group that owns the row)
            EGPs[SLOT(r)] = ROK(r) ? dot : 0.0;
            if (ROK(r)) { pf += 0.5 * dot; eGp[ROW(r)] = dot; }
        }
        if (OK(r)) st2_sc1(rs_gp, ((unsigned)ROW(r) * (unsigned)d.ld + 2 * sub) * 8u, gpr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the proposal's gradient rows are in place before the post
    if (!psync(slots, gen++, GS, 3, pf, pgg, prd, sh, shb, err, bid, backoff)) return;
    {   // trustregions.m:548-729, identical in every workgroup (same bits in, same decision out)
        const double fp = pf, ggp = pgg;
        double rhonum = fx - fp;                                             // :548
        double rhoden = -prd;                                                // :550
        const double rreg = fmax(1.0, fabs(fx)) * 2.220446049250313e-16 * rho_reg_opt;   // :579
        rhonum += rreg;
        rhoden += rreg;
        const bool model_decreased = rhoden >= 0.0;                          // :614
        const double rho = rhonum / rhoden;                                  // :621
        if (rho < 0.25 || !model_decreased || isnan(rho)) Delta = Delta / 4.0;            // :653
        else if (rho > 0.75 && (stop == 1 || stop == 2)) Delta = fmin(2.0 * Delta, Delta_bar);   // :669
        const bool accept = model_decreased && rho > rho_prime;              // :688
        if (lead) {
            Ctl* cw = d.ctl;
            cw->rho = rho; cw->rhonum = rhonum; cw->rhoden = rhoden; cw->fx_prop = fp; cw->gg_prop = ggp;
            if (accept) cw->accepted++; else cw->rejected++;
            cw->hessvecs += j;
            cw->cost_evals++;
            cw->last_stop_inner = stop;
        }
        if (accept) {
            cur ^= 1;
            fx = fp; gg = ggp;
#pragma unroll 1
            for (int r = 0; r < R; ++r) {
                Ys[r * PB + threadIdx.x] = YPs[r * PB + threadIdx.x];
                Gs[r * PB + threadIdx.x] = GPs[r * PB + threadIdx.x];
                if (sub == 0) eGs[SLOT(r)] = EGPs[SLOT(r)];
            }
        }
        ++k_it;                                                              // :729
    }
    __syncthreads();                                                         // eGs / Ys / Gs updates visible to the whole workgroup
    if (sqrt(gg) < tolgradnorm || k_it >= maxiter) break;                    // stoppingcriterion.m:51-72
  }
    if (lead) {
        Ctl* cw = d.ctl;
        cw->fx = fx; cw->gg = gg; cw->norm_grad = sqrt(gg); cw->Delta = Delta;
        cw->k = k_it; cw->cur = cur;
        cw->done = 1;
        cw->tcg_running = 0;
        frame_store(&d.F[0], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, cw->last_stop_inner, 0, 0, 0);
    }
}

template <int LPR, int EW, int R, bool FUSE, bool TRACE = false, bool XR = false, bool EARLYP = false, bool XR2 = false, int EP = 1>
__global__ __launch_bounds__(PB) void k_tcg_persist_obl(Dev d, unsigned long long* slots, int* err) {
    tcg_persist_body<LPR, EW, R, FUSE, TRACE, XR, EARLYP, XR2, EP>(d, slots, err, (int)blockIdx.x);
}
// In-process ranks: ONE launch carries the workgroups of all members (member q owns the blocks [q*G, (q+1)*G)), so that their
// co-residency does not depend on how the runtime maps the members' streams onto hardware queues (two launches on one queue
// would wait for each other for ever).  Same body, same protocol as separate launches on separate devices would run.
struct XrDevs2 { Dev d[2]; };
struct XrDevs4 { Dev d[4]; };
template <int LPR, int EW, int R>
__global__ __launch_bounds__(PB) void k_tcg_persist_xr2(XrDevs2 ds, unsigned long long* slots, int* err) {
    const int G = ds.d[0].G;
    if ((int)blockIdx.x < G) tcg_persist_body<LPR, EW, R, false, false, true, false>(ds.d[0], slots, err, (int)blockIdx.x);
    else tcg_persist_body<LPR, EW, R, false, false, true, false>(ds.d[1], slots, err, (int)blockIdx.x - G);
}
template <int LPR, int EW, int R>
__global__ __launch_bounds__(PB) void k_tcg_persist_xr4(XrDevs4 ds, unsigned long long* slots, int* err) {
    const int G = ds.d[0].G, q = (int)blockIdx.x / G, bx = (int)blockIdx.x - q * G;
    switch (q) {
        case 0: tcg_persist_body<LPR, EW, R, false, false, true, false>(ds.d[0], slots, err, bx); break;
        case 1: tcg_persist_body<LPR, EW, R, false, false, true, false>(ds.d[1], slots, err, bx); break;
        case 2: tcg_persist_body<LPR, EW, R, false, false, true, false>(ds.d[2], slots, err, bx); break;
        default: tcg_persist_body<LPR, EW, R, false, false, true, false>(ds.d[3], slots, err, bx); break;
    }
}

#undef XSYNC
#undef XBAR
#undef VOFF
#undef ETA_GET
#undef ETA_SET
#undef Y_GET
#undef G_GET
#undef MD_GET
#undef MD_SET
#undef HMD_GET
#undef HMD_SET
#undef SLOT
#undef ROW
#undef ROK
#undef OK

#include "msdp_pipe.h"

// ------------------------------------------------------------------ host side
struct PersistPlan { int lpr, ew, r; size_t lds; int ep; };
static int plan_rstep(const PersistPlan& pl) { return PWAVES * (64 / (pl.lpr * (pl.ep > 1 ? pl.ep : 1))); }   // rows per pass of a workgroup

static bool persist_plan(const Dev& d, int G, PersistPlan& pl, bool allow_ep = true) {
    if (!d.rowptr) return false;
    int half = d.ld / 2, lpr = 1;
    while (lpr < half && lpr < 64) lpr <<= 1;
    if (half > 32) return false;                       // p <= 64: the resident rows must fit the register budget
    if (lpr < 8) lpr = 8;
    // (p <= 8 runs the eight-lane instances with half of their lanes masked.  A four-lane instance -- one row slot of 128 rows
    // per workgroup -- was measured in round 3 and is no faster: 7.29 against 7.20 us per trip on G81 at p = 8; its gathers
    // touch sixteen 64-byte row segments per wave instruction instead of eight 128-byte ones.)
    pl.lpr = lpr;
    pl.ew = (d.ellW < 1 || d.ellW > 8) ? 0 : (d.ellW <= 5 ? 5 : 8);      // 0: CSR rows of any length
    pl.r = lpr / 4;                                    // 128 row slots per workgroup
    pl.ep = 1;
    const int rstep = PWAVES * (64 / lpr);
    const int need = (d.n_loc + G - 1) / G;
    // CSR rows on a grid that leaves most lanes without a row (round 6): all 64 / lpr lane groups of a wave share ONE row and split its
    // entries (EP instances: 8 rows per pass, R = 2 / 3 / 5 passes at 8 / 16 / 32 lanes per row) -- G1 (800 rows of ~48 entries) 12.5 us
    // per trip on 16 workgroups -> one gather round trip on 104
    if (pl.ew == 0 && allow_ep && d.persist_ep != 0) {
        const int rr = lpr == 8 ? 2 : (lpr == 16 ? 3 : 5);
        if (need <= rr * PWAVES) {
            pl.r = rr; pl.ep = 64 / lpr;
            const size_t rows = (size_t)pl.r * PWAVES;
            pl.lds = (size_t)2 * pl.r * PB * sizeof(double2) + rows * sizeof(double);
            return true;
        }
    }
    // p = 33..64: five row slots (80 rows per workgroup: n <= 20480 on 256 CUs) keep every vector in registers and take
    // the two-synchronisation trip; eight slots (LOWREG: mdelta / Hmdelta in LDS, three synchronisations) beyond that
    if (lpr == 32 && need <= 5 * rstep) pl.r = 5;
    // p = 17..32: three row slots (96 rows per workgroup) when they suffice -- the fourth slot of the 128-slot form would be
    // a quarter of the trip's gathers and arithmetic spent on masked rows (G81 on 256 CUs owns 79 rows per workgroup)
    if (lpr == 16 && need <= 3 * rstep) pl.r = 3;
    // p = 17..32 beyond 128 rows per workgroup (n > 32768 on 256 CUs): eight row slots in the LOWREG form (mdelta / Hmdelta in
    // LDS, three synchronisations) instead of falling back to the chunked path
    if (lpr == 16 && need > 4 * rstep && need <= 8 * rstep) pl.r = 8;
    // ... but up to 160 rows per workgroup (n <= 40960 on 256 CUs) five slots still keep every vector in registers: the
    // two-synchronisation trip instead of LOWREG's three (round 4)
    if (lpr == 16 && need > 4 * rstep && need <= 5 * rstep) pl.r = 5;
    // p <= 16 (64 rows per slot): four slots (all in registers, two synchronisations) up to 256 rows per workgroup, i.e.
    // n <= 65536 on 256 CUs (eight slots would need 163 KB of LDS with the ELL rows)
    if (lpr == 8 && need > 2 * rstep && need <= 4 * rstep) pl.r = 4;
    // A/B (option persist_slots): force the number of row slots of the p = 17..32 plan (3 -> 96 rows, 4 -> 128 rows per workgroup)
    if (d.persist_slots > 0 && lpr == 16 && (d.persist_slots == 3 || d.persist_slots == 4) && need <= d.persist_slots * rstep) pl.r = d.persist_slots;
    if (need > pl.r * rstep) return false;
    const size_t rows = (size_t)pl.r * rstep;
    pl.lds = (size_t)2 * pl.r * PB * sizeof(double2) + rows * sizeof(double) + (size_t)pl.ew * rows * (sizeof(double) + sizeof(int));
    return true;
}

typedef void (*persist_fn)(Dev, unsigned long long*, int*);
// EARLY instances (round 5): rows of <= 5 entries, everything in registers
static persist_fn persist_kernel_early(const PersistPlan& pl, bool fuse) {
    if (pl.ew != 5) return nullptr;
    if (pl.lpr == 16 && pl.r == 3) return fuse ? k_tcg_persist_obl<16, 5, 3, true, false, false, true> : k_tcg_persist_obl<16, 5, 3, false, false, false, true>;
    if (pl.lpr == 8 && pl.r == 2) return fuse ? k_tcg_persist_obl<8, 5, 2, true, false, false, true> : k_tcg_persist_obl<8, 5, 2, false, false, false, true>;
    // (four and five row slots: the R x EW row registers of the early gather on top of five resident vectors spill 350-600 bytes per
    // lane -- those sizes keep the round-4 trip)
    return nullptr;
}
// One-reduction instances (round 5, msdp_pipe.h): rows of <= 5 entries, every vector in registers
static persist_fn persist_kernel_pipe(const PersistPlan& pl, bool fuse = false) {
    if (pl.ep > 1) {                                   // CSR rows, entry-parallel lanes (round 6)
        if (pl.ew != 0) return nullptr;
        if (pl.lpr == 8 && pl.r == 2) return fuse ? k_tcg_pipe_obl<8, 0, 2, false, true, 0, 8> : k_tcg_pipe_obl<8, 0, 2, false, false, 0, 8>;
        if (pl.lpr == 16 && pl.r == 3) return fuse ? nullptr : k_tcg_pipe_obl<16, 0, 3, false, false, 0, 4>;   // (fused: 148 bytes of scratch -- the two-reduction instance runs fused)
        return nullptr;
    }
    // rows of 6..8 entries (3-D grids: six neighbours + the diagonal): every row through the buffer, per-iteration launches
    if (pl.ew == 8 && pl.lpr == 8 && pl.r == 2) return fuse ? k_tcg_pipe_obl<8, 8, 2, false, true> : k_tcg_pipe_obl<8, 8, 2>;
    if (pl.ew == 8 && pl.lpr == 16 && pl.r == 3) return fuse ? nullptr : k_tcg_pipe_obl<16, 8, 3>;   // (fused: 290 bytes of scratch)
    if (pl.ew != 5) return nullptr;
    if (pl.lpr == 16 && pl.r == 3) return fuse ? k_tcg_pipe_obl<16, 5, 3, false, true> : k_tcg_pipe_obl<16, 5, 3>;
    if (pl.lpr == 8 && pl.r == 2) return fuse ? k_tcg_pipe_obl<8, 5, 2, false, true> : k_tcg_pipe_obl<8, 5, 2>;
    // (four row slots -- 97..128 rows per workgroup at p = 17..32, 129..256 at p <= 16: the sixth resident vector spills, measured
    // 9.6 us per trip against 8.2 for the two-reduction trip on a 180 x 180 grid at p = 32; tools/archive/pipe_sizes_probe.py)
    return nullptr;
}
// early: 0 none, 1 the EARLY trip, 2 the one-reduction trip
static persist_fn persist_kernel(const PersistPlan& pl, bool fuse = false, int early = 0) {
    if (pl.ep > 1 && early == 2) { persist_fn f = persist_kernel_pipe(pl, fuse); if (f) return f; }
    if (pl.ep > 1) {                                   // CSR rows, entry-parallel lanes (two-reduction trip)
        if (pl.lpr == 8 && pl.r == 2) return fuse ? k_tcg_persist_obl<8, 0, 2, true, false, false, false, false, 8> : k_tcg_persist_obl<8, 0, 2, false, false, false, false, false, 8>;
        if (pl.lpr == 16 && pl.r == 3) return fuse ? k_tcg_persist_obl<16, 0, 3, true, false, false, false, false, 4> : k_tcg_persist_obl<16, 0, 3, false, false, false, false, false, 4>;
        if (pl.lpr == 32 && pl.r == 5) return fuse ? nullptr : k_tcg_persist_obl<32, 0, 5, false, false, false, false, false, 2>;
        return nullptr;
    }
    if (early == 2) { persist_fn f = persist_kernel_pipe(pl, fuse); if (f) return f; }
    if (early == 1) { persist_fn f = persist_kernel_early(pl, fuse); if (f) return f; }
#define PK(L, E) if (pl.lpr == L && pl.ew == E && pl.r == L / 4) return k_tcg_persist_obl<L, E, L / 4, false>;
    if (pl.lpr == 16 && pl.r == 3) {
        if (fuse) {
            if (pl.ew == 5) return k_tcg_persist_obl<16, 5, 3, true>;
            if (pl.ew == 8) return k_tcg_persist_obl<16, 8, 3, true>;
            if (pl.ew == 0) return k_tcg_persist_obl<16, 0, 3, true>;
        } else {
            if (pl.ew == 5) return k_tcg_persist_obl<16, 5, 3, false>;
            if (pl.ew == 8) return k_tcg_persist_obl<16, 8, 3, false>;
            if (pl.ew == 0) return k_tcg_persist_obl<16, 0, 3, false>;
        }
    }
    if (!fuse && pl.lpr == 8 && pl.r == 4) {
        if (pl.ew == 5) return k_tcg_persist_obl<8, 5, 4, false>;
        if (pl.ew == 8) return k_tcg_persist_obl<8, 8, 4, false>;
        if (pl.ew == 0) return k_tcg_persist_obl<8, 0, 4, false>;
    }
    if (!fuse && pl.lpr == 16 && pl.r == 8) {
        if (pl.ew == 5) return k_tcg_persist_obl<16, 5, 8, false>;
        if (pl.ew == 8) return k_tcg_persist_obl<16, 8, 8, false>;
        if (pl.ew == 0) return k_tcg_persist_obl<16, 0, 8, false>;
    }
    if (!fuse && pl.lpr == 16 && pl.r == 5) {
        if (pl.ew == 5) return k_tcg_persist_obl<16, 5, 5, false>;
        if (pl.ew == 8) return k_tcg_persist_obl<16, 8, 5, false>;
        if (pl.ew == 0) return k_tcg_persist_obl<16, 0, 5, false>;
    }
    if (!fuse && pl.lpr == 32 && pl.r == 5) {
        if (pl.ew == 5) return k_tcg_persist_obl<32, 5, 5, false>;
        if (pl.ew == 8) return k_tcg_persist_obl<32, 8, 5, false>;
        if (pl.ew == 0) return k_tcg_persist_obl<32, 0, 5, false>;
    }
#define PKF(L, E) if (fuse && pl.lpr == L && pl.ew == E && pl.r == L / 4) return k_tcg_persist_obl<L, E, L / 4, true>;
    PKF(8, 5) PKF(8, 8) PKF(16, 5) PKF(16, 8) PKF(8, 0) PKF(16, 0)
    if (fuse) return nullptr;                          // the fused form needs Y and grad in LDS (p <= 32)
    PK(8, 5) PK(8, 8) PK(16, 5) PK(16, 8) PK(32, 5) PK(32, 8) PK(8, 0) PK(16, 0) PK(32, 0)
#undef PK
#undef PKF
    return nullptr;
}

static int persist_mode(msdp_handle h) { return h->tune.persist_early ? 1 : (h->tune.persist_pipe ? 2 : 0); }   // (persist_early is off by default: asking for it wins)
static bool persist_is_pipe(msdp_handle h, const PersistPlan& pl, bool fuse) { return !h->tune.persist_early && h->tune.persist_pipe && persist_kernel_pipe(pl, fuse) != nullptr; }
static bool persist_is_early(msdp_handle h, const PersistPlan& pl, bool fuse) { return !persist_is_pipe(h, pl, fuse) && h->tune.persist_early && persist_kernel_early(pl, fuse) != nullptr; }
static size_t early_lds(const PersistPlan& pl);
static size_t pipe_lds(const PersistPlan& pl) { const size_t rows = (size_t)pl.r * plan_rstep(pl); return (size_t)pl.ew * rows * sizeof(int) + (size_t)2 * pl.r * PB * sizeof(double2); }   // + ls, HQs

// The persistent kernel has its own grid: at most one workgroup per CU (co-residency), independent of the grid
// of the row-parallel kernels around it (those exchange data through global memory only).
static int persist_grid(const Dev& d) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = v;
        else cus = 0;
        (void)hipGetLastError();
    }
    // (round 5: NOT d.G -- the grid of the row-parallel kernels follows THEIR lanes per row: 4 lanes at p <= 8, i.e. 256 rows per
    // pass and 80 workgroups for G81, on which this kernel (8 lanes per row at least) needed its four-slot instance and left two
    // thirds of the CUs idle: p = 8 6.8 us per trip against 5.5 at p = 16, p = 4 not eligible at all (12.6 us on the chunked path);
    // tools/archive/p_sweep_probe.py)
    int g = 256;                                       // psync polls 4 x 64 slots
    if (g > cus) g = (cus / 8) * 8;
    // the fewest workgroups that need the same number of row slots: a workgroup with 93 rows in three slots of 32 takes as
    // long as one with 79, and the grid synchronisation has fewer slots to poll
    PersistPlan pl;
    if (g >= 8 && persist_plan(d, g, pl)) {
        // (CSR rows, round 5: the row slots of a workgroup are walked one after the other, each a chain of gather batches as long as
        // its longest row, and an empty slot is skipped -- ONE slot per workgroup while the CUs last: G1, 800 rows of ~49 entries,
        // 8 workgroups x 2 slots 20.7 us per trip, 16 x 1 12.8; tools/archive/g1_trip_probe.py)
        const int cap = (pl.ew == 0 ? 1 : pl.r) * plan_rstep(pl);
        int gmin = (((d.n_loc + cap - 1) / cap + 7) / 8) * 8;
        if (gmin < 8) gmin = 8;
        if (gmin < g) g = gmin;
    }
    return g;
}

// 1: the persistent kernel can run this handle's tCG (and all its workgroups are co-resident); 0: use the chunked path
int msdp_persist_eligible(msdp_handle h) {
    const Dev& d = h->d;
    if (!h->tune.persist || h->persist_failed || h->use_comm || h->nranks != 1 || d.costkind != COST_SPARSE || d.manifold != MANI_OBLIQUE) return 0;
    if (!h->psync_slots) return 0;
    const int G = persist_grid(d);
    if (G < 8) return 0;
    PersistPlan pl;
    if (!persist_plan(d, G, pl)) return 0;
    persist_fn fn = persist_kernel(pl, false, persist_mode(h));
    if (!fn) return 0;
    if (persist_is_early(h, pl, false)) pl.lds += early_lds(pl);
    if (persist_is_pipe(h, pl, false)) pl.lds += pipe_lds(pl);
    // the ELL copy must be stored with the width the kernel is instantiated for
    if (pl.ew > 0 && d.ellW != pl.ew) return 0;
    if (h->persist_sig_lpr == pl.lpr && h->persist_sig_ew == pl.ew && h->persist_sig_r == pl.r && h->persist_sig_G == G && h->persist_sig_fn == (const void*)fn)
        return h->persist_sig_ok;
    int ok = 0;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, PB, pl.lds) == hipSuccess)
        ok = (per_cu >= 1 && cus >= G) ? 1 : 0;        // one workgroup per CU: never rely on two sharing a CU
    (void)hipGetLastError();
    h->persist_sig_lpr = pl.lpr; h->persist_sig_ew = pl.ew; h->persist_sig_r = pl.r; h->persist_sig_G = G; h->persist_sig_fn = (const void*)fn;
    h->persist_sig_ok = ok;
    return ok;
}

int msdp_launch_tcg_persist(msdp_handle h, int reset_slots) {
    const int G = persist_grid(h->d);
    PersistPlan pl;
    if (!persist_plan(h->d, G, pl)) { msdp_set_error("persistent tCG: not eligible"); return MSDP_ESTATE; }
    persist_fn fn = persist_kernel(pl, false, persist_mode(h));
    if (!fn) { msdp_set_error("persistent tCG: no kernel instance"); return MSDP_ESTATE; }
    if (persist_is_early(h, pl, false)) pl.lds += early_lds(pl);
    if (persist_is_pipe(h, pl, false)) pl.lds += pipe_lds(pl);
    Dev dp = h->d;
    dp.G = G;
    if (reset_slots) {
        hipLaunchKernelGGL(k_psync_reset, dim3(8), dim3(256), 0, h->stream, h->psync_slots, h->psync_err);
        HIPCHK(hipGetLastError());
    }
    if (dp.trace) {
        if (!(pl.lpr == 16 && pl.ew == 5 && pl.r == 3)) { msdp_set_error("persistent trace: only the <16, 5, 3> instance (17 <= p <= 32, rows of <= 5 entries) is traced"); return MSDP_EUNSUPPORTED; }
        fn = h->tune.persist_early ? k_tcg_persist_obl<16, 5, 3, false, true, false, true> : h->tune.persist_pipe ? k_tcg_pipe_obl<16, 5, 3, true> : k_tcg_persist_obl<16, 5, 3, false, true>;
        HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    }
    hipLaunchKernelGGL(fn, dim3(G), dim3(PB), pl.lds, h->stream, dp, h->psync_slots, h->psync_err);
    HIPCHK(hipGetLastError());
    return 0;
}
// Which trip the persistent kernel of this handle runs: 0 two reductions per trip (round 4), 1 the EARLY form, 2 one reduction per trip
// (msdp_pipe.h); -1: the handle's tCG is not persistent
extern "C" int msdp_debug_persist_form(msdp_handle h, int32_t* form) {
    if (!h || !form) return MSDP_EINVAL;
    *form = -1;
    if (!msdp_persist_eligible(h)) return 0;
    PersistPlan pl;
    if (!persist_plan(h->d, persist_grid(h->d), pl)) return 0;
    *form = persist_is_pipe(h, pl, false) ? 2 : (persist_is_early(h, pl, false) ? 1 : 0);
    return 0;
}
int msdp_persist_trace_dims(msdp_handle h, int* G, int* nj, int* j0) { *G = persist_grid(h->d); *nj = MSDP_TRACE_NJ; *j0 = MSDP_TRACE_J0; return 0; }

int msdp_tr_tail_grid(msdp_handle h) { return persist_grid(h->d); }

// LDS of the fused form: + the proposal point, its gradient (R x PB double2 each) and eG
static size_t fused_lds(const PersistPlan& pl) {
    const size_t rows = (size_t)pl.r * plan_rstep(pl);
    return pl.lds + 16 + (size_t)2 * pl.r * PB * sizeof(double2) + rows * sizeof(double);
}
// EARLY instances keep eta in LDS, behind everything else (R x PB double2)
static size_t early_lds(const PersistPlan& pl) { return 16 + (size_t)pl.r * PB * sizeof(double2); }

// Whole trustregions() loop in one launch (FUSE = true): tCG + retraction + cost/gradient at the proposal + the
// accept/reject logic, iterated on the device until gradnorm < tol or maxiter.  Needs Y and grad in LDS (p <= 32).
// Default where it applies (option fused_rtr = 0 keeps two launches per TR iteration): 10.5 vs 11.1 ms per RTR call on
// G81 p = 32 once the per-iteration statistics moved from registers to d.ctl (the first version spilled 33 registers
// inside the tCG loop and was 5% slower).
int msdp_persist_fused_ok(msdp_handle h) {
    if (!h->tune.fused_rtr) return 0;
    if (!msdp_persist_eligible(h)) return 0;
    PersistPlan pl;
    const int G = persist_grid(h->d);
    if (!persist_plan(h->d, G, pl)) return 0;
    // where the one-reduction trip exists for per-iteration launches only, those beat the fused two-reduction launch (G81 p = 32:
    // 164 000 against 143 000 Hess-vec/s)
    if (persist_is_pipe(h, pl, false) && !persist_is_pipe(h, pl, true)) return 0;
    persist_fn fn = persist_kernel(pl, true, persist_mode(h));
    if (!fn) return 0;
    {   // (the one-reduction form: + ls and HQs)
        const bool pipe = persist_is_pipe(h, pl, true), early = persist_is_early(h, pl, true);
        const size_t rows = (size_t)pl.r * plan_rstep(pl);
        pl.lds = fused_lds(pl) + (early ? early_lds(pl) : 0) + (pipe ? (size_t)pl.ew * rows * sizeof(int) + (size_t)2 * pl.r * PB * sizeof(double2) : 0);   // (+ ls, + HQs: the proposal's buffers are their own since round 6)
    }
    if (h->fused_sig_lpr == pl.lpr && h->fused_sig_ew == pl.ew && h->fused_sig_G == G && h->fused_sig_fn == (const void*)fn) return h->fused_sig_ok;
    int ok = 0, per_cu = 0;
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, PB, pl.lds) == hipSuccess)
        ok = per_cu >= 1 ? 1 : 0;
    (void)hipGetLastError();
    h->fused_sig_lpr = pl.lpr; h->fused_sig_ew = pl.ew; h->fused_sig_G = G; h->fused_sig_fn = (const void*)fn; h->fused_sig_ok = ok;
    return ok;
}

int msdp_launch_rtr_fused(msdp_handle h) {
    const int G = persist_grid(h->d);
    PersistPlan pl;
    if (!persist_plan(h->d, G, pl)) { msdp_set_error("fused RTR: not eligible"); return MSDP_ESTATE; }
    persist_fn fn = persist_kernel(pl, true, persist_mode(h));
    if (!fn) { msdp_set_error("fused RTR: no kernel instance"); return MSDP_ESTATE; }
    if (h->tune.timing) fprintf(stderr, "[msdp_rtr] fused launch: <%d, %d, %d>, %s, G = %d\n", pl.lpr, pl.ew, pl.r,
                                persist_is_pipe(h, pl, true) ? "one reduction per trip" : "two reductions per trip", G);
    {   // (the one-reduction form: + ls and HQs)
        const bool pipe = persist_is_pipe(h, pl, true), early = persist_is_early(h, pl, true);
        const size_t rows = (size_t)pl.r * plan_rstep(pl);
        pl.lds = fused_lds(pl) + (early ? early_lds(pl) : 0) + (pipe ? (size_t)pl.ew * rows * sizeof(int) + (size_t)2 * pl.r * PB * sizeof(double2) : 0);   // (+ ls, + HQs: the proposal's buffers are their own since round 6)
    }
    Dev dp = h->d;
    dp.G = G;
    if (dp.trace) {
        // (msdp_debug_persist_trace with reps <= 0: the phases of the TR iterations around the tCGs, FSTAMP in msdp_pipe.h)
        if (!(pl.lpr == 16 && pl.ew == 5 && pl.r == 3 && persist_is_pipe(h, pl, true))) { msdp_set_error("fused trace: only the one-reduction <16, 5, 3> instance (17 <= p <= 32, rows of <= 5 entries) is traced"); return MSDP_EUNSUPPORTED; }
        fn = k_tcg_pipe_obl<16, 5, 3, true, true>;
        HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
    }
    hipLaunchKernelGGL(k_psync_reset, dim3(8), dim3(256), 0, h->stream, h->psync_slots, h->psync_err);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(fn, dim3(G), dim3(PB), pl.lds, h->stream, dp, h->psync_slots, h->psync_err);
    HIPCHK(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------ cross-rank persistent tCG (XR)
// N in-process ranks (msdp_comm_init_local) run the persistent tCG TOGETHER: N x G workgroups, G = (256 / N) rounded down to a
// multiple of 8, all co-resident (one per CU), in ONE launch that member 0 issues for everybody (k_tcg_persist_xr2 / _xr4: member q
// owns the blocks [q*G, (q+1)*G) and its own Dev) -- separate launches of one process share a handful of hardware queues and may
// end up behind each other.  The plan follows the LARGEST row count of a rank (cap), so every member has the same row slots and
// the same synchronisation scheme; the ELL width is the members' common one, or the CSR form.
typedef void (*xr2_fn)(XrDevs2, unsigned long long*, int*);
typedef void (*xr4_fn)(XrDevs4, unsigned long long*, int*);
static bool xr_instance(int lpr, int ew, int r) {
#define XK(L, E, RR) if (lpr == L && ew == E && r == RR) return true;
    XK(8, 5, 2) XK(8, 0, 2) XK(8, 5, 4) XK(8, 0, 4) XK(16, 5, 3) XK(16, 0, 3) XK(16, 5, 5) XK(16, 0, 5) XK(32, 5, 5) XK(32, 0, 5)
#undef XK
    return false;
}
static xr2_fn xr2_kernel(int lpr, int ew, int r) {
#define XK(L, E, RR) if (lpr == L && ew == E && r == RR) return k_tcg_persist_xr2<L, E, RR>;
    XK(8, 5, 2) XK(8, 0, 2) XK(8, 5, 4) XK(8, 0, 4) XK(16, 5, 3) XK(16, 0, 3) XK(16, 5, 5) XK(16, 0, 5) XK(32, 5, 5) XK(32, 0, 5)
#undef XK
    return nullptr;
}
static xr4_fn xr4_kernel(int lpr, int ew, int r) {
#define XK(L, E, RR) if (lpr == L && ew == E && r == RR) return k_tcg_persist_xr4<L, E, RR>;
    XK(8, 5, 2) XK(8, 0, 2) XK(8, 5, 4) XK(8, 0, 4) XK(16, 5, 3) XK(16, 0, 3) XK(16, 5, 5) XK(16, 0, 5) XK(32, 5, 5) XK(32, 0, 5)
#undef XK
    return nullptr;
}
// two-level reductions (msdp_psync.h psync2): process ranks on different devices, or more than four of them, or on request (option
// xr_twolevel); the in-process group (one launch for everybody) keeps the flat ones
static bool xr_two_level(msdp_handle h, int nranks) {
    return h->lgroup_is_ipc && (h->xr2_multi || nranks > 4 || h->tune.xr_twolevel);
}
static bool xr_plan(msdp_handle h, int nranks, PersistPlan& pl, int* G_out) {
    const Dev& d = h->d;
    if (!h->tune.persist || !h->tune.xpersist || h->persist_failed || d.costkind != COST_SPARSE || d.manifold != MANI_OBLIQUE || d.rowfree) return false;
    if (!h->xr_ok || !d.xr_colind || !d.xr_pq) return false;     // a row referenced by more than two other members: lock-step trips
    const bool two = xr_two_level(h, nranks);
    if (nranks < 2 || nranks > (two ? 8 : 4)) return false;
    Dev dc = d;
    dc.n_loc = (d.n + nranks - 1) / nranks;                // the plan of the rank with the most rows
    int G;
    if (two) {
        // every member synchronises over ITS OWN grid: a whole device's when it owns one (the one-rank grid for its rows), a share of
        // the 256 workgroups when several members sit on one device (the single-GPU tests: 8 processes x 32 workgroups)
        const int share = h->xr2_share > 0 ? h->xr2_share : 1;
        G = persist_grid(dc);
        const int gcap = (256 / share) & ~7;
        if (G > gcap) G = gcap;
    } else G = (256 / nranks) & ~7;
    if (G < 8) return false;
    if (!persist_plan(dc, G, pl, false)) return false;
    if (pl.r > 5) return false;                            // LOWREG instances are not built for XR
    if (pl.ew == 8) pl.ew = 0;                             // rows of 6..8 entries: the CSR form
    if (!xr_instance(pl.lpr, pl.ew, pl.r)) return false;
    *G_out = G;
    return true;
}
static size_t xr_lds(int lpr, int ew, int r) {
    const size_t rows = (size_t)r * PWAVES * (64 / lpr);
    return (size_t)2 * r * PB * sizeof(double2) + rows * sizeof(double) + (size_t)ew * rows * (sizeof(double) + sizeof(int)) + 16 + 4 * rows * sizeof(int);   // + the push descriptors
}
int msdp_xpersist_eligible(msdp_handle h, int nranks) {
    PersistPlan pl; int G = 0;
    if (!xr_plan(h, nranks, pl, &G)) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    // every workgroup of the members that share this device must be resident at once
    const int here = xr_two_level(h, nranks) ? (h->xr2_share > 0 ? h->xr2_share : 1) : nranks;
    return cus >= G * here ? 1 : 0;
}
// Bytes of the shared synchronisation block: two slot regions (they alternate with the TR iteration)
size_t msdp_xpersist_slot_bytes() { return 4 * PSYNC_REGION * sizeof(unsigned long long); }   // tCG: regions 0 / 1, cross-rank TR tail: 2 / 3
int msdp_xpersist_reset(hipStream_t stream, unsigned long long* slots, int* err) {
    hipLaunchKernelGGL(k_psync_reset, dim3(8), dim3(256), 0, stream, slots, err);
    hipLaunchKernelGGL(k_psync_reset, dim3(8), dim3(256), 0, stream, slots + 2 * PSYNC_REGION, err);
    HIPCHK(hipGetLastError());
    return 0;
}
// This member's Dev for the combined launch and its plan {lanes per row, ELL width (0: CSR), row slots}
// slot addresses of the push exchange: row i of this member -> &rows[q][(position there) * ld] for each member q that references it
struct XrRows8 { double* r[8]; };
__global__ void k_xr_paddr(const int* __restrict__ pq, const int* __restrict__ pidx, XrRows8 rows, int ld, int n_loc,
                           unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n_loc) return;
    const int q = pq[i];
    double* base = rows.r[0];
#pragma unroll
    for (int t = 1; t < 8; ++t) if (q == t) base = rows.r[t];
    out[i] = q < 0 ? 0ULL : (unsigned long long)(base + (int64_t)pidx[i] * ld);
}
int msdp_xpersist_member(msdp_handle h, int nranks, int rank, double* const* rows, int halo_rows, Dev* out, int* plan3) {
    PersistPlan pl; int G = 0;
    if (!xr_plan(h, nranks, pl, &G)) { msdp_set_error("cross-rank persistent tCG: not eligible"); return MSDP_ESTATE; }
    const bool two = xr_two_level(h, nranks);
    *out = h->d;
    out->G = G; out->xr_gid0 = two ? 0 : rank * G; out->xr_gtot = two ? G : nranks * G; out->status = nullptr; out->trace = nullptr;
    // (the kernels address this member's own buffer only -- the others' slots through the push addresses below)
    for (int q = 0; q < 4; ++q) out->xr_rows[q] = rows[rank];
    out->xr_cap = (h->d.n + nranks - 1) / nranks; out->xr_me = 0; out->xr_halo = halo_rows;
    out->xr2_on = two ? 1 : 0; out->xr2_n = nranks; out->xr2_me = rank; out->xr2_skip = 0;
    // the one-reduction trip across the members (round 6): process ranks, the two shapes it is instantiated for (the launcher also wants
    // the members' common ELL width to be 5)
    out->xr_pipe = (h->lgroup_is_ipc && h->tune.persist_pipe && !h->tune.persist_early && pl.ew == 5 &&
                    ((pl.lpr == 16 && pl.r == 3) || (pl.lpr == 8 && pl.r == 2))) ? 1 : 0;
    out->xr_sys = h->xr2_multi ? 1 : 0;
    out->xr2_blk = h->xr2_blk; out->xr2_peers = h->xr2_peers;
    if (two && (!h->xr2_blk || !h->xr2_peers)) { msdp_set_error("cross-rank persistent tCG: the two-level blocks are missing"); return MSDP_ESTATE; }
    {   // the push addresses follow the members' buffers, the leading dimension and the partition
        bool same = h->xr_paddr && h->xr_paddr_ld == h->d.ld && h->xr_paddr_n == h->d.n_loc && h->xr_paddr_pq == h->d.xr_pq;
        for (int q = 0; q < 8; ++q) same = same && h->xr_paddr_key[q] == rows[q < nranks ? q : rank];
        if (!same) {
            const size_t need = (size_t)2 * (size_t)h->d.n_loc;
            if (h->xr_paddr_cap < need) {
                if (h->xr_paddr) (void)hipFree(h->xr_paddr);
                h->xr_paddr = nullptr; h->xr_paddr_cap = 0;
                if (hipMalloc(&h->xr_paddr, need * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); msdp_set_error("cross-rank persistent tCG: out of device memory"); return MSDP_ENOMEM; }
                h->xr_paddr_cap = need;
            }
            XrRows8 r8;
            for (int q = 0; q < 8; ++q) r8.r[q] = rows[q < nranks ? q : rank];
            if (need) hipLaunchKernelGGL(k_xr_paddr, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, h->stream, h->d.xr_pq, h->d.xr_pidx,
                                         r8, h->d.ld, h->d.n_loc, h->xr_paddr);
            if (hipGetLastError() != hipSuccess) { msdp_set_error("cross-rank persistent tCG: launch failed"); return MSDP_EHIP; }
            for (int q = 0; q < 8; ++q) h->xr_paddr_key[q] = r8.r[q];
            h->xr_paddr_ld = h->d.ld; h->xr_paddr_n = h->d.n_loc; h->xr_paddr_pq = h->d.xr_pq;
        }
        out->xr_paddr = h->xr_paddr;
    }
    plan3[0] = pl.lpr; plan3[1] = pl.ew; plan3[2] = pl.r;
    return 0;
}
// this member's two-level block back to its start state (every slot and member line the sentinel, the error word clear)
__global__ void k_xr2_reset(unsigned long long* blk) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < XR2_ERR_OFF; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i % XR2_REGION;
        blk[i] = (o >= PSYNC_CNT_OFF && o < PSYNC_REGION) ? 0ULL : PSYNC_SENT;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) blk[XR2_ERR_OFF + threadIdx.x] = 0ULL;
}
size_t msdp_xr2_block_bytes() { return XR2_BLOCK_U64 * sizeof(unsigned long long); }
size_t msdp_xr2_err_offset() { return XR2_ERR_OFF * sizeof(unsigned long long); }
int msdp_xr2_reset(hipStream_t stream, unsigned long long* blk) {
    hipLaunchKernelGGL(k_xr2_reset, dim3(64), dim3(256), 0, stream, blk);
    HIPCHK(hipGetLastError());
    return 0;
}
// Members in different processes (msdp_comm_init_ipc): every member launches its own G workgroups -- the same body, the same protocol;
// the launches of different processes run side by side (tools/ipc_probe.hip: a barrier across two such launches costs 1.75 us).
int msdp_launch_tcg_xpersist_one(hipStream_t stream, const Dev& dv, const int* plan, unsigned long long* slots, int* err) {
    const int lpr = plan[0], ew = plan[1], r = plan[2];
    persist_fn fn = nullptr;
#define XK(L, E, RR) if (lpr == L && ew == E && r == RR) fn = dv.xr2_on ? k_tcg_persist_obl<L, E, RR, false, false, true, false, true> : k_tcg_persist_obl<L, E, RR, false, false, true>;
    XK(8, 5, 2) XK(8, 0, 2) XK(8, 5, 4) XK(8, 0, 4) XK(16, 5, 3) XK(16, 0, 3) XK(16, 5, 5) XK(16, 0, 5) XK(32, 5, 5) XK(32, 0, 5)
#undef XK
    if (!fn) { msdp_set_error("cross-rank persistent tCG: no kernel instance"); return MSDP_ESTATE; }
    size_t lds = xr_lds(lpr, ew, r);
    if (dv.xr_pipe && ew == 5) {
        // ONE grid reduction per trip across the members (msdp_pipe.h, XRM = 1 flat / 2 two-level)
        persist_fn fp = nullptr;
        if (lpr == 16 && r == 3) fp = dv.xr2_on ? k_tcg_pipe_obl<16, 5, 3, false, false, 2> : k_tcg_pipe_obl<16, 5, 3, false, false, 1>;
        if (lpr == 8 && r == 2) fp = dv.xr2_on ? k_tcg_pipe_obl<8, 5, 2, false, false, 2> : k_tcg_pipe_obl<8, 5, 2, false, false, 1>;
        if (fp) {
            const size_t rows = (size_t)r * PWAVES * (64 / lpr);
            fn = fp;
            lds += (size_t)ew * rows * sizeof(int) + (size_t)2 * r * PB * sizeof(double2);      // + ls, HQs
        }
    }
    HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3(dv.G), dim3(PB), lds, stream, dv, slots, err);
    HIPCHK(hipGetLastError());
    return 0;
}

// Member 0: one launch for all members.  devs[q] / plans[3q..] as filled by msdp_xpersist_member on every member.
int msdp_launch_tcg_xpersist_all(hipStream_t stream, int nranks, const Dev* devs, const int* plans, unsigned long long* slots, int* err) {
    const int lpr = plans[0], r = plans[2];
    int ew = plans[1];
    for (int q = 1; q < nranks; ++q) {
        if (plans[3 * q] != lpr || plans[3 * q + 2] != r) { msdp_set_error("cross-rank persistent tCG: the members' plans differ"); return MSDP_ESTATE; }
        if (plans[3 * q + 1] != ew) ew = 0;                // different ELL widths: everybody walks its CSR rows
    }
    const int G = devs[0].G;
    const size_t lds = xr_lds(lpr, ew, r);
    if (nranks == 2) {
        xr2_fn fn = xr2_kernel(lpr, ew, r);
        if (!fn) { msdp_set_error("cross-rank persistent tCG: no kernel instance"); return MSDP_ESTATE; }
        XrDevs2 ds; ds.d[0] = devs[0]; ds.d[1] = devs[1];
        HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, dim3(2 * G), dim3(PB), lds, stream, ds, slots, err);
    } else {
        xr4_fn fn = xr4_kernel(lpr, ew, r);
        if (!fn) { msdp_set_error("cross-rank persistent tCG: no kernel instance"); return MSDP_ESTATE; }
        XrDevs4 ds;
        for (int q = 0; q < 4; ++q) ds.d[q] = devs[q < nranks ? q : nranks - 1];
        HIPCHK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, dim3(nranks * G), dim3(PB), lds, stream, ds, slots, err);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
