// msdp_pipe.h -- ONE grid synchronisation per tCG trip (round 5, option persist_pipe): the persistent kernel of msdp_persist.hip with
// the two reductions of a trip (tCG.m:166 <delta, H delta>; :227-241 model value and <r', r'>) folded into one.
//
// Why: a trip of k_tcg_persist_obl is latency -- two grid reductions of 1.9 us each, the drain of the published rows (0.7) and the
// gather (1.2) in front of 0.7 us of arithmetic (profiles/r4_persist_timeline_p32.md); taking the gather off the critical path was
// tried three ways in round 5 and lost to the visibility latency of the exchanged rows.  What can go is one of the reductions:
// everything the second one carries is a polynomial in the step length alpha whose coefficients are inner products of vectors known
// BEFORE alpha (tCG.m:215-241 with eta' = eta - alpha md, r' = r - alpha Hmd):
//     <r', r'>            = <r, r> - 2 alpha <r, Hmd> + alpha^2 <Hmd, Hmd>
//     <eta', g>           = <eta, g> - alpha <md, g>
//     <eta', r' - g>      = <eta, r - g> - alpha (<eta, Hmd> + <md, r - g>) + alpha^2 <md, Hmd>
// so ONE reduction of eight values -- <md, Hmd>, <r, Hmd>, <Hmd, Hmd>, <md, g>, <eta, Hmd>, <md, r - g>, and <r, r> and the model value
// of the point committed one trip ago, summed DIRECTLY (they replace the values the formulas gave: nothing is carried by recurrence
// for more than one trip) -- decides the whole trip: alpha, the boundary / negative-curvature test (:183), the model test (:228),
// the stopping test (:249), beta (:272).  The rows the neighbours need must then be stored BEFORE that reduction (it is the only
// barrier left), i.e. before alpha and beta are known: the workgroups publish the rows of Hmd, and the products follow from linearity
//     C tangent(r') = C tangent(r) - alpha C Hmd          (Hmd is tangent: it is a projection minus a multiple of md)
//     C md'         = C tangent(r') + beta C md           (md' = tangent(r' + beta md), tCG.m:273,283)
// with C tangent(r) (ctr) and C md (cmd) kept in registers -- one more resident vector than the two-reduction trip.  The published
// rows alternate between two halves of the exchange buffer (a workgroup that has passed the reduction of trip j may store its rows
// of trip j+1 while a neighbour still gathers those of trip j).  ctr and cmd are recurrences where the two-reduction trip gathers
// C tangent(r') directly, and what they lose grows with the SQUARE of the trips since they were last formed from gathers (the error of
// ctr grows by a rounding per trip and feeds cmd every trip): |Heta - Hess(eta)| / |Heta| on G81 after 50 trips 4.2e-11 with a
// refresh every 32 trips, 1.0e-11 every 16, 2.2e-12 every 8 (1.4e-13 for the two-reduction trip; tools/pipe_drift_probe.py).  The
// refresh needs no barrier here: every `pipe_refresh`-th trip publishes the rows of tangent(r) and md of ITS START next to those of
// Hmd (regions 3 and 2 of the buffer, in front of the same reduction) and the next trip gathers all three -- C tangent(r) and C md
// direct, then the step of the recurrences as on every trip.  (pipe_refresh >= 2, enforced by msdp_set_option / fill_ctl: with 1 every trip
// would both gather regions 2 / 3 at its top and store into them before its reduction -- no reduction between one workgroup's store and a
// slower one's gather; with >= 2 the reduction of the trip in between separates them.)
// Round 6: CSR rows (EW = 0) run this trip too, in the entry-parallel form of msdp_persist.hip (EP lane groups per row: see tcg_pipe_body).
// Per-row arithmetic of eta, r, md and Hmd: the statements of the two-reduction kernel (same reference lines).  What differs from
// tCG.m in floating point: C md is assembled (as in the two-reduction kernel), and <r', r'> / the model value that decide the
// stopping and model tests of a trip are the expanded forms above (relative error eps <r, r> / <r', r'>); the values that enter
// alpha and the next test are the directly summed ones.  Parity: tests/test_one_reduction_algebra.py (the trip restated in NumPy
// against the oracle's tCG: same trip counts and stop codes over radii, inner caps and refresh intervals -- no GPU), and on the GPU
// tests/test_gpu_onlyunitdiag.py (test_persistent_tcg_matches_oracle over both trip forms, fused and per-iteration launches;
// test_one_reduction_trip_solves_G81_like_the_two_reduction_trip; test_persistent_tcg_keeps_heta_equal_to_hess_eta_on_G81).
#pragma once
#include <type_traits>
#include "msdp_psync.h"

// Eight-value grid reduction: wave w polls value array w (PSYNC_NV = 8 arrays per generation); same slot protocol and layout as psync().
// (The layout with the eight values of a workgroup in ONE 64-byte line per replica is psync8_lines below.)
// The eight per-lane partials are reduced over the wave TOGETHER: a butterfly that halves the number of values a lane carries at each
// of its first three steps (10 exchanges and additions instead of the 48 of eight separate wave sums; lane 8 i ends with value i).
// sh8: 8 x PWAVES doubles, shb8: 16 doubles.  Returns false when a bounded spin ran out.
// on_ready(): called by every wave as soon as ITS poll has returned -- all workgroups have posted, so everything they stored in front
// of their posts is visible: the caller issues the next trip's gather there, under the rest of the reduction (wave sum, workgroup
// barrier, results to registers) and the trip's arithmetic.  The barrier behind it is a bare s_barrier (LDS traffic waited for
// explicitly): __syncthreads() carries a fence that would wait for those loads.
template <class F>
__device__ __forceinline__ bool psync8(unsigned long long* slots, unsigned gen, int G, double (&v)[8], double* sh8, double* shb8, int* err,
                                       int bid, int backoff, unsigned long long* tr, F on_ready) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        // v_permlane32_swap / v_permlane16_swap (gfx950) hand the lower half's copy of value 4 + k to the upper half and the upper
        // half's copy of value k to the lower one in ONE VALU instruction per dword -- no selects, no LDS crossbar (the first version,
        // six __shfl_xor stages, took 0.47 us of the trip)
        const bool h8 = (lane & 8) != 0;
        double a[4], b[2], x;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = msdp_swap_add<32>(v[k], v[4 + k]);
#pragma unroll
        for (int k = 0; k < 2; ++k) b[k] = msdp_swap_add<16>(a[k], a[2 + k]);
        x = (h8 ? b[1] : b[0]) + msdp_dpp<0x128>(h8 ? b[0] : b[1]);          // row_ror:8 = lane ^ 8 inside a row of 16
        x += msdp_dpp<MSDP_DPP_XOR1>(x); x += msdp_dpp<MSDP_DPP_XOR2>(x); x += msdp_dpp<MSDP_DPP_HALF_MIRROR>(x);
        if ((lane & 7) == 0) sh8[(lane >> 3) * PWAVES + w] = x;     // value lane / 8 of this wave
    }
    if (tr && threadIdx.x == 0) tr[4] = __builtin_readcyclecounter();
    // the caller's row stores (issued in front of the butterfly, draining under it) are performed before the workgroup posts
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[2] = __builtin_readcyclecounter();
    {
        unsigned long long* gbase = slots + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;
        if (w == 0) {
            const int rep = lane / PSYNC_NV, vi = lane % PSYNC_NV;
            double s = 0.0;
            for (int i = 0; i < PWAVES; ++i) s += sh8[vi * PWAVES + i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the reset store of this slot's other generations has been performed
            __hip_atomic_store(gbase + ((size_t)rep * PSYNC_NV + vi) * MSDP_MAX_GRID + bid,
                               (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long* p0 = gbase + ((size_t)(bid & (PSYNC_REP - 1)) * PSYNC_NV + w) * MSDP_MAX_GRID + lane;
        double r0;
        int spins = 0;
        bool fail = false;
        const int first = ((backoff >> 16) & 0xff) ? ((backoff >> 16) & 0xff) : 27;     // (this reduction's own figure: tools/pipe_probe.py)
        for (int q = 0; q < first; ++q) __builtin_amdgcn_s_sleep(1);
        if (tr && threadIdx.x == 0) tr[3] = __builtin_readcyclecounter();
        for (;;) {
            unsigned long long b0[4];
            asm volatile(
                "global_load_dwordx2 %0, %4, off sc1\n\t"
                "global_load_dwordx2 %1, %4, off offset:512 sc1\n\t"
                "global_load_dwordx2 %2, %4, off offset:1024 sc1\n\t"
                "global_load_dwordx2 %3, %4, off offset:1536 sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(b0[0]), "=&v"(b0[1]), "=&v"(b0[2]), "=&v"(b0[3])
                : "v"(p0)
                : "memory");
            bool ok = true;
            double t0 = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (lane + 64 * q < G) {
                    ok = ok && b0[q] != PSYNC_SENT;
                    t0 += __longlong_as_double((long long)b0[q]);
                }
            }
            r0 = t0;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT ||
                ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
            for (int q = 0; q < ((backoff >> 8) & 0xff); ++q) __builtin_amdgcn_s_sleep(1);
        }
        if (tr && threadIdx.x == 0) tr[7] = (__builtin_readcyclecounter() << 4) + (unsigned long long)(spins < 15 ? spins : 15);
        if (!fail) on_ready();
        r0 = msdp_wave_sum(r0);
        if (lane == 0) {
            shb8[w] = r0; shb8[8 + w] = fail ? 1.0 : 0.0;
            if (fail) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every workgroup has posted value array 0 of this generation, i.e. has finished reading the previous one: my slots of it back
        // to the sentinel
        if (w == 0)
            __hip_atomic_store(slots + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID +
                                   (size_t)lane * MSDP_MAX_GRID + bid,
                               PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    double bad = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = msdp_readlane(shb8[i], 0); bad += msdp_readlane(shb8[8 + i], 0); }
    return bad == 0.0;
}

// The same reduction with the eight values of a workgroup in ONE 64-byte line per replica (option pipe_lines):
//     slot(gen, rep, b, vi) = (gen * PSYNC_REP + rep) * PSYNC_NV * MSDP_MAX_GRID + b * 8 + vi,            b < 256
// a post is 8 lines instead of 64, and wave w polls the lines of the workgroups 32 w .. 32 w + 31 with two 16-byte loads per lane
// (lane l: workgroup 32 w + 16 q + l / 4, values 2 (l % 4) and 2 (l % 4) + 1).  Sums: over q, over the sixteen lanes of a value
// pair (DPP rotations inside a row, permlane swaps across rows), over the waves in index order -- the same order in every workgroup.
// shp: PWAVES x 8 doubles.
typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
template <class F>
__device__ __forceinline__ bool psync8_lines(unsigned long long* slots, unsigned gen, int G, double (&v)[8], double* sh8, double* shp, double* shb8,
                                             int* err, int bid, int backoff, unsigned long long* tr, F on_ready) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const bool h8 = (lane & 8) != 0;
        double a[4], b[2], x;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = msdp_swap_add<32>(v[k], v[4 + k]);
#pragma unroll
        for (int k = 0; k < 2; ++k) b[k] = msdp_swap_add<16>(a[k], a[2 + k]);
        x = (h8 ? b[1] : b[0]) + msdp_dpp<0x128>(h8 ? b[0] : b[1]);
        x += msdp_dpp<MSDP_DPP_XOR1>(x); x += msdp_dpp<MSDP_DPP_XOR2>(x); x += msdp_dpp<MSDP_DPP_HALF_MIRROR>(x);
        if ((lane & 7) == 0) sh8[(lane >> 3) * PWAVES + w] = x;
    }
    if (tr && threadIdx.x == 0) tr[4] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tr && threadIdx.x == 0) tr[2] = __builtin_readcyclecounter();
    unsigned long long* gbase = slots + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;
    if (w == 0) {
        const int rep = lane >> 3, vi = lane & 7;
        double s = 0.0;
        for (int i = 0; i < PWAVES; ++i) s += sh8[vi * PWAVES + i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(gbase + (size_t)rep * PSYNC_NV * MSDP_MAX_GRID + (size_t)bid * 8 + vi,
                           (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long* p0 = gbase + (size_t)(bid & (PSYNC_REP - 1)) * PSYNC_NV * MSDP_MAX_GRID + (size_t)(32 * w + (lane >> 2)) * 8 + 2 * (lane & 3);
    const bool in0 = 32 * w + (lane >> 2) < G, in1 = 32 * w + 16 + (lane >> 2) < G;   // lines >= G hold the sentinel for ever
    double t0 = 0.0, t1 = 0.0;
    int spins = 0;
    bool fail = false;
    const int first = ((backoff >> 16) & 0xff) ? ((backoff >> 16) & 0xff) : (backoff & 0xff);   // (19 units by default: tools/archive/pipe_ab_probe.py)
    for (int q = 0; q < first; ++q) __builtin_amdgcn_s_sleep(1);
    if (tr && threadIdx.x == 0) tr[3] = __builtin_readcyclecounter();
    if (__builtin_amdgcn_ballot_w64(in0) != 0ULL) {               // (a wave whose 32 workgroups do not exist has nothing to wait for)
        for (;;) {
            v2ul a0, a1;
            asm volatile(
                "global_load_dwordx4 %0, %2, off sc1\n\t"
                "global_load_dwordx4 %1, %2, off offset:1024 sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(a0), "=&v"(a1)
                : "v"(p0)
                : "memory");
            bool ok = true;
            t0 = 0.0; t1 = 0.0;
            if (in0) { ok = a0.x != PSYNC_SENT && a0.y != PSYNC_SENT; t0 = __longlong_as_double((long long)a0.x); t1 = __longlong_as_double((long long)a0.y); }
            if (in1) { ok = ok && a1.x != PSYNC_SENT && a1.y != PSYNC_SENT; t0 += __longlong_as_double((long long)a1.x); t1 += __longlong_as_double((long long)a1.y); }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT ||
                ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
            for (int q = 0; q < ((backoff >> 8) & 0xff); ++q) __builtin_amdgcn_s_sleep(1);
        }
    }
    if (tr && threadIdx.x == 0) tr[7] = (__builtin_readcyclecounter() << 4) + (unsigned long long)(spins < 15 ? spins : 15);
    // lanes l, l ^ 4, l ^ 8, l ^ 12 of a row, then the four rows: the sixteen lanes that hold the same value pair
    t0 += msdp_dpp<0x124>(t0); t1 += msdp_dpp<0x124>(t1);         // row_ror:4
    t0 += msdp_dpp<0x128>(t0); t1 += msdp_dpp<0x128>(t1);         // row_ror:8
    t0 = msdp_rowpair_sum<16>(t0); t1 = msdp_rowpair_sum<16>(t1);
    t0 = msdp_rowpair_sum<32>(t0); t1 = msdp_rowpair_sum<32>(t1);
    if (lane < 4) { shp[w * 8 + 2 * lane] = t0; shp[w * 8 + 2 * lane + 1] = t1; }
    if (lane == 0) {
        shb8[8 + w] = fail ? 1.0 : 0.0;
        if (fail) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // every wave's poll has returned: ALL workgroups have posted this generation (a wave alone has only seen the posts of its 32
    // workgroups: on_ready() -- whatever needs every workgroup's stores -- must not run before this barrier), i.e. finished reading
    // the previous one -- my lines of it back to the sentinel
    on_ready();
    if (w == 0)
        __hip_atomic_store(slots + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID +
                               (size_t)(lane >> 3) * PSYNC_NV * MSDP_MAX_GRID + (size_t)bid * 8 + (lane & 7),
                           PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // value i = the sum over the waves, formed by lane i (every wave does the same), then broadcast
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < PWAVES; ++k) s += shp[k * 8 + (lane & 7)];
    const double fl = shb8[8 + (lane & 7)];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = msdp_readlane(s, i);
    return __builtin_amdgcn_ballot_w64(fl != 0.0) == 0ULL;
}

// A barrier on the slots of psync8_lines -- a post without values (round 6, the TR tail of the fused launch): one generation of the same
// ring, so posts, polls and sentinel resets interleave with the reductions around it.  The counter barrier of msdp_psync.h costs 3.3 us
// on 216 workgroups (profiles/r6_fused_timeline_p32_before.md: 27 read-modify-writes per counter, one behind the other at the memory
// side), this one what a reduction costs minus its sums.  The caller's stores are performed before the post (the wait is in here).
__device__ __forceinline__ bool pbar8_lines(unsigned long long* slots, unsigned gen, int G, double* shb8, int* err, int bid, int backoff) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long* gbase = slots + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;
    if (w == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(gbase + (size_t)(lane >> 3) * PSYNC_NV * MSDP_MAX_GRID + (size_t)bid * 8 + (lane & 7), 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long* p0 = gbase + (size_t)(bid & (PSYNC_REP - 1)) * PSYNC_NV * MSDP_MAX_GRID + (size_t)(32 * w + (lane >> 2)) * 8 + 2 * (lane & 3);
    const bool in0 = 32 * w + (lane >> 2) < G, in1 = 32 * w + 16 + (lane >> 2) < G;
    int spins = 0;
    bool fail = false;
    const int first = ((backoff >> 16) & 0xff) ? ((backoff >> 16) & 0xff) : (backoff & 0xff);
    for (int q = 0; q < first; ++q) __builtin_amdgcn_s_sleep(1);
    if (__builtin_amdgcn_ballot_w64(in0) != 0ULL) {
        for (;;) {
            v2ul a0, a1;
            asm volatile(
                "global_load_dwordx4 %0, %2, off sc1\n\t"
                "global_load_dwordx4 %1, %2, off offset:1024 sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(a0), "=&v"(a1)
                : "v"(p0)
                : "memory");
            bool ok = true;
            if (in0) ok = a0.x != PSYNC_SENT && a0.y != PSYNC_SENT;
            if (in1) ok = ok && a1.x != PSYNC_SENT && a1.y != PSYNC_SENT;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT ||
                ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
            for (int q = 0; q < ((backoff >> 8) & 0xff); ++q) __builtin_amdgcn_s_sleep(1);
        }
    }
    if (lane == 0) {
        shb8[8 + w] = fail ? 1.0 : 0.0;
        if (fail) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (w == 0)
        __hip_atomic_store(slots + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID +
                               (size_t)(lane >> 3) * PSYNC_NV * MSDP_MAX_GRID + (size_t)bid * 8 + (lane & 7),
                           PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double fl = shb8[8 + (lane & 7)];
    return __builtin_amdgcn_ballot_w64(fl != 0.0) == 0ULL;
}

// The eight-value reduction in its two-level form (msdp_psync.h psync2, members that own a device each): the wave butterfly of psync8,
// the member's leader polls the eight local value arrays with its eight waves and pushes the member's eight sums -- ONE 64-byte line --
// into every member's block; everybody polls the N lines of its own block.  One cross-device hop per trip.
template <class F>
__device__ __forceinline__ bool psync2_8(unsigned long long* blk, unsigned long long* const* peers, int N, int me, int ri, unsigned gen, int G,
                                         double (&v)[8], double* sh8, double* shb8, int* err, int bid, int backoff, F on_ready) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const bool h8 = (lane & 8) != 0;
        double a[4], b[2], x;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = msdp_swap_add<32>(v[k], v[4 + k]);
#pragma unroll
        for (int k = 0; k < 2; ++k) b[k] = msdp_swap_add<16>(a[k], a[2 + k]);
        x = (h8 ? b[1] : b[0]) + msdp_dpp<0x128>(h8 ? b[0] : b[1]);
        x += msdp_dpp<MSDP_DPP_XOR1>(x); x += msdp_dpp<MSDP_DPP_XOR2>(x); x += msdp_dpp<MSDP_DPP_HALF_MIRROR>(x);
        if ((lane & 7) == 0) sh8[(lane >> 3) * PWAVES + w] = x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the caller's row stores (and pushes) are performed before the post
    __syncthreads();
    unsigned long long* ls = blk + (size_t)ri * XR2_REGION;
    unsigned long long* ml = ls + PSYNC_REGION;
    unsigned long long* gbase = ls + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;   // replica 0
    if (w == 0 && lane < 8) {
        double s = 0.0;
        for (int i = 0; i < PWAVES; ++i) s += sh8[lane * PWAVES + i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(gbase + (size_t)lane * MSDP_MAX_GRID + bid, (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool fail = false;
    if (bid == 0) {
        // the member's leader: wave w takes value array w of the local slots (the order of psync8) and hands the sum to every member
        const unsigned long long* p0 = gbase + (size_t)w * MSDP_MAX_GRID + lane;
        double r0 = 0.0;
        int spins = 0;
        for (;;) {
            unsigned long long b0[4];
            asm volatile(
                "global_load_dwordx2 %0, %4, off sc1\n\t"
                "global_load_dwordx2 %1, %4, off offset:512 sc1\n\t"
                "global_load_dwordx2 %2, %4, off offset:1024 sc1\n\t"
                "global_load_dwordx2 %3, %4, off offset:1536 sc1\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(b0[0]), "=&v"(b0[1]), "=&v"(b0[2]), "=&v"(b0[3])
                : "v"(p0)
                : "memory");
            bool ok = true;
            double t0 = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (lane + 64 * q < G) {
                    ok = ok && b0[q] != PSYNC_SENT;
                    t0 += __longlong_as_double((long long)b0[q]);
                }
            }
            r0 = t0;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
        }
        r0 = msdp_wave_sum(r0);
        if (!fail && lane < N)
            __hip_atomic_store(peers[lane] + (size_t)ri * XR2_REGION + PSYNC_REGION + ((size_t)(gen % PSYNC_GEN) * 8 + me) * 8 + w,
                               (unsigned long long)__double_as_longlong(r0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (fail) xr2_fail(blk, peers, N, err);
        if (w > 0 && lane == 0) shb8[8 + w] = fail ? 1.0 : 0.0;
    }
    if (w == 0) {
        const unsigned long long* p = ml + (size_t)(gen % PSYNC_GEN) * 64 + lane;
        const bool need = (lane >> 3) < N;
        const int first = bid == 0 ? 0 : (((backoff >> 16) & 0xff) ? ((backoff >> 16) & 0xff) : (backoff & 0xff));
        for (int q = 0; q < first; ++q) __builtin_amdgcn_s_sleep(1);
        unsigned long long x = 0ULL;
        int spins = 0;
        for (;;) {
            x = ld_u64_sys(p);
            if (__builtin_amdgcn_ballot_w64(need && x == PSYNC_SENT) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
            for (int q = 0; q < ((backoff >> 8) & 0xff); ++q) __builtin_amdgcn_s_sleep(1);
        }
        double t = need ? __longlong_as_double((long long)x) : 0.0;
        t += msdp_dpp<0x128>(t);
        t = msdp_rowpair_sum<16>(t);
        t = msdp_rowpair_sum<32>(t);
        if (lane < 8) { shb8[lane] = t; if (bid != 0 && lane > 0) shb8[8 + lane] = 0.0; }
        if (lane == 0) shb8[8] = fail ? 1.0 : 0.0;
        if (fail) xr2_fail(blk, peers, N, err);
        if (lane < PSYNC_NV)
            __hip_atomic_store(ls + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID + (size_t)lane * MSDP_MAX_GRID + bid,
                               PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bid == 0)
            __hip_atomic_store(ml + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * 64 + lane, PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    on_ready();
    double bad = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = msdp_readlane(shb8[i], 0); bad += msdp_readlane(shb8[8 + i], 0); }
    return bad == 0.0;
}

// TRACE stamps of this form: 0 top of the trip (gather about to be issued), 1 products, Hmd and the eight partial sums formed, rows of
// Hmd stored; inside the reduction 4 wave butterfly done, 2 stores performed + workgroup barrier, 3 posted + slept, 7 wave 0's poll
// returned; 5 the reduction returned, 6 new direction formed (end of the trip)
// FUSE: the whole trustregions() loop in this launch (trustregions.m:441-767), as in k_tcg_persist_obl<..., FUSE = true>: tCG, retraction,
// cost / gradient at the proposal, the accept / reject decision, iterated until the gradient norm or the iteration cap stops it.
// Which of the two reductions: the line layout (default; G81 p = 32 4.82 us per trip against 4.99, p = 16 3.54 against 3.74, with the
// cross-wave sum formed by eight lanes instead of every thread reading 64 partials -- the first version of it lost to the arrays)
#ifndef MSDP_PIPE_LINES
#define MSDP_PIPE_LINES 1
#endif
#if MSDP_PIPE_LINES
#define PSYNC8(slots, gen, G, v, sh8, shb8, err, bid, backoff, tr, f) psync8_lines(slots, gen, G, v, sh8, shp, shb8, err, bid, backoff, tr, f)
#else
#define PSYNC8(slots, gen, G, v, sh8, shb8, err, bid, backoff, tr, f) psync8(slots, gen, G, v, sh8, shb8, err, bid, backoff, tr, f)
#endif
// XRM (round 6): 0 one rank; 1 / 2 = a member of a group of ranks whose launches run ONE tCG together (the cross-rank form of
// msdp_persist.hip XR with this trip): 1 the flat reductions over all members' workgroups (members that share a device, N <= 4),
// 2 the two-level ones (psync2_8: members on devices of their own, N <= 8).  The rows travel as there: every member's exchange buffer holds
// its rows and a slot per foreign row its rows of C reference (buffer-local column indices d.xr_ellc), the owner of a boundary row
// stores it into the slots of the members that reference it (d.xr_paddr) -- here in four regions (H md alternating, the refresh rows),
// and the first direction (the gradient, which every member keeps to itself) is published behind one barrier.  XRM = 2: the gathers are
// system-scope loads (a halo slot may have been stored by another device), the pushed rows and member sums system-scope stores.
// EW = 0, EP > 1 (round 6): CSR rows with entry-parallel lanes -- the 64 / LPR lane groups of a wave share ONE row (group epi = 0 owns it:
// registers, sums, stores) and split its entries (msdp_persist.hip, option persist_ep); no ELL copy, no local columns, no gather requested
// inside the reduction: a trip's products come from ONE batch of gathers per lane for rows of up to 8 x EP entries.
template <int LPR, int EW, int R, bool TRACE, bool FUSE, int XRM = 0, int EP = 1>
__device__ __forceinline__ void tcg_pipe_body(const Dev& d, unsigned long long* slots, int* err, const int bx) {
    constexpr bool XR = XRM != 0, XTWO = XRM == 2;
    constexpr bool CSR = EW == 0;
    static_assert(!XR || (!FUSE && !TRACE), "cross-rank instances: per-iteration launches");
    static_assert(R <= 5, "pipelined trip: every vector in registers");
    static_assert(CSR ? (EP == 64 / LPR && !XR && !TRACE) : EP == 1, "CSR rows: one row per wave, one rank");
    static_assert(PSYNC_NV == 8 && PSYNC_REP * PSYNC_NV == 64, "psync8 posts one slot per lane of wave 0");
    extern __shared__ double lds[];
    __shared__ double sh8[8 * PWAVES];
    __shared__ double shp[8 * PWAVES];
    __shared__ double shb8[16];
    constexpr int RPW = 64 / (LPR * EP);
    constexpr int RSTEP = PWAVES * RPW;
    constexpr int ROWS = R * RSTEP;
    // (FUSE: Ys / Gs / eGs and YPs / GPs / EGPs are the CURRENT point and the PROPOSAL -- two sets of LDS buffers that change roles when a
    // step is accepted, round 6; the names are pointers, not fixed addresses)
    double2* Ys = reinterpret_cast<double2*>(lds);                 // [R][PB]
    double2* Gs = Ys + R * PB;                                     // [R][PB]
    double* eGs = reinterpret_cast<double*>(Gs + R * PB);          // [ROWS]
    double* vs = eGs + ROWS;                                       // [EW][ROWS]
    int* cs = reinterpret_cast<int*>(vs + EW * ROWS);              // [EW][ROWS]
    // Where an entry's row lives (round 5): most neighbours of a row belong to the SAME workgroup when C is banded / a grid in its
    // natural order, and the diagonal entry is the lane's own row -- those need no trip through the texture path, which is what bounds
    // the gather (a 16-byte-per-lane load occupies it for 16 cycles: 8 waves x R x EW loads = 0.8 us of a 4.9-us trip at p = 32).
    // ls = LDS element of the row in HQs (+ class in bits 24..25: 0 in this workgroup, 1 the lane's own row, 2 empty slot) or -1 (another
    // workgroup's row: exchange buffer); HQs = this workgroup's own rows of Hmd, two halves alternating like the buffer's.
    int* ls = cs + EW * ROWS;                                      // [EW][ROWS]
    double2* HQs = reinterpret_cast<double2*>(ls + EW * ROWS);     // [2][R][PB]
    // FUSE only: the proposal point, its gradient and eG (round 6: buffers of their own behind HQs -- sharing its space meant a copy of
    // 2 x R x PB double2 per accepted step, 1.1 us of every TR iteration)
    unsigned long long* pds = reinterpret_cast<unsigned long long*>(HQs + 2 * R * PB);   // XR: [2][ROWS] (never FUSE: the space is free)
    double2* YPs = HQs + 2 * R * PB;                               // [R][PB]
    double2* GPs = YPs + R * PB;                                   // [R][PB]
    double* EGPs = reinterpret_cast<double*>(GPs + R * PB);        // [ROWS]

    const Ctl* c = d.ctl;
    const bool lead = bx == 0 && threadIdx.x == 0;
    const int k_tr = c->k;
    if (c->done) {
        if (lead) {
            frame_store(&d.F[0], c->gg, c->gg, 0.0, 0.0, 0.0, sqrt(c->gg), 0.0, 0.0, 0, 0, 5, 0, 0, 1);
            d.ctl->tcg_running = 0;
            msdp_publish(d, k_tr, 0, 0);
        }
        return;
    }
    if (lead && !FUSE) msdp_publish(d, k_tr, 0, 1);
    const int bid = (XR && !XTWO) ? d.xr_gid0 + bx : bx;
    const int GS = (XR && !XTWO) ? d.xr_gtot : d.G;                // the workgroups that synchronise (two-level: this member's)
    const int xri = k_tr & 1;                                      // XR: the slot region of this launch (they alternate with the TR iteration)
    __shared__ unsigned long long* shpeer[XTWO ? 8 : 1];
    if (XTWO) {
        if (threadIdx.x < 8) shpeer[threadIdx.x] = threadIdx.x < d.xr2_n ? d.xr2_peers[threadIdx.x] : d.xr2_blk;
        psync2_reset_other(d.xr2_blk, xri ^ 1, bid, GS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (XR) {
        unsigned long long* other = slots + (size_t)(xri ^ 1) * PSYNC_REGION;
        slots += (size_t)xri * PSYNC_REGION;
        psync_reset_other(other, bid, GS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // performed before this workgroup's first post of this launch
    } else if (!FUSE) psync_reset_other(slots + PSYNC_REGION, bid, GS);   // region B belongs to the TR-iteration tail kernel
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi, 0, bx);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (LPR - 1), rsub = lane / (LPR * EP);
    const int epi = (lane / LPR) & (EP - 1);                       // CSR: which of the row's lane groups (0 owns the row)
    const bool colok_g = 2 * sub < d.ld;                           // the lane has columns (gathers)
    const bool colok = colok_g && epi == 0;                        // ... and owns them (registers, sums, stores)
    int cur = c->cur;
    const bool bench = c->bench_mode != 0;
    double Delta = c->Delta;
    const double kappa = c->kappa, theta = c->theta;
    const int mininner = c->mininner, maxinner = c->maxinner;
    double gg = c->gg;
    // trust-region level state (FUSE).  Round 6: the options and the counters are workgroup-uniform values (scalar registers) read ONCE --
    // read per TR iteration they were a chain of scalar loads and, for the counters, read-modify-writes of global memory by the lead
    // thread on the critical path between the iteration's reduction and the next tCG
    double fx = c->fx;
    int k_it = c->k;
    const double Delta_bar = c->Delta_bar, rho_prime = c->rho_prime, rho_reg_opt = c->rho_reg, tolgradnorm = c->tolgradnorm;
    const int maxiter = c->maxiter;
    int n_acc = c->accepted, n_rej = c->rejected, n_hv = c->hessvecs, n_ce = c->cost_evals;
    const double* __restrict__ Yl = cur ? d.Y[1] : d.Y[0];
    const double* __restrict__ gl = cur ? d.Gr[1] : d.Gr[0];
    const double* __restrict__ eGl = cur ? d.eG[1] : d.eG[0];
    const int slot0 = wave * RPW + rsub;
#define SLOT(r) ((r) * RSTEP + slot0)
#define ROW(r) (lo + SLOT(r))
#define ROK(r) (ROW(r) < hi)
#define OK(r) (ROK(r) && colok)
    const int refresh = c->pipe_refresh;
    // (bits 16..23 = this reduction's own figure; the option psync8_backoff fills them where psync_backoff leaves them empty)
    const int backoff = (c->psync_backoff & 0xff00ffff) | ((((c->psync_backoff >> 16) & 0xff) ? ((c->psync_backoff >> 16) & 0xff) : (c->psync8_backoff & 0xff)) << 16);
    const double2 zz = make_double2(0.0, 0.0);
    constexpr bool LOC = !CSR && R * EW <= 15;                     // (four row slots: the source selection costs registers that spill)
    constexpr bool MULTI = LOC && !(FUSE && LPR >= 16);            // three instances of the trip loop (per-wave local columns), see below
    double2 eta[R], rr[R], md[R], hmd[R], cmd[R], ctr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool rok = ROK(r);
        const int rc = rok ? ROW(r) : lo;
        const int64_t o = (int64_t)rc * d.ld + (colok ? 2 * sub : 0);
        double2 y = ld2(Yl + o), g = ld2(gl + o);
        if (!OK(r)) { y = zz; g = zz; }
        Ys[r * PB + threadIdx.x] = y; Gs[r * PB + threadIdx.x] = g;
        eta[r] = zz; rr[r] = g; md[r] = g; hmd[r] = zz; cmd[r] = zz; ctr[r] = zz;       // tCG.m:102-157
    }
    // (the ELL rows in a loop of their own, NOT unrolled: nothing in it is indexed by the row slot but LDS, and unrolled next to the
    // register set-up above its sort was where the kernel's first spills came from)
#pragma unroll 1
    for (int r = 0; r < (CSR ? 0 : R); ++r) {
        const bool rok = ROK(r);
        const int rc = rok ? ROW(r) : lo;
        const double egv = eGl[rc];
        int cw[EW > 0 ? EW : 1];
        double vw[EW > 0 ? EW : 1];
#pragma unroll
        for (int w = 0; w < EW; ++w) {
            cw[w] = XR ? d.xr_ellc[(int64_t)w * d.ell_stride + rc] : d.ellc[(int64_t)w * d.ell_stride + rc];
            vw[w] = d.ellv[(int64_t)w * d.ell_stride + rc];
        }
        // the entries of the row ordered own row, rows of this workgroup, other rows, empty slots: a column of the ELL block then
        // holds ONE kind for (nearly) all rows of a wave, and the wave picks the source per column, not per lane
        int kw[EW > 0 ? EW : 1];
#pragma unroll
        for (int w = 0; w < EW; ++w) {
            if (!rok) vw[w] = 0.0;
            kw[w] = vw[w] == 0.0 ? 3 : (cw[w] == rc ? 0 : ((cw[w] >= lo && cw[w] < hi) ? 1 : 2));
        }
#pragma unroll
        for (int a = 0; a < (LOC ? EW - 1 : 0); ++a)
#pragma unroll
            for (int b2 = 0; b2 < EW - 1 - a; ++b2)
                if (kw[b2] > kw[b2 + 1]) {
                    const int tk = kw[b2]; kw[b2] = kw[b2 + 1]; kw[b2 + 1] = tk;
                    const int tc = cw[b2]; cw[b2] = cw[b2 + 1]; cw[b2 + 1] = tc;
                    const double tv = vw[b2]; vw[b2] = vw[b2 + 1]; vw[b2 + 1] = tv;
                }
        if (XR && sub == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)                            // push targets of this row (the slot's address in region 0 of that member's buffer; 0: none)
                pds[t * ROWS + SLOT(r)] = rok ? d.xr_paddr[(int64_t)t * d.n_loc + rc] : 0ULL;
        }
        if (sub == 0) {
            eGs[SLOT(r)] = rok ? egv : 0.0;
#pragma unroll
            for (int w = 0; w < EW; ++w) {
                cs[w * ROWS + SLOT(r)] = cw[w];
                vs[w * ROWS + SLOT(r)] = vw[w];
                const int li = cw[w] - lo;
                const int el = (li / RSTEP) * PB + (li % RSTEP) * LPR;          // element of row li's lane 0 in a half of HQs
                ls[w * ROWS + SLOT(r)] = kw[w] == 2 ? -1 : (kw[w] == 3 ? (2 << 24) : ((kw[w] == 0 ? (1 << 24) : 0) | el));
            }
        }
    }
    if (CSR) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (sub == 0 && epi == 0) eGs[SLOT(r)] = ROK(r) ? eGl[ROW(r)] : 0.0;
    }
    __syncthreads();
    // nl = the leading columns whose rows ALL live in this workgroup for every row of this WAVE (own row, neighbour in the chunk, or an
    // empty slot): read from LDS; the columns behind them go through the exchange buffer (which holds every row: also right for a column
    // that mixes kinds).  Three instances of the trip loop: nl = 3 (a grid row away from the chunk's ends: own row, two neighbours in
    // the chunk, two far ones), 2 (a wave that holds an end of the chunk), 0 (everything else, pipe_local = 0, four row slots).
    int nl = 0;
    if (LOC && c->pipe_local != 0) {
        bool lead_ok = true;
#pragma unroll
        for (int w = 0; w < EW; ++w) {
            bool glob = false;
#pragma unroll
            for (int r = 0; r < R; ++r) glob = glob || ls[w * ROWS + SLOT(r)] < 0;
            lead_ok = lead_ok && __builtin_amdgcn_ballot_w64(glob) == 0ULL;
            if (lead_ok) nl = w + 1;
        }
    }
    unsigned gen = 0;
    // (XR: a region of a member's buffer = its rows + the slots of the foreign rows it references)
    const unsigned half_bytes = XR ? (unsigned)(((size_t)d.xr_cap + (size_t)d.xr_halo) * d.ld * sizeof(double)) : (unsigned)((size_t)d.n_loc * d.ld * sizeof(double));
    // exchange buffer: halves 0 / 1 = the rows of Hmd (alternating trips), 2 = the rows of md' and 3 = those of tangent(r') of a refresh trip;
    // FUSE (round 6): 4 = the rows of the proposal point, 5 / 6 = the gradient rows of the point in slot 0 / 1 -- the TR tail's exchanges
    // went through d.Y / d.Gr of the proposal slot (ordinary device memory, sc1 accesses) and its two cold gathers took 4.2 and 2.8 us
    // where a trip's gather from this buffer (fine-grained memory) is done in 1.4 with its arithmetic (profiles/r6_fused_timeline_p32_*.md)
    const __amdgpu_buffer_rsrc_t rs_md = XR ? __builtin_amdgcn_make_buffer_rsrc(d.xr_rows[0], 0, 4u * half_bytes, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc(d.mdx, 0, (FUSE ? 7u : 4u) * half_bytes, 0x00020000);
    // XR: this lane's 16 bytes of local row slot r of region `reg` also go to the members that reference the row
    bool xr_has_push = false, xr_has_push2 = false;                // wave-uniform: some row of this wave is referenced by another member / by two
    if (XR) {
        bool any = false, any2 = false;
#pragma unroll
        for (int r = 0; r < R; ++r) { any = any || pds[SLOT(r)] != 0ULL; any2 = any2 || pds[ROWS + SLOT(r)] != 0ULL; }
        xr_has_push2 = __builtin_amdgcn_ballot_w64(any2) != 0ULL;
        xr_has_push = xr_has_push2 || __builtin_amdgcn_ballot_w64(any) != 0ULL;
    }
    auto xr_push = [&](int r, unsigned reg, double2 val) {
        if (!XR || !xr_has_push) return;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !xr_has_push2) continue;
            const unsigned long long a = pds[t * ROWS + SLOT(r)];
            if (a != 0ULL && OK(r)) {
                double* ptr = reinterpret_cast<double*>(a + (unsigned long long)reg * half_bytes) + 2 * sub;
                if (XTWO) {                                        // the slot may live on another device: system scope
                    __hip_atomic_store(ptr, val.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(ptr + 1, val.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __hip_atomic_store(ptr, val.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ptr + 1, val.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    // the gradient rows of the current point: where an earlier launch left them (d.Gr) until a step has been accepted in THIS launch
    __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(cur ? d.Gr[1] : d.Gr[0], 0, half_bytes, 0x00020000);
    unsigned g_base = 0u;
    const unsigned gld = (unsigned)d.ld, gcol = colok ? 2 * sub : 0;
    // CSR rows: C[row, :] * X[:, my columns] with X read through rs at byte offset base -- lane group epi takes the entries s0 + epi,
    // s0 + epi + EP, ..., eight of them in flight per lane; the groups' partial products are added, the owner keeps the sum.
    // The lane's first eight (column, value) pairs of every row slot are static: they stay in registers for the whole launch where there
    // are two row slots (eight lanes per row: 48 registers) -- a trip's gathers then wait for nothing but the rows themselves.
    constexpr int CSR_CB = 8;
    constexpr bool CSR_KEEP = CSR && R <= 2 && !FUSE;              // (the fused launch has no registers left for them: 244 bytes of scratch; kept
                                                                   //  in LDS instead, one element per thread: 172 bytes, and G1's RTR time does not move)
    int ckeep[CSR_KEEP ? R : 1][CSR_CB];
    double vkeep[CSR_KEEP ? R : 1][CSR_CB];
    if (CSR_KEEP) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool rok = ROK(r);
            const int s0 = rok ? d.rowptr[ROW(r)] : 0, s1 = rok ? d.rowptr[ROW(r) + 1] : 0;
#pragma unroll
            for (int u = 0; u < CSR_CB; ++u) {
                const int k = s0 + epi + u * EP;
                const bool in = k < s1;
                ckeep[r][u] = in ? d.colind[k] : (rok ? ROW(r) : lo);
                vkeep[r][u] = in ? d.cval[k] : 0.0;
            }
        }
    }
    auto csr_gather = [&](int r, __amdgpu_buffer_rsrc_t rs, unsigned base) -> double2 {
        constexpr int CB = CSR_CB;
        double2 acc = make_double2(0.0, 0.0);
        const bool rok = ROK(r);
        const int s0 = (CSR && rok) ? d.rowptr[ROW(r)] : 0, s1 = (CSR && rok) ? d.rowptr[ROW(r) + 1] : 0;
        const unsigned gcg = colok_g ? 2 * sub : 0;
        int kfirst = s0 + epi;
        if (CSR_KEEP) {
            double2 x[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) x[u] = ld2_sc1(rs, base + ((unsigned)ckeep[CSR_KEEP ? r : 0][u] * gld + gcg) * 8u);
#pragma unroll
            for (int u = 0; u < CB; ++u) { const double cv = vkeep[CSR_KEEP ? r : 0][u]; acc.x = fma(cv, x[u].x, acc.x); acc.y = fma(cv, x[u].y, acc.y); }
            kfirst += CB * EP;
        }
        for (int k0 = kfirst; __builtin_amdgcn_ballot_w64(k0 < s1) != 0ULL; k0 += CB * EP) {
            double2 x[CB];
            double cvk[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int k = k0 + u * EP;
                const bool in = k < s1;
                const int cidx = in ? d.colind[k] : (rok ? ROW(r) : lo);
                cvk[u] = in ? d.cval[k] : 0.0;
                x[u] = ld2_sc1(rs, base + ((unsigned)cidx * gld + gcg) * 8u);
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) { acc.x = fma(cvk[u], x[u].x, acc.x); acc.y = fma(cvk[u], x[u].y, acc.y); }
        }
#pragma unroll
        for (int m = LPR; m < LPR * EP; m <<= 1) { acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); }
        if (!colok) acc = make_double2(0.0, 0.0);
        return acc;
    };

    double z_r = gg, d_Pd = gg, e_Pd = 0.0, e_Pe = 0.0, model_value = 0.0, alpha = 0.0, beta = 0.0;
    double norm_r0 = sqrt(gg);
    int j = 0, stop = 5;
    // TRACE && FUSE (msdp_debug_persist_trace with reps <= 0): thread 0 of every workgroup stamps the phases of the first MSDP_TRACE_NJ TR
    // iterations of the call into d.trace[(workgroup * NJ + iteration) * 8 + phase]: 0 iteration starts (tCG.m:102-157), 1 first trip's
    // products and partial sums formed, 2 the tCG has ended (bits 56..63: its trips), 3 proposal rows stored and performed, 4 barrier
    // returned, 5 cost / gradient rows of the proposal formed, 6 the iteration's reduction returned, 7 decision taken, point committed
    const int k_it0 = c->k;
#define FSTAMP(ph) do { if (TRACE && FUSE && threadIdx.x == 0 && k_it - k_it0 < MSDP_TRACE_NJ) \
        d.trace[((size_t)bx * MSDP_TRACE_NJ + (k_it - k_it0)) * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
    bool first = true, direct = false, failed = false;
    unsigned xq = 0;                                               // the half this trip's rows of Hmd go to
    bool have_x = false;
  // TRACE with FUSE (round 6): the stamps are those of the TR iteration around the tCG, not of the trips (FSTAMP below)
// (... plus, in a second block of the trace buffer behind the G x NJ x 8 stamps of the iterations, the trips' own phase stamps -- those of
// the per-iteration instance -- for the trips 0 .. NJ - 1 of ONE TR iteration, MSDP_TRACE_KSEL)
#define MSDP_TRACE_KSEL 7
#define FTRIP_ON (TRACE && FUSE && k_it - k_it0 == MSDP_TRACE_KSEL && j < MSDP_TRACE_NJ)
#define FTRIP_PTR (d.trace + (size_t)GS * MSDP_TRACE_NJ * 8 + ((size_t)bx * MSDP_TRACE_NJ + j) * 8)
#define PTSTAMP(ph) do { if (!FUSE) { TSTAMP(ph); } else if (FTRIP_ON && threadIdx.x == 0) FTRIP_PTR[ph] = __builtin_readcyclecounter(); } while (0)
  auto trips = [&](auto nlc) {
    constexpr int NL = decltype(nlc)::value;                       // columns [0, NL) from LDS, [NL, EW) through the buffer
    constexpr int NG = EW - NL;
    // the next trip's rows requested inside the reduction: where they are few enough to stay in flight across the trip's arithmetic
#ifndef MSDP_PIPE_VARIANT
#define MSDP_PIPE_VARIANT 0
#endif
    constexpr bool PREF = !CSR && R * NG <= ((FUSE && MSDP_PIPE_VARIANT == 0) ? 9 : 15);
    // SPLIT: the four sums without H md under the gather, the other four behind the row stores (A/B builds: variant 1 = the gather
    // prefetched in the fused launch too, round 5's order; variant 2 = prefetched AND split)
    constexpr bool SPLIT = (MSDP_PIPE_VARIANT == 2 && FUSE) ? true : !PREF;
    double2 X[R][NG > 0 ? NG : 1];                                 // the gathered rows (in flight across the end of a trip)
    for (;;) {
        PTSTAMP(0);
        // ---- the products: C md of this trip (cmd) and C tangent(r) (ctr)
        // columns [w0, w0 + NG) of the rows my rows reference, requested from (rs, base) / folded into acc
#define PIPE_ISSUE(rs, base, w0) do { \
            _Pragma("unroll") for (int r = 0; r < R; ++r) \
            _Pragma("unroll") for (int w = 0; w < NG; ++w) { \
                if ((w0) + w < EW) { \
                    const int cidx = cs[((w0) + w) * ROWS + SLOT(r)]; \
                    X[r][w] = ld2_cp<XTWO ? 17 : MSDP_CPOL_SC1>((rs), (base) + ((unsigned)cidx * gld + gcol) * 8u); } } } while (0)
#define PIPE_FOLDX(r, acc, w0) do { \
            _Pragma("unroll") for (int w = 0; w < NG; ++w) { \
                if ((w0) + w < EW) { \
                    const double vv = vs[((w0) + w) * ROWS + SLOT(r)]; \
                    (acc).x = fma(vv, X[r][w].x, (acc).x); (acc).y = fma(vv, X[r][w].y, (acc).y); } } } while (0)
        // the leading NL columns from the workgroup's own rows in LDS (src: a [R][PB] array -- a half of HQs, Gs, YPs)
#define PIPE_FOLDL(r, acc, src) do { \
            _Pragma("unroll") for (int w = 0; w < NL; ++w) { \
                const double vv = vs[w * ROWS + SLOT(r)]; \
                const double2 xx = (src)[(ls[w * ROWS + SLOT(r)] & 0xffffff) + sub]; \
                (acc).x = fma(vv, xx.x, (acc).x); (acc).y = fma(vv, xx.y, (acc).y); } } while (0)
        // all EW columns through the buffer (the gradient / the refresh vectors are not in LDS): passes of NG columns
#define PIPE_GATHER_ALL(rs, base, dst) do { \
            if (CSR) { _Pragma("unroll") for (int r = 0; r < R; ++r) (dst)[r] = csr_gather(r, (rs), (base)); break; } \
            _Pragma("unroll") for (int r = 0; r < R; ++r) (dst)[r] = zz; \
            _Pragma("unroll") for (int w0 = 0; w0 < EW; w0 += (NG > 0 ? NG : 1)) { \
                PIPE_ISSUE(rs, base, w0); \
                _Pragma("unroll") for (int r = 0; r < R; ++r) PIPE_FOLDX(r, (dst)[r], w0); } \
            if (!colok) { _Pragma("unroll") for (int r = 0; r < R; ++r) (dst)[r] = zz; } } while (0)
        // the eight partial sums of this trip's reduction.  Round 6, the instances that do NOT have the next gather requested inside the
        // reduction (!PREF: the fused launch at 16 lanes per row): four sums -- <md, g>, <md, Heta>, <r, r>, the model value -- do not
        // involve H md and are formed while the gathers of this trip are in flight, the other four behind the rows of H md and their
        // stores (whose drain they run under): G81 p = 32, the fused launch 5 426 -> 5 137 us per call, 182 100 -> 192 400 Hess-vec/s.
        // With the gather prefetched the same order LOSES (the per-iteration <16, 5, 3> instance 4.82 -> 4.98 us per trip, <8, 5, 2>
        // 3.59 -> 3.77): those keep the one loop of round 5.
        double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        auto pre_sums = [&]() {
            if (!SPLIT) return;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double2 g = Gs[r * PB + threadIdx.x], mdr = md[r], e0 = eta[r], rv = rr[r];
                const double2 rg = make_double2(rv.x - g.x, rv.y - g.y);                  // Heta (:220)
                v[3] += mdr.x * g.x + mdr.y * g.y;                                        // <md, g>
                v[5] += mdr.x * rg.x + mdr.y * rg.y;                                      // <md, Heta>
                v[6] += rv.x * rv.x + rv.y * rv.y;                                        // <r, r>      (:241 of the trip before)
                v[7] += (e0.x * g.x + e0.y * g.y) + 0.5 * (e0.x * rg.x + e0.y * rg.y);    // model value (:227 of the trip before)
            }
        };
        if (first) {
            // the first direction = the gradient (tangent): r = md = grad.  Its rows: in global memory since an earlier launch, in the
            // exchange buffer since the TR tail that proposed this point
            pre_sums();
            PIPE_GATHER_ALL(rs_g, g_base, cmd);
#pragma unroll
            for (int r = 0; r < R; ++r) ctr[r] = cmd[r];
        } else {
            if (direct) {
                // the trip before published tangent(r) and md next to Hmd: both products start afresh from direct gathers
                PIPE_GATHER_ALL(rs_md, 3u * half_bytes, ctr);
                PIPE_GATHER_ALL(rs_md, 2u * half_bytes, cmd);
            }
            // the neighbours' rows of last trip's Hmd: requested inside that trip's reduction already (have_x), except behind a refresh
            if (!CSR && !have_x) PIPE_ISSUE(rs_md, (xq ^ 1u) * half_bytes, NL);
            pre_sums();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double2 a = zz;
                if (CSR) a = csr_gather(r, rs_md, (xq ^ 1u) * half_bytes);
                else { PIPE_FOLDL(r, a, HQs + (xq ^ 1u) * R * PB); PIPE_FOLDX(r, a, NL); }
                if (!colok) a = zz;
                ctr[r].x = fma(-alpha, a.x, ctr[r].x); ctr[r].y = fma(-alpha, a.y, ctr[r].y);      // C tangent(r') = C tangent(r) - alpha C Hmd
                cmd[r].x = fma(beta, cmd[r].x, ctr[r].x); cmd[r].y = fma(beta, cmd[r].y, ctr[r].y);  // C md' = C tangent(r') + beta C md
            }
        }
        have_x = false;
        // every `refresh`-th trip publishes tangent(r) and md of ITS start next to Hmd (no barrier of its own: the reduction orders them
        // like the rows of Hmd); the next trip gathers all three
        const bool pub = !first && refresh > 0 && ((j + 1) % refresh) == 0;
        // ---- Hmd = proj(C*md) - md.*eG (tCG.m:163, ManiSDP_onlyunitdiag.m:127-130), its rows to the neighbours, the eight partial sums
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 acc = cmd[r];
            const double2 y = Ys[r * PB + threadIdx.x], mdr = md[r], rv = rr[r];
            const double dot = msdp_group_sum<LPR>(acc.x * y.x + acc.y * y.y);
            const double eg = eGs[SLOT(r)];
            double2 hq = make_double2(acc.x - y.x * dot - mdr.x * eg, acc.y - y.y * dot - mdr.y * eg);
            if (!OK(r)) hq = zz;
            hmd[r] = hq;
            if (MULTI) HQs[xq * R * PB + r * PB + threadIdx.x] = hq;    // (read by the instances with local columns only)
            if (OK(r)) st2_sc1(rs_md, xq * half_bytes + ((unsigned)ROW(r) * gld + 2 * sub) * 8u, hq);
            xr_push(r, xq, hq);
            if (pub) {
                const double dn = msdp_group_sum<LPR>(rv.x * y.x + rv.y * y.y);
                if (OK(r)) {
                    st2_sc1(rs_md, 2u * half_bytes + ((unsigned)ROW(r) * gld + 2 * sub) * 8u, mdr);
                    st2_sc1(rs_md, 3u * half_bytes + ((unsigned)ROW(r) * gld + 2 * sub) * 8u, make_double2(rv.x - y.x * dn, rv.y - y.y * dn));
                }
                xr_push(r, 2u, mdr);
                xr_push(r, 3u, make_double2(rv.x - y.x * dn, rv.y - y.y * dn));
            }
            if (!SPLIT) {                                                                 // (round 5's order: all eight sums row by row)
                const double2 g = Gs[r * PB + threadIdx.x], e0 = eta[r];
                const double2 rg = make_double2(rv.x - g.x, rv.y - g.y);                  // Heta (:220)
                v[0] += mdr.x * hq.x + mdr.y * hq.y;                                      // <md, Hmd>   (:166)
                v[1] += rv.x * hq.x + rv.y * hq.y;                                        // <r, Hmd>
                v[2] += hq.x * hq.x + hq.y * hq.y;                                        // <Hmd, Hmd>
                v[3] += mdr.x * g.x + mdr.y * g.y;                                        // <md, g>
                v[4] += e0.x * hq.x + e0.y * hq.y;                                        // <eta, Hmd>
                v[5] += mdr.x * rg.x + mdr.y * rg.y;                                      // <md, Heta>
                v[6] += rv.x * rv.x + rv.y * rv.y;                                        // <r, r>      (:241 of the trip before)
                v[7] += (e0.x * g.x + e0.y * g.y) + 0.5 * (e0.x * rg.x + e0.y * rg.y);    // model value (:227 of the trip before)
            }
        }
        if (SPLIT) {
            // (every row store of the trip is on its way: the sums that need H md run under their drain)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double2 hq = hmd[r], mdr = md[r], e0 = eta[r], rv = rr[r];
                v[0] += mdr.x * hq.x + mdr.y * hq.y;                                      // <md, Hmd>   (:166)
                v[1] += rv.x * hq.x + rv.y * hq.y;                                        // <r, Hmd>
                v[2] += hq.x * hq.x + hq.y * hq.y;                                        // <Hmd, Hmd>
                v[4] += e0.x * hq.x + e0.y * hq.y;                                        // <eta, Hmd>
            }
        }
        PTSTAMP(1);
        if (TRACE && FUSE && j == 0) { FSTAMP(1); }
        // (my rows of Hmd are performed before I post: the wait sits inside psync8, behind the wave reduction)
        // (the next trip's gather goes out as soon as this wave has seen every workgroup's post: its latency runs under the rest of the
        // reduction and the arithmetic behind it.  Not behind a refresh trip: that one's two direct gathers come first.)
        auto next_gather = [&]() { if (PREF && !pub) { PIPE_ISSUE(rs_md, xq * half_bytes, NL); have_x = true; } };
        if (XTWO) {
            if (!psync2_8(d.xr2_blk, shpeer, d.xr2_n, d.xr2_me, xri, gen++, GS + d.xr2_skip, v, sh8, shb8, err, bid, backoff, next_gather)) { failed = true; break; }
        } else
        if (!PSYNC8(slots, gen++, GS, v, sh8, shb8, err, bid, backoff,
                    (TRACE && !FUSE && j >= MSDP_TRACE_J0 && j < MSDP_TRACE_J0 + MSDP_TRACE_NJ) ? d.trace + ((size_t)bx * MSDP_TRACE_NJ + (j - MSDP_TRACE_J0)) * 8 :
                    (FTRIP_ON ? FTRIP_PTR : nullptr),
                    next_gather)) { failed = true; break; }
        PTSTAMP(5);
        const double d_Hd = v[0];                                                         // :166
        z_r = v[6];
        model_value = v[7];
        alpha = z_r / d_Hd;                                                               // :170
        const double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;         // :173
        if (!bench && (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {                       // :183
            const double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;   // :188
#pragma unroll
            for (int r = 0; r < R; ++r) {
                eta[r].x -= tau * md[r].x; eta[r].y -= tau * md[r].y;                     // :192
                rr[r].x -= tau * hmd[r].x; rr[r].y -= tau * hmd[r].y;                     // :198 (Heta = r - grad)
            }
            stop = (d_Hd <= 0.0) ? 1 : 2;
            ++j;
            break;
        }
        // the trial step's three inner products (tCG.m:215-241), expanded in alpha
        const double new_model = model_value - alpha * v[3] - 0.5 * alpha * (v[4] + v[5]) + 0.5 * alpha * alpha * d_Hd;   // :227
        const double r_r = fmax(z_r - 2.0 * alpha * v[1] + alpha * alpha * v[2], 0.0);                                    // :241
        e_Pe = e_Pe_new;
        if (!bench && new_model >= model_value) { stop = 6; ++j; break; }                 // :228 (eta, Heta stay)
#pragma unroll
        for (int r = 0; r < R; ++r) {                                                     // :233-238
            eta[r].x -= alpha * md[r].x; eta[r].y -= alpha * md[r].y;
            rr[r].x -= alpha * hmd[r].x; rr[r].y -= alpha * hmd[r].y;
        }
        model_value = new_model;
        ++j;
        const double norm_r = sqrt(r_r);
        const double nr0t = (theta == 1.0) ? norm_r0 : pow(norm_r0, theta);
        if (!bench && j >= mininner && norm_r <= norm_r0 * fmin(nr0t, kappa)) {           // :249
            stop = (kappa < nr0t) ? 3 : 4;
            break;
        }
        if (j >= maxinner) break;                                                         // :160 (stop stays 5)
        beta = r_r / z_r;                                                                 // :272
        e_Pd = beta * (e_Pd + alpha * d_Pd);                                              // :286
        d_Pd = r_r + beta * beta * d_Pd;                                                  // :287
        z_r = r_r;
        // ---- mdelta = tangent(r + beta*mdelta)  (:273,283)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 y = Ys[r * PB + threadIdx.x];
            const double2 vv = make_double2(rr[r].x + beta * md[r].x, rr[r].y + beta * md[r].y);
            const double dot = msdp_group_sum<LPR>(vv.x * y.x + vv.y * y.y);
            md[r] = make_double2(vv.x - y.x * dot, vv.y - y.y * dot);
        }
        direct = pub;
        first = false;
        xq ^= 1u;
        { --j; PTSTAMP(6); ++j; }
    }
#undef PIPE_ISSUE
#undef PIPE_FOLDX
#undef PIPE_FOLDL
#undef PIPE_GATHER_ALL
#undef PTSTAMP
#undef FTRIP_ON
#undef FTRIP_PTR
  };
  // (test hook debug_xr_skip, flat form: the first workgroup of the group never posts -- everybody else's bounded spin runs out; the
  // two-level form waits for local workgroups that do not exist, GS + d.xr2_skip)
  if (XR && !XTWO && d.xr2_skip && bid == 0) return;
  if (XR) {
    // the first direction = the gradient, whose rows every member keeps to itself: hand them to the others first (region 2: the refresh
    // trips use it from trip `refresh` on, long behind the first trip's gather)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (OK(r)) st2_sc1(rs_md, 2u * half_bytes + ((unsigned)ROW(r) * gld + 2 * sub) * 8u, md[r]);
        xr_push(r, 2u, md[r]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (XTWO) {
        double z0 = 0.0, z1 = 0.0, z2 = 0.0;
        if (!psync2(d.xr2_blk, shpeer, d.xr2_n, d.xr2_me, xri, gen++, GS + d.xr2_skip, 0, z0, z1, z2, sh8, shb8, err, bid, backoff, false)) return;
    } else if (!pbar8_lines(slots, gen++, GS, shb8, err, bid, backoff)) return;
    rs_g = rs_md; g_base = 2u * half_bytes;
  }
  bool first_tr = true;
  for (;;) {   // ---- trust-region iterations (exactly one pass when !FUSE)
    FSTAMP(0);
    if (FUSE && !first_tr) {
        // tCG.m:102-157 at the (possibly new) current point: eta = 0, r = mdelta = grad
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double2 g = Gs[r * PB + threadIdx.x];
            eta[r] = zz; rr[r] = g; md[r] = g; hmd[r] = zz; cmd[r] = zz; ctr[r] = zz;   // (cmd / ctr: dead across the tail, said so)
        }
        z_r = gg; d_Pd = gg; e_Pd = 0.0; e_Pe = 0.0; model_value = 0.0; alpha = 0.0; beta = 0.0;
        norm_r0 = sqrt(gg);
        j = 0; stop = 5; first = true; direct = false; have_x = false;
    }
    first_tr = false;
    // (FUSE at 16 lanes per row: ONE instance of the trip loop -- with three of them inside the loop over the TR iterations the
    // register allocation spills 400 bytes per lane into the trips, 103 000 Hess-vec/s on G81 at p = 32 against 170 000 with one.
    // Round 6: an instance with its local columns fixed for the whole KERNEL -- two columns of every row from LDS, three through the
    // buffer, one overflow column for the 1 to 4 rows of G81 that need it, the next gather requested inside the reduction -- was built
    // and measured: 12 420 ticks per trip against 12 365, i.e. nothing; a trip is its reduction chain (3.3 us) and 2.2 us of fp64
    // VALU work at 2 waves per SIMD, not its gather.  Removed.)
    // RELIANCE: `nl` is per WAVE, so the waves of one workgroup may sit in different instances of `trips` and meet
    // in the workgroup barriers of PSYNC8 (`__syncthreads()` / `s_barrier`) from different program counters.  gfx950's s_barrier counts
    // arrivals per workgroup regardless of the PC and every instance executes the SAME sequence of barriers per trip (PSYNC8 is the
    // only place with barriers; an instance leaves the loop on workgroup-uniform values only), so the counts pair up -- this is
    // hardware behaviour, not the HIP model's guarantee.  Making `nl` workgroup-uniform would send every workgroup of a grid in natural
    // order to NL = 2 (each holds the two ends of its chunk).  Covered on the device by the G81 tests: a workgroup there mixes NL = 3
    // waves (interior), NL = 2 (the two waves at the chunk's ends) and NL = 0 (the waves with the wrap-around columns of a grid row).
    if (MULTI && nl >= 3) trips(std::integral_constant<int, MULTI ? 3 : 0>());
    else if (MULTI && nl == 2) trips(std::integral_constant<int, MULTI ? 2 : 0>());
    else trips(std::integral_constant<int, 0>());
    if (failed) return;
    if (TRACE && FUSE && threadIdx.x == 0 && k_it - k_it0 < MSDP_TRACE_NJ)
        d.trace[((size_t)bx * MSDP_TRACE_NJ + (k_it - k_it0)) * 8 + 2] = (__builtin_readcyclecounter() & 0x00ffffffffffffffULL) | ((unsigned long long)(j & 0xff) << 56);
    if (!FUSE) {
        // ---- hand eta, Heta = r - grad and the final scalars to the RTR kernels (trustregions.m:540-550)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (OK(r)) {
                const int64_t o = (int64_t)ROW(r) * d.ld + 2 * sub;
                const double2 g = Gs[r * PB + threadIdx.x];
                st2(d.eta[0] + o, eta[r]);
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
    // Round 6 (profiles/r6_fused_timeline_p32_before.md: 14 us per iteration outside its trips).  What changed: the step stays in its
    // registers (it went through LDS and a rolled loop); the barrier in front of the gathers is a post on the reduction's slot lines
    // (the counter barrier: 3.3 us); the proposal's rows and its gradient's rows travel through the fine-grained exchange buffer like
    // the rows of a trip, not through d.Y / d.Gr; the neighbours' rows of the proposal's gradient -- the first gather of the next tCG if
    // the step is accepted, as it mostly is -- are requested inside the iteration's reduction; an accepted step swaps the roles of the
    // two sets of LDS buffers instead of copying one into the other; the point goes back to global memory once, when the launch ends.
    // x_prop = retr(x, eta) (ManiSDP_onlyunitdiag.m:142-145), <eta, grad + .5*Heta> (trustregions.m:549-550)
    double tv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};          // [0] cost, [1] |grad|^2 at the proposal, [2] <eta, grad + .5 Heta>
    const unsigned yx_base = 4u * half_bytes, gx_base = (5u + (unsigned)(cur ^ 1)) * half_bytes;   // the proposal's rows / its gradient's rows
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double2 y = Ys[r * PB + threadIdx.x], g = Gs[r * PB + threadIdx.x];
        const double2 e0 = eta[r];
        const double2 he = make_double2(rr[r].x - g.x, rr[r].y - g.y);
        tv[2] += e0.x * (g.x + 0.5 * he.x) + e0.y * (g.y + 0.5 * he.y);
        const double2 x = make_double2(y.x + e0.x, y.y + e0.y);
        double nn = sqrt(msdp_group_sum<LPR>(x.x * x.x + x.y * x.y));
        if (!(nn > 0.0)) nn = 1.0;                                  // empty row slot
        const double2 ypr = OK(r) ? make_double2(x.x / nn, x.y / nn) : zz;
        YPs[r * PB + threadIdx.x] = ypr;
        if (OK(r)) st2_sc1(rs_md, yx_base + ((unsigned)ROW(r) * gld + 2 * sub) * 8u, ypr);
    }
    FSTAMP(3);
    // (my proposal rows are performed before the post: the wait is inside)
    if (!pbar8_lines(slots, gen++, GS, shb8, err, bid, backoff)) return;
    FSTAMP(4);
    // cost and gradient at the proposal (ManiSDP_onlyunitdiag.m:117-125): YC = Y*C, eG = sum(YC.*Y), G = YC - Y.*eG
    auto prop_finish = [&](int r, double2 acc) {
        if (!colok) acc = zz;
        const double2 ypr = YPs[r * PB + threadIdx.x];
        const double dot = msdp_group_sum<LPR>(acc.x * ypr.x + acc.y * ypr.y);
        const double2 gpr = OK(r) ? make_double2(acc.x - ypr.x * dot, acc.y - ypr.y * dot) : zz;
        GPs[r * PB + threadIdx.x] = gpr;
        tv[1] += gpr.x * gpr.x + gpr.y * gpr.y;
        if (sub == 0 && epi == 0) {
            EGPs[SLOT(r)] = ROK(r) ? dot : 0.0;
            if (ROK(r)) tv[0] += 0.5 * dot;
        }
        if (OK(r)) st2_sc1(rs_md, gx_base + ((unsigned)ROW(r) * gld + 2 * sub) * 8u, gpr);
    };
    if (CSR) {
#pragma unroll
        for (int r = 0; r < R; ++r) prop_finish(r, csr_gather(r, rs_md, yx_base));
    } else if (R * EW <= 16) {                                     // all gathers in flight (wider rows: a row slot at a time, registers)
        double2 XG[R][EW > 0 ? EW : 1];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int w = 0; w < EW; ++w) XG[r][w] = ld2_sc1(rs_md, yx_base + ((unsigned)cs[w * ROWS + SLOT(r)] * gld + gcol) * 8u);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double2 acc = zz;
#pragma unroll
            for (int w = 0; w < EW; ++w) { const double vv = vs[w * ROWS + SLOT(r)]; acc.x = fma(vv, XG[r][w].x, acc.x); acc.y = fma(vv, XG[r][w].y, acc.y); }
            prop_finish(r, acc);
        }
    } else {
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            double2 xw[EW > 0 ? EW : 1];
#pragma unroll
            for (int w = 0; w < EW; ++w) xw[w] = ld2_sc1(rs_md, yx_base + ((unsigned)cs[w * ROWS + SLOT(r)] * gld + gcol) * 8u);
            double2 acc = zz;
#pragma unroll
            for (int w = 0; w < EW; ++w) { const double vv = vs[w * ROWS + SLOT(r)]; acc.x = fma(vv, xw[w].x, acc.x); acc.y = fma(vv, xw[w].y, acc.y); }
            prop_finish(r, acc);
        }
    }
    FSTAMP(5);
    // (the proposal's gradient rows are in place before the post: psync8 waits for the stores.  Requesting the neighbours' rows of that
    // gradient inside this reduction -- the first gather of the next tCG when the step is accepted -- was built and measured: the R x EW
    // row registers do not survive the decision block, the compiler parks them in scratch and WAITS for the loads to do so: the
    // reduction went from 6 300 to 15 500 ticks.  The first trip gathers them itself.)
    if (!PSYNC8(slots, gen++, GS, tv, sh8, shb8, err, bid, backoff, nullptr, []() {})) return;
    FSTAMP(6);
    {   // trustregions.m:548-729, identical in every workgroup (same bits in, same decision out)
        const double fp = tv[0], ggp = tv[1], prd = tv[2];
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
        if (accept) ++n_acc; else ++n_rej;
        n_hv += j;
        ++n_ce;
        if (lead) {
            Ctl* cw = d.ctl;
            cw->rho = rho; cw->rhonum = rhonum; cw->rhoden = rhoden; cw->fx_prop = fp; cw->gg_prop = ggp;
            cw->accepted = n_acc; cw->rejected = n_rej; cw->hessvecs = n_hv; cw->cost_evals = n_ce;
            cw->last_stop_inner = stop;
        }
        if (accept) {
            // the proposal becomes the point: the two sets of LDS buffers change roles, and its gradient's rows are where the
            // neighbours (and later tCGs at this point) find them -- slot cur of the exchange buffer
            cur ^= 1;
            fx = fp; gg = ggp;
            { double2* t = Ys; Ys = YPs; YPs = t; }
            { double2* t = Gs; Gs = GPs; GPs = t; }
            { double* t = eGs; eGs = EGPs; EGPs = t; }
            rs_g = rs_md; g_base = gx_base;
        }
        FSTAMP(7);
        ++k_it;                                                              // :729
    }
    if (sqrt(gg) < tolgradnorm || k_it >= maxiter) break;                    // stoppingcriterion.m:51-72
  }
    // the point the solve ends at goes back to global memory (its rows lived in LDS and in the exchange buffer since the launch began)
    {
        double* Yo = cur ? d.Y[1] : d.Y[0];
        double* Go = cur ? d.Gr[1] : d.Gr[0];
        double* eo = cur ? d.eG[1] : d.eG[0];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (OK(r)) {
                const int64_t o = (int64_t)ROW(r) * d.ld + 2 * sub;
                st2(Yo + o, Ys[r * PB + threadIdx.x]);
                st2(Go + o, Gs[r * PB + threadIdx.x]);
            }
            if (sub == 0 && ROK(r)) eo[ROW(r)] = eGs[SLOT(r)];
        }
    }
    if (lead) {
        Ctl* cw = d.ctl;
        cw->fx = fx; cw->gg = gg; cw->norm_grad = sqrt(gg); cw->Delta = Delta;
        cw->k = k_it; cw->cur = cur;
        cw->done = 1;
        cw->tcg_running = 0;
        frame_store(&d.F[0], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0, 0, stop, 0, 0, 0);
    }
#undef FSTAMP
#undef SLOT
#undef ROW
#undef ROK
#undef OK
}

template <int LPR, int EW, int R, bool TRACE = false, bool FUSE = false, int XRM = 0, int EP = 1>
__global__ __launch_bounds__(PB) void k_tcg_pipe_obl(Dev d, unsigned long long* slots, int* err) {
    tcg_pipe_body<LPR, EW, R, TRACE, FUSE, XRM, EP>(d, slots, err, (int)blockIdx.x);
}
