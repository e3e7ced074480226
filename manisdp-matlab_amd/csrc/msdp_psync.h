// msdp_psync.h -- grid-wide synchronisation / reduction primitives of the persistent kernels
// (msdp_persist.hip: tCG and the TR-iteration tail).  See msdp_persist.hip for the design notes.
#pragma once
#include "msdp_device.h"

#define PSYNC_NV 8                       // value arrays per generation (round 5: eight, for the one-reduction trip of msdp_pipe.h)
#define PSYNC_GEN 3
// Every workgroup posts its partials into PSYNC_REP replicas and polls replica (blockIdx & 7), i.e. the one of
// its XCD under round-robin dispatch: 32 pollers per cache line instead of 256 (tools/microbench_sync.hip:
// 3.03 -> 2.09 us per grid reduction at G = 256).  The barrier that carries no value uses 8 counters the same way.
#ifndef PSYNC_REP
#define PSYNC_REP 8
#endif
#define PSYNC_CNT_OFF ((size_t)PSYNC_GEN * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID)   // counters behind the slots, 64 B apart
// One region = the slots of psync (3 generations x 8 replicas x PSYNC_NV values x MSDP_MAX_GRID) + 8 barrier counters.
#define PSYNC_REGION (PSYNC_CNT_OFF + 64)
#define PSYNC_SENT 0xFFF8DEADBEEF0001ULL  // NaN payload no arithmetic produces
#define PSYNC_SPIN_LIMIT (1 << 22)
// 512 threads per workgroup, one workgroup per CU: 2 waves per SIMD, i.e. a 256-register budget per lane
// for the resident rows (1024-thread workgroups leave 128 and spill)
#define PB 512
#define PWAVES (PB / 64)

// Rows of the new direction are exchanged between workgroups on different XCDs (one L2 each) inside the launch.
// Agent-scope release/acquire fences (buffer_wbl2 / buffer_inv sc1 by every wave) were measured at 37 us per
// trip; instead the exchanged rows are written and gathered with sc1 (agent-coherent) buffer accesses, which
// the other L2s never hold stale, and ordered by the grid reduction that follows the stores.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define MSDP_CPOL_SC1 16
__device__ __forceinline__ double2 ld2_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, MSDP_CPOL_SC1);
    double2 o;
    o.x = __longlong_as_double(((long long)v.y << 32) | (long long)v.x);
    o.y = __longlong_as_double(((long long)v.w << 32) | (long long)v.z);
    return o;
}
// the same gather with the cache policy as a template argument: MSDP_CPOL_SC1 (agent scope) on one device, 17 = sc0 | sc1 (system scope)
// where the row may have been stored by another device (the two-level cross-rank instances)
template <int CPOL>
__device__ __forceinline__ double2 ld2_cp(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, CPOL);
    double2 o;
    o.x = __longlong_as_double(((long long)v.y << 32) | (long long)v.x);
    o.y = __longlong_as_double(((long long)v.w << 32) | (long long)v.z);
    return o;
}
__device__ __forceinline__ void st2_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, double2 d2) {
    const long long a = __double_as_longlong(d2.x), b = __double_as_longlong(d2.y);
    v4u v;
    v.x = (unsigned)(a & 0xffffffffLL); v.y = (unsigned)((unsigned long long)a >> 32);
    v.z = (unsigned)(b & 0xffffffffLL); v.w = (unsigned)((unsigned long long)b >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, MSDP_CPOL_SC1);
}


// Reduce (a, b, c) over the whole grid; nv = number of meaningful values (1 or 3).  Returns false when the
// spin bound was hit (error flag set; the caller leaves the kernel).
// bid: this workgroup's index among the G that synchronise -- blockIdx.x, or rank * (workgroups per rank) + blockIdx.x when the
// launches of several ranks share one slot region (cross-rank persistent tCG, msdp_persist.hip XR)
// Round 4: with three values, three WAVES poll side by side, one value array each (wave 0 posts all of them): the s_memtime
// trace of the trip (profiles/r4_persist_timeline_p32.md) showed the three-value reduction at 2.75 us against 1.94 us for the
// one-value one -- a poll of twelve loads per lane by one wave against four.  shb needs 8 doubles.
// (a', b') = v_permlane<W>_swap(a, b): a' = a in the even rows of W lanes and b's even-row copy in the odd ones, b' = a's odd-row copy in
// the even rows and b in the odd ones; a' + b' = a summed over the row pair (even rows) / b summed over the row pair (odd rows)
template <int W>
__device__ __forceinline__ double msdp_swap_add(double a, double b) {
    const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const unsigned alo = (unsigned)(ba & 0xffffffffLL), ahi = (unsigned)((unsigned long long)ba >> 32);
    const unsigned blo = (unsigned)(bb & 0xffffffffLL), bhi = (unsigned)((unsigned long long)bb >> 32);
    unsigned l0, l1, h0, h1;
    if (W == 16) {
        const auto rl = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    } else {
        const auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        l0 = rl[0]; l1 = rl[1]; h0 = rh[0]; h1 = rh[1];
    }
    return __longlong_as_double((long long)(((unsigned long long)h0 << 32) | l0)) + __longlong_as_double((long long)(((unsigned long long)h1 << 32) | l1));
}
// Three per-lane partials summed over the wave TOGETHER (round 5): two permlane swaps leave value k in row k of 16 lanes (row 3: zero),
// one row reduction finishes all of them -- a third of the instructions of three separate wave sums.  sh[k * PWAVES + wave] = value k.
// ONE helper for psync(nv = 3) and psync_post3: the split reduction promises the bits of the unsplit one.
__device__ __forceinline__ void psync_wave3(double a, double b, double c, double* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double x = msdp_swap_add<16>(msdp_swap_add<32>(a, c), msdp_swap_add<32>(b, 0.0));
    x += msdp_dpp<MSDP_DPP_XOR1>(x); x += msdp_dpp<MSDP_DPP_XOR2>(x);
    x += msdp_dpp<MSDP_DPP_HALF_MIRROR>(x); x += msdp_dpp<MSDP_DPP_MIRROR>(x);
    if ((lane & 15) == 0 && lane < 48) sh[(lane >> 4) * PWAVES + w] = x;
}
// drain (round 5): the caller has row stores in flight that must be performed before the workgroup posts -- the wait sits here, behind
// the wave sums, instead of in front of the call (the stores drain while the sums are formed)
__device__ __forceinline__ bool psync(unsigned long long* slots, unsigned gen, int G, int nv, double& a, double& b,
                                      double& c, double* sh, double* shb, int* err, int bid_in = -1, int backoff = 0, bool drain = false) {
    const int bid = bid_in < 0 ? (int)blockIdx.x : bid_in;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (nv > 1) psync_wave3(a, b, c, sh);
    else {
        a = msdp_wave_sum(a);
        if (lane == 0) { sh[w] = a; sh[PWAVES + w] = b; sh[2 * PWAVES + w] = c; }
    }
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (w < nv) {                                                   // polling wave w takes value array w
        unsigned long long* gbase = slots + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;
        if (w == 0 && lane < PSYNC_REP * PSYNC_NV) {
            const int rep = lane / PSYNC_NV, vi = lane % PSYNC_NV;
            if (vi < nv) {
                double s = 0.0;
                for (int i = 0; i < PWAVES; ++i) s += sh[vi * PWAVES + i];
                // the reset store of this slot's other generations (issued one sync ago) must have been performed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(gbase + ((size_t)rep * PSYNC_NV + vi) * MSDP_MAX_GRID + bid,
                                   (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const unsigned long long* p0 = gbase + ((size_t)(bid & (PSYNC_REP - 1)) * PSYNC_NV + w) * MSDP_MAX_GRID + lane;
        double r0;
        int spins = 0;
        bool fail = false;
        // A/B (option psync_backoff): nothing can be visible for the first half microsecond after the posts, and every poll of the 216
        // workgroups is traffic the posts compete with -- sleep before the first poll and after a failed one
        const int first = (nv > 1 && ((backoff >> 16) & 0xff)) ? ((backoff >> 16) & 0xff) : (backoff & 0xff);     // bits 16..23: the three-value reductions' own figure
        for (int q = 0; q < first; ++q) __builtin_amdgcn_s_sleep(1);
        for (;;) {
            // all slot loads of one poll are issued back to back with ONE wait (the compiler puts a full
            // s_waitcnt after every atomic load: serialized round trips, measured 20 us per sync)
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
                if (lane + 64 * q < G) {                                   // slots >= G hold the sentinel for ever
                    ok = ok && b0[q] != PSYNC_SENT;
                    t0 += __longlong_as_double((long long)b0[q]);         // same order as msdp_sum_partials
                }
            }
            r0 = t0;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT ||
                ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
            for (int q = 0; q < ((backoff >> 8) & 0xff); ++q) __builtin_amdgcn_s_sleep(1);
        }
        r0 = msdp_wave_sum(r0);
        if (lane == 0) {
            shb[w] = r0; shb[4 + w] = fail ? 1.0 : 0.0;
            if (nv == 1) { shb[1] = 0.0; shb[2] = 0.0; shb[5] = 0.0; shb[6] = 0.0; }
            if (fail) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // everybody has finished reading the previous generation (they all posted this one): reset my slots of it
        if (w == 0 && lane < PSYNC_REP * PSYNC_NV)
            __hip_atomic_store(slots + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID +
                                   (size_t)lane * MSDP_MAX_GRID + bid,
                               PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // wave-uniform results: readlane moves them to scalar registers (they live across the whole solve)
    a = msdp_readlane(shb[0], 0); b = msdp_readlane(shb[1], 0); c = msdp_readlane(shb[2], 0);
    return msdp_readlane(shb[4], 0) + msdp_readlane(shb[5], 0) + msdp_readlane(shb[6], 0) == 0.0;
}

// ---- Split form of the three-value reduction (round 5, msdp_persist.hip EARLY): the posts go out first, the caller does other work
// (waits for its neighbours' row flags, gathers their rows), polls the slots with the SAME loads and the same summation order as
// psync() wherever it has a wait anyway, and ends with psync_finish3.  Same bits as psync(..., nv = 3, ...).
__device__ __forceinline__ void psync_post3(unsigned long long* slots, unsigned gen, double a, double b, double c, double* sh, int bid) {
    psync_wave3(a, b, c, sh);                                      // (the wave sums of psync(nv = 3): same instructions, same bits)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (w == 0 && lane < PSYNC_REP * PSYNC_NV) {
        unsigned long long* gbase = slots + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;
        const int rep = lane / PSYNC_NV, vi = lane % PSYNC_NV;
        if (vi < 3) {
            double s = 0.0;
            for (int i = 0; i < PWAVES; ++i) s += sh[vi * PWAVES + i];
            // (no wait for the reset store of this generation's slots here: it was issued two synchronisations ago and the psync() of
            // reduction 1 in between waited for everything in front of its own post -- a wait at this point would put the drain
            // of the row stores the caller has just issued in front of the post)
            __hip_atomic_store(gbase + ((size_t)rep * PSYNC_NV + vi) * MSDP_MAX_GRID + bid,
                               (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// Where wave w (< 3) polls value array w of generation gen
__device__ __forceinline__ const unsigned long long* psync_poll_base(const unsigned long long* slots, unsigned gen, int bid) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return slots + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID +
           ((size_t)(bid & (PSYNC_REP - 1)) * PSYNC_NV + (w < 3 ? w : 0)) * MSDP_MAX_GRID + lane;
}
// One poll (four slot loads, ONE wait -- which also covers whatever loads the caller issued in front of this call): true when all G
// slots are filled; t0 = this lane's share of the sum (the order of psync / msdp_sum_partials).
__device__ __forceinline__ bool psync_poll_once(const unsigned long long* p0, int G, double& t0) {
    const int lane = threadIdx.x & 63;
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
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (lane + 64 * q < G) {
            ok = ok && b0[q] != PSYNC_SENT;
            t += __longlong_as_double((long long)b0[q]);
        }
    }
    t0 = t;
    return __builtin_amdgcn_ballot_w64(!ok) == 0ULL;
}
// Closes the split reduction: the polling waves hand their sums over, wave 0 resets its slots of the previous generation.
// `fail` (any wave): a bounded spin ran out.  Returns false in every thread if any wave failed.
// shb[7] = the workgroup's failure flag (0.0 at kernel start; no extra static LDS: the dynamic array behind sh / shb must stay
// 16-byte aligned -- one more int had moved it to offset 264 and every ds_read_b128 of the resident rows went misaligned)
__device__ __forceinline__ bool psync_finish3(unsigned long long* slots, unsigned gen, double t0, bool fail, double& a, double& b, double& c,
                                              double* shb, int* err, int bid) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (fail && lane == 0) { shb[7] = 1.0; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (w < 3) {
        const double r0 = msdp_wave_sum(t0);
        if (lane == 0) shb[w] = r0;
        if (w == 0 && lane < PSYNC_REP * PSYNC_NV)
            __hip_atomic_store(slots + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID +
                                   (size_t)lane * MSDP_MAX_GRID + bid,
                               PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    a = msdp_readlane(shb[0], 0); b = msdp_readlane(shb[1], 0); c = msdp_readlane(shb[2], 0);
    return msdp_readlane(shb[7], 0) == 0.0;
}

// ---- Two-level reduction across MEMBERS that own a device each (round 6, the N-GPU form of the cross-rank kernels; N <= 8).
// The flat protocol above lets every workgroup of every member poll one slot region: fine on one device, but between devices every poll
// would cross xGMI and N x G workgroups would have to fit 256 slots.  Here a member reduces over ITS OWN grid in ITS OWN memory and only
// eight sums per member travel:
//   level 1   every workgroup posts its partials into the member's local slot array (one replica: only the leader polls it);
//             the member's LEADER workgroup (local index 0) polls the G local slots -- wave w takes value array w, the order of psync() --
//             and stores the member's sum of value w into line `me` of generation gen in EVERY member's block (its own included):
//             N stores of 8 bytes per value over the peer mappings, system scope;
//   level 2   wave 0 of EVERY workgroup of every member polls the N lines of its OWN block (local memory: one 8-byte load per lane,
//             lane = 8 member + value) until none holds the sentinel and adds the members' sums in one fixed order -- the same lanes,
//             the same instructions on every member: the same bits, the same decisions everywhere.
// Block of a member (fine-grained memory of its device, mapped by the others): XR2_REGIONS regions (tCG 0 / 1, TR tail 2 / 3, alternating
// with the TR iteration like the flat regions), each [the slots of psync (PSYNC_CNT_OFF + 64)] [PSYNC_GEN x 8 members x 8 values] [pad];
// behind them the error word.  Generations rotate as above: a workgroup puts its local slots of generation gen - 1 back to the sentinel
// when it has passed gen, the leader does the same for the member lines in its own block (at that point every workgroup of every
// member has posted gen, i.e. finished reading gen - 1; the next writes into those lines are the pushes of gen + 2, which a peer's leader
// issues only after it has passed gen + 1 -- behind this member's push of gen + 1, in front of which the reset store was waited for).
// nv = 0: a barrier (one dummy value).  peers[q] = member q's block as THIS process maps it (the caller keeps the N pointers in LDS).
#define XR2_LINES (PSYNC_GEN * 8 * 8)
#define XR2_REGION (PSYNC_REGION + XR2_LINES + 64)
#define XR2_REGIONS 4
#define XR2_ERR_OFF ((size_t)XR2_REGIONS * XR2_REGION)            // (u64 units) the member's error word
#define XR2_BLOCK_U64 (XR2_ERR_OFF + 64)
#define MSDP_CPOL_SYS 17                                          // sc0 | sc1: system scope
__device__ __forceinline__ unsigned long long ld_u64_sys(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void xr2_fail(unsigned long long* blk, unsigned long long* const* peers, int N, int* err) {   // (peers: the table in LDS)
    // this member's error word, and every peer's: their bounded spins look at it every 1024 polls and give up at once
    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = threadIdx.x & 63;
    if (lane < N) __hip_atomic_store(reinterpret_cast<int*>(peers[lane] + XR2_ERR_OFF), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool psync2(unsigned long long* blk, unsigned long long* const* peers, int N, int me, int ri, unsigned gen, int G, int nv,
                                       double& a, double& b, double& c, double* sh, double* shb, int* err, int bid, int backoff, bool drain = false) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nvp = nv > 0 ? nv : 1;
    if (nv > 1) psync_wave3(a, b, c, sh);
    else if (nv == 1) { a = msdp_wave_sum(a); if (lane == 0) sh[w] = a; }
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long* ls = blk + (size_t)ri * XR2_REGION;
    unsigned long long* ml = ls + PSYNC_REGION;
    unsigned long long* gbase = ls + (size_t)(gen % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID;   // replica 0
    if (w == 0 && lane < nvp) {
        double s = 0.0;
        if (nv > 0) for (int i = 0; i < PWAVES; ++i) s += sh[lane * PWAVES + i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the reset store of this slot's other generations has been performed
        __hip_atomic_store(gbase + (size_t)lane * MSDP_MAX_GRID + bid, (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool fail = false;
    if (bid == 0 && w < nvp) {
        // the member's leader: value array w of the local slots, summed in the order of psync(); the sum goes to every member's block
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
    }
    if (w == 0) {
        // everybody: the N member lines of this generation in my own block
        const unsigned long long* p = ml + (size_t)(gen % PSYNC_GEN) * 64 + lane;
        const bool need = (lane >> 3) < N && (lane & 7) < nvp;
        const int first = bid == 0 ? 0 : (backoff & 0xff);          // (the leader comes from its level-1 poll: nothing to sleep for)
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
        // value v = the sum over the members, lanes v, v + 8, ..., v + 56: one fixed butterfly, the same on every member
        double t = need ? __longlong_as_double((long long)x) : 0.0;
        t += msdp_dpp<0x128>(t);                                   // row_ror:8 = lane ^ 8 inside a row of 16
        t = msdp_rowpair_sum<16>(t);
        t = msdp_rowpair_sum<32>(t);
        if (lane < 3) shb[lane] = nv > lane ? t : 0.0;
        if (lane == 0) { shb[4] = fail ? 1.0 : 0.0; if (bid != 0 || nvp < 2) shb[5] = 0.0; if (bid != 0 || nvp < 3) shb[6] = 0.0; }
        if (fail) xr2_fail(blk, peers, N, err);
        // my local slots of the previous generation back to the sentinel; the leader: the member lines of it in this member's block
        if (lane < PSYNC_NV)
            __hip_atomic_store(ls + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * PSYNC_REP * PSYNC_NV * MSDP_MAX_GRID + (size_t)lane * MSDP_MAX_GRID + bid,
                               PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bid == 0)
            __hip_atomic_store(ml + (size_t)((gen + PSYNC_GEN - 1) % PSYNC_GEN) * 64 + lane, PSYNC_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (bid == 0 && w < nvp && lane == 0) shb[4 + w] = fail ? 1.0 : 0.0;     // (the leader's other polling waves)
    __syncthreads();
    a = msdp_readlane(shb[0], 0); b = msdp_readlane(shb[1], 0); c = msdp_readlane(shb[2], 0);
    return msdp_readlane(shb[4], 0) + msdp_readlane(shb[5], 0) + msdp_readlane(shb[6], 0) == 0.0;
}
// the member's OTHER region (the one the next launch of the alternating pair will use): local slots by all workgroups, the member lines by
// the leader -- performed before this launch ends, long before a peer's push into them can arrive (a peer reaches the launch that uses
// this region only through a reduction this member takes part in first)
__device__ __forceinline__ void psync2_reset_other(unsigned long long* blk, int ri_other, int bid, int G) {
    unsigned long long* ls = blk + (size_t)ri_other * XR2_REGION;
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < PSYNC_CNT_OFF; i += (size_t)G * blockDim.x) ls[i] = PSYNC_SENT;
    if (bid == 0 && threadIdx.x < XR2_LINES) ls[PSYNC_REGION + threadIdx.x] = PSYNC_SENT;
}

// Barrier without a value (the new direction rows are in place): workgroup b adds to counter b & 7, everybody
// polls the 8 counters (64 B apart).  nbar = number of barriers passed before this one.  G is a multiple of 8.
__device__ __forceinline__ bool pbarrier(unsigned long long* slots, unsigned nbar, int G, double* shb, int* err, int bid_in = -1) {
    const int bid = bid_in < 0 ? (int)blockIdx.x : bid_in;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long* cnt = slots + PSYNC_CNT_OFF;
        if (lane == 0) __hip_atomic_fetch_add(cnt + 8 * (bid & 7), 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long want = (unsigned long long)(nbar + 1) * (unsigned)(G / 8);
        int spins = 0;
        bool fail = false;
        for (;;) {
            unsigned long long v = want;
            if (lane < 8) v = __hip_atomic_load(cnt + 8 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__builtin_amdgcn_ballot_w64(v < want) == 0ULL) break;
            ++spins;
            if (spins > PSYNC_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { fail = true; break; }
        }
        if (lane == 0) {
            shb[3] = fail ? 1.0 : 0.0;
            if (fail) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    return shb[3] == 0.0;
}


// Two kernels alternate on a stream (persistent tCG, TR-iteration tail), each with its own region; a kernel may
// not reset its own region while its workgroups poll it, so each one resets the OTHER kernel's region at its start:
// the workgroups share the slots out among themselves, workgroup 0 clears the counters (completed at the kernel boundary).
__device__ __forceinline__ void psync_reset_other(unsigned long long* other, int bid_in = -1, int G_in = -1) {
    const int bid = bid_in < 0 ? (int)blockIdx.x : bid_in;
    const int G = G_in < 0 ? (int)gridDim.x : G_in;                // the workgroups that share the region (all of them call this)
    // (round 5: every slot of the region, whatever layout the other kernel gives it -- psync() arrays or the lines of psync8())
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < PSYNC_CNT_OFF; i += (size_t)G * blockDim.x) other[i] = PSYNC_SENT;
    if (bid == 0 && threadIdx.x >= 128 && threadIdx.x < 192) other[PSYNC_CNT_OFF + (threadIdx.x - 128)] = 0ULL;
}
