// msdp_device.h -- device-side helpers shared by the kernel translation units.
#pragma once
#include "msdp_common.h"

// Workgroup b -> row chunk.  Workgroups b and b+8 share an XCD (round-robin
// dispatch, observed; a pure speed heuristic), so XCD x gets the contiguous
// chunk range [x*G/8, (x+1)*G/8): its L2 then serves one contiguous 1/8 of the
// rows of every vector.  G is a multiple of 8.
__device__ __forceinline__ void msdp_chunk_rows(int n_loc, int G, int& lo, int& hi) {
    const int b = blockIdx.x;
    const int c = (b & 7) * (G >> 3) + (b >> 3);
    lo = (int)(((int64_t)n_loc * c) / G);
    hi = (int)(((int64_t)n_loc * (c + 1)) / G);
}

__device__ __forceinline__ double msdp_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LPR>
__device__ __forceinline__ double msdp_group_sum(double v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPR);
    return v;
}

// Deterministic workgroup sum; result valid in every thread.
__device__ __forceinline__ double msdp_block_sum(double v, double* sh /* >= 8 doubles */) {
    v = msdp_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) s += sh[i];
    return s;
}

// Write this workgroup's partial sum.
__device__ __forceinline__ void msdp_put_partial(double* P, int which, double v, double* sh) {
    const double s = msdp_block_sum(v, sh);
    if (threadIdx.x == 0) P[which * MSDP_MAX_GRID + blockIdx.x] = s;
}

// Every workgroup re-reduces the <= 512 partials of the previous launch in the
// same fixed order, so all workgroups (and all ranks) take identical decisions
// without atomics or fences.
__device__ __forceinline__ double msdp_sum_partials(const double* P, int which, int G, double* sh) {
    const double* a = P + which * MSDP_MAX_GRID;
    double v = 0.0;
    for (int i = threadIdx.x; i < G; i += blockDim.x) v += a[i];
    return msdp_block_sum(v, sh);
}

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double2 v) { *reinterpret_cast<double2*>(p) = v; }
