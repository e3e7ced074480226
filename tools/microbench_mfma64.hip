// microbench_mfma64.hip -- what fp64 MFMA rate can gfx950 sustain?  (tools only; not part of the library)
//   variant 0: NACC independent accumulator chains of v_mfma_f64_16x16x4_f64, operands in registers
//   variant 1: the same plus one ds_read_b64 per MFMA (B operand from LDS, as k_dense_partial2 does)
//   variant 2: v_mfma_f64_4x4x4_4b
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_mfma64.hip -o tools/bin/microbench_mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int VAR>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    __shared__ double lds[64 * 36];
    for (int e = threadIdx.x; e < 64 * 36; e += 256) lds[e] = b0 + e * 1e-9;
    __syncthreads();
    double4_t acc[NACC];
    double s4[NACC];
    for (int t = 0; t < NACC; ++t) { acc[t] = (double4_t){0, 0, 0, 0}; s4[t] = 0; }
    const int lane = threadIdx.x & 63;
    double a = a0 + lane * 1e-9, b = b0;
    const double* bp = &lds[(lane >> 4) * 36 + (lane & 15)];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            double bv[NACC];
#pragma unroll
            for (int t = 0; t < NACC; ++t) bv[t] = (VAR == 1) ? bp[((u * 4) % 60) * 36 + 16 * (t & 1)] : b;
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if (VAR == 2) s4[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bv[t], s4[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[t], acc[t], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3] + s4[t];
    if (s == 12345.678) out[0] = s;
}

template <int NACC, int VAR>
static void run(const char* name, int wgs_per_cu, double* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int G = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<NACC, VAR>), dim3(G), dim3(256), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, VAR>), dim3(G), dim3(256), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = (VAR == 2) ? 2.0 * 4 * 4 * 4 * 4 : 2.0 * 16 * 16 * 4;
    const double flop = (double)G * 4 * iters * 8 * NACC * per;
    printf("%-34s NACC=%d waves/SIMD=%d : %7.3f ms  %6.2f TFLOP/s\n", name, NACC, wgs_per_cu, ms, flop / ms * 1e-9);
}

int main() {
    double* out; hipMalloc(&out, 8);
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 0>("mfma_f64_16x16x4 regs", w, out);
        run<2, 0>("mfma_f64_16x16x4 regs", w, out);
        run<4, 0>("mfma_f64_16x16x4 regs", w, out);
        run<2, 1>("mfma_f64_16x16x4 + ds_read_b64", w, out);
        run<4, 1>("mfma_f64_16x16x4 + ds_read_b64", w, out);
        run<4, 2>("mfma_f64_4x4x4_4b regs", w, out);
    }
    return 0;
}
