// Microbenchmark 3: streaming kernel whose INPUTS were just written by the previous kernel
// (the real tCG situation), vs inputs that stay clean (possibly L2-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int NR, int NW, int OFFW>
__global__ __launch_bounds__(1024) void k_stream(double2* const* __restrict__ bufs, size_t n2, double s) {
    double2* const* b = bufs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 acc = make_double2(s, s);
#pragma unroll
        for (int r = 0; r < NR; ++r) { double2 v = b[r][i]; acc.x += v.x; acc.y += v.y; }
#pragma unroll
        for (int w = 0; w < NW; ++w) b[OFFW + w][i] = make_double2(acc.x * 0.37 + w, acc.y * 0.41);
    }
}

template <typename F>
float timeit(F enqueue, int reps, hipStream_t s) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < reps; ++i) enqueue();
    (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    (void)hipEventRecord(a, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : 20000, p = 32, n2 = n * p / 2;
    const int NB = 12;
    std::vector<double2*> h(NB);
    const int rnd = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<double> init(n2 * 2);
    for (size_t i = 0; i < init.size(); ++i) init[i] = rnd ? 1e-3 * ((i * 2654435761u) % 1000) - 0.5 : 0.0;
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&h[i], n2 * sizeof(double2))); CK(hipMemcpy(h[i], init.data(), n2 * sizeof(double2), hipMemcpyHostToDevice)); }
    double2** d; CK(hipMalloc(&d, NB * sizeof(double2*)));
    CK(hipMemcpy(d, h.data(), NB * sizeof(double2*), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int grid = (int)((n2 + 1023) / 1024), blk = 1024;
    const double MB = n2 * 16 / 1e6;
    float tw6 = timeit([&] { hipLaunchKernelGGL((k_stream<0, 6, 0>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0); }, 100, s);
    float tr6w3 = timeit([&] { hipLaunchKernelGGL((k_stream<6, 3, 6>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0); }, 100, s);
    float tboth = timeit([&] {
        hipLaunchKernelGGL((k_stream<0, 6, 0>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0);
        hipLaunchKernelGGL((k_stream<6, 3, 6>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0); }, 100, s);
    float tr3w1 = timeit([&] { hipLaunchKernelGGL((k_stream<3, 1, 6>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0); }, 100, s);
    float tw3 = timeit([&] { hipLaunchKernelGGL((k_stream<0, 3, 0>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0); }, 100, s);
    float tboth31 = timeit([&] {
        hipLaunchKernelGGL((k_stream<0, 3, 0>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0);
        hipLaunchKernelGGL((k_stream<3, 1, 6>), dim3(grid), dim3(blk), 0, s, d, n2, 1.0); }, 100, s);
    printf("n=%zu vec=%.2f MB grid=%d random=%d\n", n, MB, grid, rnd);
    {   // sustained: replay a 100-launch r6w3 graph for ~0.3 s and report per-launch time every 50 ms
        hipGraph_t g; hipGraphExec_t ge; hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_stream<6, 3, 6>), dim3(grid), dim3(blk), 0, s, d, n2, 1e-9);
        (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("sustained r6w3 per-launch us:");
        for (int rep = 0; rep < 8; ++rep) {
            (void)hipEventRecord(a, s);
            for (int k = 0; k < 60; ++k) (void)hipGraphLaunch(ge, s);
            (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf(" %.2f", ms * 1e3 / 6000);
        }
        printf("\n");
    }
    printf("w6 alone %.2f us (%.0f GB/s) | r6w3 clean inputs %.2f us (%.0f GB/s) | w6 then r6w3: %.2f us => r6w3 on dirty inputs %.2f us (%.0f GB/s)\n",
           tw6, 6 * MB / tw6 * 1e3, tr6w3, 9 * MB / tr6w3 * 1e3, tboth, tboth - tw6, 9 * MB / (tboth - tw6) * 1e3);
    printf("w3 alone %.2f us | r3w1 clean %.2f us (%.0f GB/s) | w3 then r3w1: %.2f => r3w1 dirty %.2f us (%.0f GB/s)\n",
           tw3, tr3w1, 4 * MB / tr3w1 * 1e3, tboth31, tboth31 - tw3, 4 * MB / (tboth31 - tw3) * 1e3);
    return 0;
}
