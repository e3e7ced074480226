// Microbenchmark: what do small (5 MB-per-vector) streaming kernels actually cost on MI355X?
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_stream.hip -o tools/microbench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__global__ void k_empty(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p == (int*)1) *p = 0; }

template <int NR, int NW>
__global__ void k_stream(double2* const* __restrict__ bufs, size_t n2) {
    double2* const* b = bufs;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < NR; ++r) { double2 v = b[r][i]; acc.x += v.x; acc.y += v.y; }
#pragma unroll
        for (int w = 0; w < NW; ++w) b[NR + w][i] = make_double2(acc.x + w, acc.y);
    }
}

template <int NR, int NW>
float run(double2** dbufs, size_t n2, int grid, int block, int reps, hipStream_t s) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    // capture reps launches in a graph so the host is out of the loop
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<NR, NW>), dim3(grid), dim3(block), 0, s, dbufs, n2);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / reps;
}

int main() {
    const size_t n = 20000, p = 32, n2 = n * p / 2;
    const int NB = 10;
    std::vector<double2*> h(NB);
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&h[i], n2 * sizeof(double2))); CK(hipMemset(h[i], 0, n2 * sizeof(double2))); }
    double2** d; CK(hipMalloc(&d, NB * sizeof(double2*)));
    CK(hipMemcpy(d, h.data(), NB * sizeof(double2*), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    // empty kernel cost in a graph
    {
        hipGraph_t g; hipGraphExec_t ge; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(320), dim3(1024), 0, s, (int*)nullptr);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("empty kernel 320x1024 in graph: %.2f us per launch\n", ms * 1e3 / 200);
    }
    const double MB = n2 * 16 / 1e6;
    int grids[] = {320, 640, 1280, 2560};
    int blocks[] = {1024, 512, 256, 256};
    for (int c = 0; c < 4; ++c) {
        float t11 = run<1, 1>(d, n2, grids[c], blocks[c], 200, s);
        float t31 = run<3, 1>(d, n2, grids[c], blocks[c], 200, s);
        float t63 = run<6, 3>(d, n2, grids[c], blocks[c], 200, s);
        float t10 = run<1, 0>(d, n2, grids[c], blocks[c], 200, s);
        printf("grid %4d x %4d: r1w1 %.2f us (%.0f GB/s)  r3w1 %.2f us (%.0f GB/s)  r6w3 %.2f us (%.0f GB/s)  r1w0 %.2f us\n",
               grids[c], blocks[c], t11, 2 * MB / t11 * 1e3, t31, 4 * MB / t31 * 1e3, t63, 9 * MB / t63 * 1e3, t10);
    }
    return 0;
}
