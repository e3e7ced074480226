// Microbenchmark 2: what in k_tcg_upd1 costs 20 us when a plain r6w3 stream costs 6.5 us?
#include "../manisdp-matlab_amd/csrc/msdp_kernels.hip"
void msdp_set_error(const char*, ...) {}
int msdp_dense_costgrad(msdp_handle, int) { return -1; }
int msdp_dense_hess(msdp_handle) { return -1; }
int msdp_affine_costgrad(msdp_handle, int) { return -1; }
int msdp_affine_hess(msdp_handle) { return -1; }
int msdp_sphere_upd2(msdp_handle) { return -1; }
int msdp_sphere_retract(msdp_handle) { return -1; }
int msdp_allreduce_partials(msdp_handle, int, int) { return 0; }
int msdp_allgather_rows(msdp_handle h, const double* l) { h->d.full = (double*)l; return 0; }
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
#define WAVES 16
struct Args { double* v[10]; double* P; Frame* F; int n_loc, ld, G; };

__device__ __forceinline__ double wave_sum(double v){
#pragma unroll
  for (int o=32;o>0;o>>=1) v += __shfl_xor(v,o,64); return v; }
__device__ __forceinline__ double block_sum(double v, double* sh){
  v = wave_sum(v); int w = threadIdx.x>>6; __syncthreads(); if((threadIdx.x&63)==0) sh[w]=v; __syncthreads();
  double s=0; for(int i=0;i<(int)(blockDim.x>>6);++i) s+=sh[i]; return s; }

// MODE bits: 1 = XCD chunk mapping, 2 = prologue (frame + partial re-reduction), 4 = epilogue partials, 8 = dyn index
template <int MODE>
__global__ __launch_bounds__(1024) void k_var(Args a) {
  __shared__ double sh[3*WAVES];
  int lo, hi;
  if (MODE & 1) { int b = blockIdx.x; int c = (b&7)*(a.G>>3) + (b>>3); lo = (int)((long)a.n_loc*c/a.G); hi = (int)((long)a.n_loc*(c+1)/a.G); }
  else { lo = (int)((long)a.n_loc*blockIdx.x/a.G); hi = (int)((long)a.n_loc*(blockIdx.x+1)/a.G); }
  double alpha = 0.5; int ix = 0;
  if (MODE & 2) {
    Frame f = a.F[0];
    if (!f.active) return;
    double v = 0; for (int i = threadIdx.x; i < a.G; i += blockDim.x) v += a.P[i];
    double dHd = block_sum(v, sh);
    alpha = f.z_r / (dHd + 3.0);
    ix = f.eta_idx;
  }
  const long e0 = (long)lo*a.ld, e1 = (long)hi*a.ld;
  const double* eta = (MODE & 8) ? a.v[ix] : a.v[0];
  const double* Heta = (MODE & 8) ? a.v[2+ix] : a.v[2];
  double* neta = (MODE & 8) ? a.v[ix^1] : a.v[1];
  double* nHeta = (MODE & 8) ? a.v[2+(ix^1)] : a.v[3];
  const double* md = a.v[4]; const double* Hmd = a.v[5]; double* r = a.v[6]; const double* g = a.v[7];
  double s1=0,s2=0,s3=0;
  for (long i = e0 + 2*threadIdx.x; i < e1; i += 2*1024) {
    double2 e = *(const double2*)(eta+i), he = *(const double2*)(Heta+i), m = *(const double2*)(md+i), hm = *(const double2*)(Hmd+i);
    double2 rr = *(const double2*)(r+i), gv = *(const double2*)(g+i);
    double2 ne = make_double2(e.x-alpha*m.x, e.y-alpha*m.y), nh = make_double2(he.x-alpha*hm.x, he.y-alpha*hm.y), nr = make_double2(rr.x-alpha*hm.x, rr.y-alpha*hm.y);
    *(double2*)(neta+i)=ne; *(double2*)(nHeta+i)=nh; *(double2*)(r+i)=nr;
    s1 += ne.x*gv.x+ne.y*gv.y; s2 += ne.x*nh.x+ne.y*nh.y; s3 += nr.x*nr.x+nr.y*nr.y;
  }
  if (MODE & 4) {
    double t1 = block_sum(s1, sh), t2 = block_sum(s2, sh), t3 = block_sum(s3, sh);
    if (threadIdx.x==0) { a.P[512+blockIdx.x]=t1; a.P[1024+blockIdx.x]=t2; a.P[1536+blockIdx.x]=t3; }
  } else if (s1+s2+s3 == 12345.678) a.P[0] = s1;
}


// templated copy of k_tcg_upd1 for bisection: bit0 = no exit branch, bit1 = no Frame struct copy (fields only),
// bit2 = no ctl reads, bit3 = int32 indexing
template <int T>
__global__ __launch_bounds__(MSDP_BLOCK) void k_copy(Dev d) {
    __shared__ double sh[3 * MSDP_WAVES];
    Frame f0;
    if (T & 2) { f0.active = d.F[0].active; f0.z_r = d.F[0].z_r; f0.e_Pe = d.F[0].e_Pe; f0.e_Pd = d.F[0].e_Pd; f0.d_Pd = d.F[0].d_Pd; f0.eta_idx = d.F[0].eta_idx; f0.j = d.F[0].j; }
    else f0 = d.F[0];
    if (!f0.active) {
        if (blockIdx.x == 0 && threadIdx.x == 0) d.F[1] = f0;
        return;
    }
    const Ctl* c = d.ctl;
    const bool bench = (T & 4) ? true : (c->bench_mode != 0);
    const double Delta = (T & 4) ? 10.0 : c->Delta;
    int lo, hi;
    msdp_chunk_rows(d.n_loc, d.G, lo, hi, 0);
    const int64_t e0 = (int64_t)lo * d.ld, e1 = (int64_t)hi * d.ld;
    const int ix = f0.eta_idx;
    const double* __restrict__ eta = ix ? d.eta[1] : d.eta[0];
    const double* __restrict__ Heta = ix ? d.Heta[1] : d.Heta[0];
    double* __restrict__ neta = ix ? d.eta[0] : d.eta[1];
    double* __restrict__ nHeta = ix ? d.Heta[0] : d.Heta[1];
    const double* __restrict__ g = (T & 4) ? d.Gr[0] : (c->cur ? d.Gr[1] : d.Gr[0]);
    const int64_t i0 = e0 + 2 * threadIdx.x;
    const double d_Hd = msdp_sum_partials(d.P, P_DHD, d.G);
    const double alpha = f0.z_r / (d_Hd + 3.0);
    const double e_Pe_new = f0.e_Pe + 2.0 * alpha * f0.e_Pd + alpha * alpha * f0.d_Pd;
    if (!(T & 1)) {
      if (!bench && (d_Hd <= 0.0 || e_Pe_new >= Delta * Delta)) {
        const double tau = (-f0.e_Pd + sqrt(f0.e_Pd * f0.e_Pd + f0.d_Pd * (Delta * Delta - f0.e_Pe))) / f0.d_Pd;
        for (int64_t i = i0; i < e1; i += 2 * MSDP_BLOCK) {
            const double2 e = ld2(eta + i), he = ld2(Heta + i), m = ld2(d.md + i), hm = ld2(d.Hmd + i);
            st2(neta + i, make_double2(e.x - tau * m.x, e.y - tau * m.y));
            st2(nHeta + i, make_double2(he.x - tau * hm.x, he.y - tau * hm.y));
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            Frame f = f0;
            f.active = 0; f.stop = (d_Hd <= 0.0) ? 1 : 2; f.eta_idx = ix ^ 1; f.j = f0.j + 1;
            d.F[1] = f;
        }
        return;
      }
    }
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t i = i0; i < e1; i += 2 * MSDP_BLOCK) {
        double2 e, he, m, hm, rr, gv;
        e = ld2(eta + i); he = ld2(Heta + i); m = ld2(d.md + i); hm = ld2(d.Hmd + i);
        rr = ld2(d.r + i); gv = ld2(g + i);
        const double2 ne = make_double2(e.x - alpha * m.x, e.y - alpha * m.y);
        const double2 nh = make_double2(he.x - alpha * hm.x, he.y - alpha * hm.y);
        const double2 nr = make_double2(rr.x - alpha * hm.x, rr.y - alpha * hm.y);
        st2(neta + i, ne); st2(nHeta + i, nh); st2(d.r + i, nr);
        s1 += ne.x * gv.x + ne.y * gv.y; s2 += ne.x * nh.x + ne.y * nh.y; s3 += nr.x * nr.x + nr.y * nr.y;
    }
    msdp_put_partials3(d.P, P_S1, s1, P_S2, s2, P_S3, s3, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (T & 2) { d.F[1].alpha = alpha; d.F[1].e_Pe = e_Pe_new; }
        else { Frame f = f0; f.alpha = alpha; f.e_Pe = e_Pe_new; d.F[1] = f; }
    }
}
template <int T> void run_copy(Dev d, hipStream_t s) {
    printf("k_copy<%d> x3:", T);
    for (int k = 0; k < 3; ++k) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      hipGraph_t g; hipGraphExec_t ge;
      (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      for (int i=0;i<400;++i) hipLaunchKernelGGL((k_copy<T>), dim3(d.G), dim3(1024), 0, s, d);
      (void)hipStreamEndCapture(s,&g); (void)hipGraphInstantiate(&ge,g,nullptr,nullptr,0);
      (void)hipGraphLaunch(ge,s); (void)hipStreamSynchronize(s);
      (void)hipEventRecord(e0,s); (void)hipGraphLaunch(ge,s); (void)hipEventRecord(e1,s); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms,e0,e1); printf(" %.2f", ms*1e3f/400);
    }
    printf("\n");
}

template <int MODE>
float run(Args a, int reps, hipStream_t s) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i=0;i<reps;++i) hipLaunchKernelGGL((k_var<MODE>), dim3(a.G), dim3(1024), 0, s, a);
  (void)hipStreamEndCapture(s,&g); (void)hipGraphInstantiate(&ge,g,nullptr,nullptr,0);
  (void)hipGraphLaunch(ge,s); (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0,s); (void)hipGraphLaunch(ge,s); (void)hipEventRecord(e1,s); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms,e0,e1); return ms*1e3f/reps;
}

int main(int argc, char** argv){
  Args a; a.n_loc=20000; a.ld=32; a.G=320;
  size_t cnt=(size_t)a.n_loc*a.ld;
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<double> init(cnt);
  for (size_t i = 0; i < cnt; ++i) init[i] = mode == 0 ? 0.0 : (mode == 1 ? 1.0 : 1e-3 * ((i * 2654435761u) % 1000) - 0.5);
  for(int i=0;i<10;++i){ CK(hipMalloc(&a.v[i],cnt*8)); CK(hipMemcpy(a.v[i],init.data(),cnt*8,hipMemcpyHostToDevice)); }
  CK(hipMalloc(&a.P, 4096*8)); CK(hipMemset(a.P,0,4096*8));
  CK(hipMalloc(&a.F, 2*sizeof(Frame)));
  Frame f{}; f.active=1; f.z_r=1.0; CK(hipMemcpy(a.F,&f,sizeof(f),hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  printf("plain          : %.2f us\n", run<0>(a,200,s));
  printf("+xcd chunks    : %.2f us\n", run<1>(a,200,s));
  printf("+prologue      : %.2f us\n", run<2>(a,200,s));
  printf("+epilogue      : %.2f us\n", run<4>(a,200,s));
  printf("+dyn index     : %.2f us\n", run<8>(a,200,s));
  printf("pro+epi        : %.2f us\n", run<6>(a,200,s));
  printf("all            : %.2f us\n", run<15>(a,200,s));
  printf("all x8 (sustained):"); for (int k=0;k<8;++k) printf(" %.2f", run<15>(a,400,s)); printf("\n");
  {
    Dev d; memset(&d, 0, sizeof(d));
    d.n = d.n_loc = a.n_loc; d.p = d.ld = a.ld; d.G = a.G; d.variant = argc > 2 ? atoi(argv[2]) : 0;
    d.eta[0] = a.v[0]; d.eta[1] = a.v[1]; d.Heta[0] = a.v[2]; d.Heta[1] = a.v[3]; d.md = a.v[4]; d.Hmd = a.v[5]; d.r = a.v[6]; d.Gr[0] = a.v[7]; d.Gr[1] = a.v[7];
    d.P = a.P; d.F = (Frame*)a.F;
    CK(hipMalloc(&d.ctl, sizeof(Ctl)));
    Ctl c; memset(&c, 0, sizeof(c)); c.bench_mode = 1; c.Delta = 10; c.maxinner = 1 << 30; c.mininner = 1; c.kappa = 0.1; c.theta = 1;
    CK(hipMemcpy(d.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
    run_copy<0>(d, s); run_copy<1>(d, s); run_copy<3>(d, s); run_copy<7>(d, s);
    printf("REAL k_tcg_upd1 on the same buffers x6:");
    for (int k = 0; k < 6; ++k) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      hipGraph_t g; hipGraphExec_t ge;
      (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      for (int i=0;i<400;++i) hipLaunchKernelGGL(k_tcg_upd1, dim3(d.G), dim3(1024), 0, s, d);
      (void)hipStreamEndCapture(s,&g); (void)hipGraphInstantiate(&ge,g,nullptr,nullptr,0);
      (void)hipGraphLaunch(ge,s); (void)hipStreamSynchronize(s);
      (void)hipEventRecord(e0,s); (void)hipGraphLaunch(ge,s); (void)hipEventRecord(e1,s); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms,e0,e1); printf(" %.2f", ms*1e3f/400);
    }
    printf("\n");
  }
  printf("plain x4:"); for (int k=0;k<4;++k) printf(" %.2f", run<0>(a,400,s)); printf("\n");
  return 0;
}
