// accuracy probe: v_rcp_f64 alone, v_rcp_f64 + 1 Newton, f32-seed + 1 Newton, over s in [1, 1e12]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* s, double* a, double* b, double* c, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  double x = s[i];
  double r0 = __builtin_amdgcn_rcp(x);
  a[i] = r0;
  double e = __builtin_fma(-x, r0, 1.0); b[i] = __builtin_fma(r0, e, r0);
  double q0 = (double)__builtin_amdgcn_rcpf((float)x);
  double e2 = __builtin_fma(-x, q0, 1.0); c[i] = __builtin_fma(q0, e2, q0);
}
int main(){
  const int n = 1<<20; std::vector<double> s(n), a(n), b(n), c(n);
  for(int i=0;i<n;++i){ double u = (double)rand()/RAND_MAX; s[i] = exp(u*27.6)*(1.0 + 1e-3*((double)rand()/RAND_MAX)); }
  double *ds,*da,*db,*dc; hipMalloc(&ds,n*8); hipMalloc(&da,n*8); hipMalloc(&db,n*8); hipMalloc(&dc,n*8);
  hipMemcpy(ds,s.data(),n*8,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k,dim3(n/256),dim3(256),0,0,ds,da,db,dc,n);
  hipMemcpy(a.data(),da,n*8,hipMemcpyDeviceToHost); hipMemcpy(b.data(),db,n*8,hipMemcpyDeviceToHost); hipMemcpy(c.data(),dc,n*8,hipMemcpyDeviceToHost);
  double ma=0,mb=0,mc=0;
  for(int i=0;i<n;++i){ long double t = 1.0L/(long double)s[i];
    ma = fmax(ma, (double)fabsl((a[i]-t)/t)); mb = fmax(mb,(double)fabsl((b[i]-t)/t)); mc = fmax(mc,(double)fabsl((c[i]-t)/t)); }
  printf("max rel err: v_rcp_f64 %.3g | v_rcp_f64+1NR %.3g | rcp_f32 seed+1NR %.3g\n", ma, mb, mc);
  return 0; }
