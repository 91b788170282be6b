// Dependent-issue latency of the fp64 / fp32 / integer VALU instructions the step kernel is made of:
// one wave per SIMD, chains of CH independent dependency chains; cycles per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o valu_latency tools/micro/valu_latency.hip && ./valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ double fast_sqrt(double x) {   // rsq + two coupled Newton steps
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? x : g;
}

__device__ __forceinline__ double fast_div(double a, double b) {   // rcp + two Newton steps + residual correction
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

template <int CH, int OP>
__global__ void chain(double *out, long long *cycles, int n, double a, double b) {
  double x[CH];
  float xf[CH];
  uint32_t xi[CH];
  for (int c = 0; c < CH; c++) {
    x[c] = 1.0 + 0.001 * (threadIdx.x + c);
    xf[c] = (float) x[c];
    xi[c] = threadIdx.x + c + 1;
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i += 32) {
#pragma unroll
    for (int u = 0; u < 32; u++)      // 32 steps of every chain per loop trip: branch cost amortised
#pragma unroll
    for (int c = 0; c < CH; c++) {
      if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
      if (OP == 1) x[c] = x[c] * a;
      if (OP == 2) x[c] = x[c] + b;
      if (OP == 3) xf[c] = __builtin_fmaf(xf[c], (float) a, (float) b);
      if (OP == 4) xi[c] = xi[c] * 2654435761u + 12345u;                 // v_mul_lo_u32 + add
      if (OP == 5) x[c] = (double) (float) x[c] + b;                      // cvt f64->f32->f64 + add
      if (OP == 6) x[c] = 1.0 / x[c] + b;                                 // IEEE division sequence
      if (OP == 7) x[c] = __builtin_sqrt(x[c]) + b;
      if (OP == 8) x[c] = log(x[c]) + 2.0;                                // ocml log
      if (OP == 9) x[c] = exp(-x[c]) + 1.0;                               // ocml exp
      if (OP == 10) x[c] = cos(x[c]) + 1.0;                               // ocml cos
      if (OP == 11) x[c] = fast_sqrt(x[c]) + b;
      if (OP == 12) x[c] = fast_div(1.0, x[c]) + b;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < CH; c++)
    s += x[c] + xf[c] + xi[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0)
    cycles[blockIdx.x] = t1 - t0;
}

template <int CH, int OP>
void run(const char *name, int waves_per_block) {
  double *out;
  long long *cyc;
  hipMalloc(&out, 1024 * sizeof(double));
  hipMalloc(&cyc, 8 * sizeof(long long));
  const int n = 4096;
  hipLaunchKernelGGL((chain<CH, OP>), dim3(1), dim3(64 * waves_per_block), 0, 0, out, cyc, n, 0.999999, 1e-7);
  hipLaunchKernelGGL((chain<CH, OP>), dim3(1), dim3(64 * waves_per_block), 0, 0, out, cyc, n, 0.999999, 1e-7);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s chains %d waves/block %d: %7.2f clock ticks per op per wave (%.2f per op issued on the CU)\n", name, CH,
         waves_per_block, (double) h / ((double) n * CH), (double) h / ((double) n * CH * waves_per_block));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  int wall = 0;
  hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("shader clock %d kHz, wall clock %d kHz (s_memtime / readcyclecounter ticks)\n", clk, wall);
  run<1, 0>("fma_f64", 1);
  run<2, 0>("fma_f64", 1);
  run<4, 0>("fma_f64", 1);
  run<8, 0>("fma_f64", 1);
  run<1, 0>("fma_f64", 4);
  run<1, 0>("fma_f64", 8);
  run<4, 0>("fma_f64", 8);
  run<1, 1>("mul_f64", 1);
  run<4, 1>("mul_f64", 1);
  run<1, 2>("add_f64", 1);
  run<4, 2>("add_f64", 1);
  run<1, 3>("fma_f32", 1);
  run<4, 3>("fma_f32", 1);
  run<1, 4>("mul_lo_u32+add", 1);
  run<4, 4>("mul_lo_u32+add", 1);
  run<1, 5>("cvt+cvt+add", 1);
  run<4, 5>("cvt+cvt+add", 1);
  run<1, 6>("div_f64+add", 1);
  run<4, 6>("div_f64+add", 1);
  run<1, 7>("sqrt_f64+add", 1);
  run<4, 7>("sqrt_f64+add", 1);
  run<1, 8>("log+add", 1);
  run<4, 8>("log+add", 1);
  run<1, 9>("exp+add", 1);
  run<4, 9>("exp+add", 1);
  run<1, 10>("cos+add", 1);
  run<4, 10>("cos+add", 1);
  run<1, 11>("fast_sqrt+add", 1);
  run<4, 11>("fast_sqrt+add", 1);
  run<1, 12>("fast_div+add", 1);
  run<4, 12>("fast_div+add", 1);
  return 0;
}
