// Cycles per wave of the random-number building blocks of the step kernel (one wave, dependent chain through
// the counter so that nothing is hoisted):  squares + uniform, Box-Muller pair, normal_triple.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -disable-machine-licm -I. -o rng_cost tools/micro/rng_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "mptrac_amd/csrc/mphip_device.hpp"
using namespace mphip;

template <int OP>
__global__ void k(double *out, long long *cycles, int n) {
  uint64_t ctr = threadIdx.x * 977u + 12345u;
  double acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
    if (OP == 0) { const double u = uniform01(ctr); acc += u; }
    if (OP == 1) { double a, b; normal_pair(ctr, 2 * (uint64_t) i, a, b); acc += a + b; }
    if (OP == 2) { double a, b, c; normal_triple(ctr, (uint64_t) i + threadIdx.x, a, b, c); acc += a + b + c; }
    if (OP == 3) { acc += log_unit(0.5 + 1e-3 * (double) (ctr & 255)); }
    if (OP == 4) { const float x = 6.2f * (float) (ctr & 1023) * (1.f / 1024.f); acc += libm_sincosf(x, 0) + libm_sincosf(x, 1); }
    if (OP == 5) { acc += cos_latitude(1.5 * (double) (ctr & 1023) * (1. / 1024.)); }
    if (OP == 6) { acc += exp(-1e-3 * (double) (ctr & 1023)); }
    ctr += (uint64_t) (acc > 1e300) + 1;     // data dependence, never more than +1
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0)
    cycles[0] = t1 - t0;
}

template <int OP>
void run(const char *name) {
  double *out;
  long long *cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  const int n = 20000;
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((k<OP>), dim3(1), dim3(64), 0, 0, out, cyc, n);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s %8.1f cycles per call (one wave, incl. loop overhead ~10)\n", name, (double) h / n);
}

int main() {
  run<0>("squares + uniform01");
  run<1>("normal_pair (Box-Muller)");
  run<2>("normal_triple");
  run<3>("log_unit");
  run<4>("sincosf (both)");
  run<5>("cos_latitude");
  run<6>("exp (library)");
  return 0;
}
