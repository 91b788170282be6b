// Throughput of per-lane gathers through the vector memory path (TA / TCP / TD) of one CU, by access width
// and alignment, with the data resident in the L1 (TCP) so that only the address / data-return path counts.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate tools/micro/gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f4a __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

template <int MODE>
__global__ void gather(const float *__restrict__ buf, const int *__restrict__ idx, float *out, long long *cycles, int n,
                       int span) {
  // per-lane pseudo-random cell sequence inside `span` bytes (L1-resident for small spans)
  unsigned s = idx[threadIdx.x + blockIdx.x * blockDim.x];
  float acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      s = s * 1664525u + 1013904223u;
      const unsigned cell = (s >> 8) & (unsigned) (span / 64 - 1);  // 64-byte cells (span is a power of two)
      const char *p = (const char *) buf + (size_t) cell * 64;
      if (MODE == 0) acc += *(const float *) p;                                           // dword
      if (MODE == 1) { f2u v = *(const f2u *) p; acc += v[0] + v[1]; }                    // dwordx2
      if (MODE == 2) { f4a v = *(const f4a *) p; acc += v[0] + v[3]; }                    // dwordx4, 16-byte aligned
      if (MODE == 3) { f4u v = *(const f4u *) (p + 8); acc += v[0] + v[3]; }              // dwordx4, 8 bytes off
      if (MODE == 4) { f4u a = *(const f4u *) (p + 8), b = *(const f4u *) (p + 24), c = *(const f4u *) (p + 40);
                       acc += a[0] + b[1] + c[2]; }                                       // 48 B as 3 x dwordx4 (+8)
      if (MODE == 5) { f4a a = *(const f4a *) p, b = *(const f4a *) (p + 16), c = *(const f4a *) (p + 32);
                       acc += a[0] + b[1] + c[2]; }                                       // 48 B as 3 aligned x4
      if (MODE == 7) { if ((threadIdx.x & 3) == 0) { f4a v = *(const f4a *) p; acc += v[0] + v[3]; } }    // 1 lane in 4 active
      if (MODE == 8) { if ((threadIdx.x & 15) == 0) { f4a v = *(const f4a *) p; acc += v[0] + v[3]; } }   // 1 lane in 16
      if (MODE == 9) { const char *q = (const char *) buf + (size_t) (cell & ~15u) * 64 + (threadIdx.x & 15) * 64;
                       f4a v = *(const f4a *) q; acc += v[0] + v[3]; }                    // 16 consecutive cells per 16 lanes
      if (MODE == 6) { f4a a = *(const f4a *) p, b = *(const f4a *) (p + 16), c = *(const f4a *) (p + 32),
                       d = *(const f4a *) (p + 48); acc += a[0] + b[1] + c[2] + d[3]; }   // 64 B as 4 aligned x4
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
  if (threadIdx.x == 0)
    cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int span, int loads_per_iter, int bytes_per_iter) {
  const int waves = 8, n = 2000;
  float *buf, *out;
  int *idx;
  long long *cyc;
  hipMalloc(&buf, 64 << 20);
  hipMemset(buf, 0, 64 << 20);
  hipMalloc(&out, 64 * waves * sizeof(float));
  hipMalloc(&idx, 64 * waves * sizeof(int));
  hipMalloc(&cyc, sizeof(long long));
  std::vector<int> h(64 * waves);
  for (size_t i = 0; i < h.size(); i++)
    h[i] = (int) (i * 2654435761u);
  hipMemcpy(idx, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((gather<MODE>), dim3(1), dim3(64 * waves), 0, 0, buf, idx, out, cyc, n, span);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  const double per_group = (double) c / ((double) n * 8 * waves);     // cycles per wave-level gather group on the CU
  printf("%-34s span %8d B: %7.1f cycles per wave gather (%d load instr, %3d B/lane) -> %5.1f B/clk/CU\n", name, span,
         per_group, loads_per_iter, bytes_per_iter, 64.0 * bytes_per_iter / per_group);
  hipFree(buf); hipFree(out); hipFree(idx); hipFree(cyc);
}

int main() {
  for (int span : { 8192, 2 << 20 }) {
    run<0>("dword", span, 1, 4);
    run<1>("dwordx2 (4-aligned)", span, 1, 8);
    run<2>("dwordx4 aligned", span, 1, 16);
    run<3>("dwordx4 +8 B", span, 1, 16);
    run<4>("3 x dwordx4 +8 B (48 B)", span, 3, 48);
    run<5>("3 x dwordx4 aligned (48 B)", span, 3, 48);
    run<6>("4 x dwordx4 aligned (64 B)", span, 4, 64);
    run<7>("dwordx4, 1 lane in 4 active", span, 1, 4);
    run<8>("dwordx4, 1 lane in 16 active", span, 1, 1);
    run<9>("dwordx4, 16-lane groups contiguous", span, 1, 16);
  }
  return 0;
}
