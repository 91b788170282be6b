// Cost of one wave-level 16-byte-per-lane gather (global_load_dwordx4) as a function of how many distinct
// 128-byte lines its 64 lanes touch, data resident in the L1 / L2 (small span): lane l reads 16 bytes at
// base + l * stride, base uniform per instruction and random.
//   hipcc --offload-arch=gfx950 -O3 -o gather_lines tools/micro/gather_lines.hip && ./gather_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4a __attribute__((ext_vector_type(4)));

__global__ void gather(const float *__restrict__ buf, float *out, long long *cycles, int n, int span, int stride,
                       int wrap) {
  unsigned s = 12345u + blockIdx.x * 977u + (threadIdx.x >> 6) * 131u;   // uniform per wave
  const int lane = threadIdx.x & 63;
  float acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      s = s * 1664525u + 1013904223u;
      const unsigned base = ((s >> 8) & (unsigned) (span / 8192 - 1)) * 8192u;      // 8 KB aligned window
      const char *p = (const char *) buf + base + (size_t) ((lane * stride) % wrap);
      const f4a v = *(const f4a *) p;
      acc += v[0] + v[3];
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
  if (threadIdx.x == 0)
    cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int waves = 8, n = 2000;
  float *buf, *out;
  long long *cyc;
  hipMalloc(&buf, 64 << 20);
  hipMemset(buf, 0, 64 << 20);
  hipMalloc(&out, 64 * waves * sizeof(float));
  hipMalloc(&cyc, sizeof(long long));
  struct { const char *name; int stride, wrap; } pat[] = {
    { "all lanes one address", 0, 8192 },          { "4 lines, two lanes per 16 B", 16, 512 },
    { "8 lines (contiguous 1 KB)", 16, 8192 },     { "16 lines (stride 32 B)", 32, 8192 },
    { "32 lines (stride 64 B)", 64, 8192 },        { "64 lines (stride 128 B)", 128, 8192 },
  };
  for (int span : { 65536, 4 << 20 })
    for (auto &p : pat) {
      for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL(gather, dim3(1), dim3(64 * waves), 0, 0, buf, out, cyc, n, span, p.stride, p.wrap);
      hipDeviceSynchronize();
      long long c;
      hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
      printf("span %8d B  %-30s %7.1f cycles per wave gather\n", span, p.name, (double) c / ((double) n * 8 * waves));
    }
  return 0;
}
