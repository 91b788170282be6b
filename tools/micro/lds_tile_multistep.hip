// x1 under the multi-step launch: a wind tile staged ONCE per launch in LDS against per-lane gathers, when every
// particle takes S time steps inside one launch (mphip_run_timesteps / the kMultiStep instantiations).
//
// Same grid, record layout and locality order as lds_tile.hip (721 x 361 x 137 cells of 24-byte two-snapshot wind
// records, level index fastest; particles in (4 x 4 column tile, level, column) order).  Per time step a particle
// evaluates five stencils -- four Runge-Kutta stages (displacements of up to 0.4 cells horizontally) and the one of
// module_diff_meso at the end position -- and then moves: `drift` cells per step horizontally in a direction of its
// own (workload C3: |u| up to 37 m/s x 180 s = 0.12 cells of 0.5 degrees at the equator, more towards the poles),
// a tenth of that vertically.
//
//   gather : per-lane loads of the corner records, re-fetched when the stencil cell changes (the wind-corner cache);
//            what the step kernel does -- from the second step on the lines come from the L1 / L2
//   tile   : the workgroup stages the bounding box of its particles' start cells, widened by `halo` cells on every
//            side, once per launch (at most `cap` cells, levels cut first); every stencil inside the tile is read from
//            LDS, the others from global memory
// LDS budget: the step kernel runs four workgroups of 256 per CU (four waves per SIMD) and each already holds 31 KB
// (axes, tropopause climatology, logarithm table): 9 KB per workgroup are free (cap = 384 cells); 24 KB (cap = 1024)
// leave three workgroups per CU, 64 KB (cap = 2730) one or two.  The kernel is compiled per cap so that the
// occupancy follows the LDS it asks for.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_tile_ms tools/micro/lds_tile_multistep.hip && ./lds_tile_ms [N] [steps] [drift] [halo]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NX = 721, NY = 361, NP = 137;

struct Rec {
  float2 a, b, c;   // {u0,v0} {u1,v1} {w0,w1}
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ void interpolate(const Rec r[8], double fx, double fy, double fz, double wt, double &u,
                                            double &v, double &w) {
  double cu[8], cv[8], cw[8];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    cu[c] = wt * (double) (r[c].b.x - r[c].a.x) + (double) r[c].a.x;
    cv[c] = wt * (double) (r[c].b.y - r[c].a.y) + (double) r[c].a.y;
    cw[c] = wt * (double) (r[c].c.y - r[c].c.x) + (double) r[c].c.x;
  }
  auto tri = [&](const double *q) {
    const double a0 = q[0] + fz * (q[1] - q[0]), a1 = q[2] + fz * (q[3] - q[2]);
    const double a2 = q[4] + fz * (q[5] - q[4]), a3 = q[6] + fz * (q[7] - q[6]);
    const double b0 = a0 + fy * (a1 - a0), b1 = a2 + fy * (a3 - a2);
    return b0 + fx * (b1 - b0);
  };
  u += tri(cu);
  v += tri(cv);
  w += tri(cw);
}

struct Motion {
  double vx, vy, vz;   // stage displacement scale (cells)
  double dx, dy, dz;   // drift per step (cells)
};

__device__ __forceinline__ Motion motion_of(long long i, double drift) {
  const unsigned h = hash32((unsigned) i), g = hash32((unsigned) i ^ 0x9e3779b9u);
  Motion m;
  m.vx = ((h & 1023) / 1023.0 - 0.5) * 0.8;
  m.vy = (((h >> 10) & 1023) / 1023.0 - 0.5) * 0.8;
  m.vz = (((h >> 20) & 1023) / 1023.0 - 0.5) * 0.2;
  const double ang = 6.283185307179586 * (g & 65535) / 65536.0;
  m.dx = drift * cos(ang);
  m.dy = drift * sin(ang);
  m.dz = 0.1 * drift * (((g >> 16) & 255) / 255.0 - 0.5);
  return m;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// TILE = 0: gathers only; otherwise the number of cells of the LDS tile
template <int TILE>
__global__ __launch_bounds__(256) void steps_kernel(const Rec *__restrict__ wind, const double *__restrict__ px,
                                                    const double *__restrict__ py, const double *__restrict__ pz,
                                                    long long n, int nsteps, double drift, int halo,
                                                    double *__restrict__ out, unsigned long long *__restrict__ stats) {
  __shared__ int s_lo[3], s_hi[3];
  __shared__ Rec s_tile[TILE > 0 ? TILE : 1];
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const bool live = i < n;
  double x = live ? px[i] : 0, y = live ? py[i] : 0, z = live ? pz[i] : 0;
  int x0 = 0, y0 = 0, z0 = 0, nx = 0, ny = 0, nz = 0;
  if (TILE > 0) {
    if (threadIdx.x < 3) {
      s_lo[threadIdx.x] = 1 << 30;
      s_hi[threadIdx.x] = -1;
    }
    __syncthreads();
    int lo[3] = { live ? (int) x : 1 << 30, live ? (int) y : 1 << 30, live ? (int) z : 1 << 30 };
    int hi[3] = { live ? (int) x + 1 : -1, live ? (int) y + 1 : -1, live ? (int) z + 1 : -1 };
    for (int d = 0; d < 3; d++) {
      for (int s = 32; s > 0; s >>= 1) {
        lo[d] = min(lo[d], __shfl_xor(lo[d], s));
        hi[d] = max(hi[d], __shfl_xor(hi[d], s));
      }
      if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_lo[d], lo[d]);
        atomicMax(&s_hi[d], hi[d]);
      }
    }
    __syncthreads();
    x0 = max(s_lo[0] - halo, 0); y0 = max(s_lo[1] - halo, 0); z0 = max(s_lo[2] - 1, 0);
    nx = min(s_hi[0] + halo, NX - 1) - x0 + 1; ny = min(s_hi[1] + halo, NY - 1) - y0 + 1; nz = min(s_hi[2] + 1, NP - 1) - z0 + 1;
    if (nx * ny * 2 > TILE) {      // too many columns: keep a square part around the start
      const int side = max(2, (int) sqrt((double) (TILE / 2)));
      nx = min(nx, side);
      ny = min(ny, max(2, TILE / 2 / nx));
    }
    nz = min(nz, TILE / (nx * ny));
    const int ncell = nx * ny * nz;
    const float2 *src = (const float2 *) wind;
    float2 *dst = (float2 *) s_tile;
    const int per_col = nz * 3;
    for (int f = threadIdx.x; f < ncell * 3; f += 256) {
      const int col = f / per_col, within = f - col * per_col;
      const int cx = col / ny, cy = col - cx * ny;
      dst[f] = src[(((size_t) (x0 + cx) * NY + (y0 + cy)) * NP + z0) * 3 + within];
    }
    __syncthreads();
    if (stats && threadIdx.x == 0)
      atomicAdd(&stats[0], (unsigned long long) ncell);
  }
  if (!live)
    return;
  const Motion m = motion_of(i, drift);
  double acc = 0;
  unsigned from_global = 0, fetches = 0;
  for (int step = 0; step < nsteps; step++) {
    Rec r[8];
    int cx = -1, cy = -1, cz = -1;      // (the corner cache starts empty in every step, as in the step kernel)
    double u = 0, v = 0, w = 0;
    for (int k = 0; k < 5; k++) {        // four stages + the stencil of module_diff_meso at the end position
      const double f = k < 4 ? 0.25 * k : 1.0;
      const double xs = clampd(x + f * m.vx, 0.0, NX - 1.001), ys = clampd(y + f * m.vy, 0.0, NY - 1.001),
                   zs = clampd(z + f * m.vz, 0.0, NP - 1.001);
      const int ix = (int) xs, iy = (int) ys, iz = (int) zs;
      if (ix != cx || iy != cy || iz != cz) {
        fetches++;
        const int tx = ix - x0, ty = iy - y0, tz = iz - z0;
        if (TILE > 0 && tx >= 0 && ty >= 0 && tz >= 0 && tx + 1 < nx && ty + 1 < ny && tz + 1 < nz) {
#pragma unroll
          for (int c = 0; c < 8; c++)
            r[c] = s_tile[((tx + (c >> 2)) * ny + ty + ((c >> 1) & 1)) * nz + tz + (c & 1)];
        } else {
          from_global++;
#pragma unroll
          for (int c = 0; c < 8; c++)
            r[c] = wind[((size_t) (ix + (c >> 2)) * NY + (iy + ((c >> 1) & 1))) * NP + iz + (c & 1)];
        }
        cx = ix; cy = iy; cz = iz;
      }
      interpolate(r, xs - ix, ys - iy, zs - iz, 0.2 * k, u, v, w);
    }
    acc += u + v + w;
    x = clampd(x + m.dx, 0.0, NX - 1.001);
    y = clampd(y + m.dy, 0.0, NY - 1.001);
    z = clampd(z + m.dz, 0.0, NP - 1.001);
  }
  out[i] = acc;
  if (stats) {
    atomicAdd(&stats[2], (unsigned long long) from_global);
    atomicAdd(&stats[3], (unsigned long long) fetches);
  }
}

int main(int argc, char **argv) {
  const long long n = argc > 1 ? (long long) atof(argv[1]) : 10000000;
  const int nsteps = argc > 2 ? atoi(argv[2]) : 20;
  const double drift = argc > 3 ? atof(argv[3]) : 0.1;
  const int halo = argc > 4 ? atoi(argv[4]) : 1;
  std::vector<double> x(n), y(n), z(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double) (s >> 11) / 9007199254740992.0; };
  for (long long i = 0; i < n; i++) {
    x[i] = rnd() * (NX - 1.001);
    y[i] = rnd() * (NY - 1.001);
    z[i] = 40.0 + rnd() * 70.0;
  }
  std::vector<unsigned long long> key(n);
  for (long long i = 0; i < n; i++) {
    const int ix = (int) x[i], iy = (int) y[i], iz = (int) z[i];
    key[i] = ((((unsigned long long) (ix / 4) * ((NY + 3) / 4) + iy / 4) * NP + iz) * 16 + (ix % 4) * 4 + iy % 4) << 32
      | (unsigned long long) i;
  }
  std::sort(key.begin(), key.end());
  std::vector<double> sx(n), sy(n), sz(n);
  for (long long i = 0; i < n; i++) {
    const long long j = (long long) (key[i] & 0xffffffffull);
    sx[i] = x[j]; sy[i] = y[j]; sz[i] = z[j];
  }
  const size_t ncell = (size_t) NX * NY * NP;
  Rec *wind;
  double *dx_, *dy_, *dz_, *out;
  unsigned long long *stats;
  hipMalloc(&wind, ncell * sizeof(Rec));
  hipMemset(wind, 0x3c, ncell * sizeof(Rec));
  hipMalloc(&dx_, n * 8); hipMalloc(&dy_, n * 8); hipMalloc(&dz_, n * 8); hipMalloc(&out, n * 8);
  hipMalloc(&stats, 4 * sizeof(unsigned long long));
  hipMemcpy(dx_, sx.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dy_, sy.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dz_, sz.data(), n * 8, hipMemcpyHostToDevice);
  const int nb = (int) ((n + 255) / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_of = [&](auto launch) {
    launch((unsigned long long *) nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 3; rep++)
      launch((unsigned long long *) nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
  };
  printf("N %.3g, %d steps per launch, drift %.2f cells per step, halo %d cells (%.2f particles per cell of the occupied levels)\n",
         (double) n, nsteps, drift, halo, (double) n / ((double) NX * NY * 70));
  std::vector<double> ref(std::min<long long>(n, 100000)), got(ref.size());
  auto run = [&](const char *name, auto kernel, int cap) {
    auto launch = [&](unsigned long long *st) {
      hipLaunchKernelGGL(kernel, dim3(nb), dim3(256), 0, 0, wind, dx_, dy_, dz_, n, nsteps, drift, halo, out, st);
    };
    const float ms = time_of(launch);
    hipMemset(stats, 0, 4 * sizeof(unsigned long long));
    launch(stats);
    hipDeviceSynchronize();
    unsigned long long st[4];
    hipMemcpy(st, stats, sizeof(st), hipMemcpyDeviceToHost);
    hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost);
    if (cap == 0)
      ref = got;
    size_t bad = 0;
    for (size_t i = 0; i < ref.size(); i++)
      bad += ref[i] != got[i];
    int blocks_per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, 256, 0);
    printf("  %-22s %8.3f ms per launch = %7.4f ms per step   workgroups per CU %d   %.2f stencil fetches per particle-step, "
           "%.1f %% of them from global memory", name, ms, ms / nsteps, blocks_per_cu, (double) st[3] / (double) n / nsteps,
           100.0 * (double) st[2] / (double) std::max(st[3], 1ull));
    if (cap)
      printf(", %.0f cells staged per workgroup", (double) st[0] / nb);
    printf("   results %s\n", bad ? "DIFFER" : "identical");
    return bad;
  };
  size_t bad = 0;
  bad += run("gather", steps_kernel<0>, 0);
  bad += run("tile   9 KB (384)", steps_kernel<384>, 384);
  bad += run("tile  24 KB (1024)", steps_kernel<1024>, 1024);
  bad += run("tile  64 KB (2730)", steps_kernel<2730>, 2730);
  return bad != 0;
}
