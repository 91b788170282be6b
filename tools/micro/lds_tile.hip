// LDS-tile staging of the wind grid against per-lane gathers, on the access pattern of the step kernel's
// Runge-Kutta advection (SURVEY x1 / north_star "met grids staged through LDS tiles per thread-block").
//
// Grid: 721 x 361 x 137 cells of 24-byte two-snapshot wind records {u0,v0,u1,v1,w0,w1}, level index fastest
// (the layout of mphip's packed grids).  Particles: N uniformly scattered over the 70 occupied levels (N = 1e7:
// 0.55 particles per cell, the density of bench workload C3), stored in the locality order of the back end
// (4 x 4 column tile, level, column) -- "fresh" right after the sort, or "aged": the same storage order after
// the particles moved by up to +-`age` cells (what the order looks like many steps after a re-sort).
// Work per particle: four stage positions (a displacement of up to 0.4 cells), at each one the eight corner
// records of the stencil (re-fetched only when the stencil cell changed, as the wind-corner cache of the step
// kernel does), tri-linear + time interpolation.
//
//   gather : every lane fetches its corner records from global memory (3 x 8-byte loads per corner)
//   tile   : a workgroup (256 consecutive particles) finds the bounding box of all its stencils (LDS atomic
//            min / max), loads the box cooperatively into LDS (coalesced along the levels; at most `cap`
//            cells, the levels are cut first when the box is larger), then every lane reads its corners from
//            LDS -- or from global memory when its stencil lies outside the staged part
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_tile tools/micro/lds_tile.hip && ./lds_tile [N] [age]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

constexpr int NX = 721, NY = 361, NP = 137;
constexpr int kCapCells = 1024;   // 24 KB of LDS per workgroup for the tile

struct Rec {
  float2 a, b, c;   // {u0,v0} {u1,v1} {w0,w1}
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// stage k position of particle i (grid units)
__device__ __forceinline__ void stage_pos(int k, long long i, double x, double y, double z, double &xs, double &ys,
                                          double &zs) {
  const unsigned h = hash32((unsigned) i);
  const double vx = ((h & 1023) / 1023.0 - 0.5) * 0.8, vy = (((h >> 10) & 1023) / 1023.0 - 0.5) * 0.8,
               vz = (((h >> 20) & 1023) / 1023.0 - 0.5) * 0.2;
  xs = fmin(fmax(x + 0.25 * k * vx, 0.0), NX - 1.001);
  ys = fmin(fmax(y + 0.25 * k * vy, 0.0), NY - 1.001);
  zs = fmin(fmax(z + 0.25 * k * vz, 0.0), NP - 1.001);
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

__global__ __launch_bounds__(256) void gather_kernel(const Rec *__restrict__ wind, const double *__restrict__ px,
                                                     const double *__restrict__ py, const double *__restrict__ pz,
                                                     long long n, double *__restrict__ out) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n)
    return;
  const double x = px[i], y = py[i], z = pz[i];
  Rec r[8];
  int cx = -1, cy = -1, cz = -1;
  double u = 0, v = 0, w = 0;
  for (int k = 0; k < 4; k++) {
    double xs, ys, zs;
    stage_pos(k, i, x, y, z, xs, ys, zs);
    const int ix = (int) xs, iy = (int) ys, iz = (int) zs;
    if (ix != cx || iy != cy || iz != cz) {
#pragma unroll
      for (int c = 0; c < 8; c++)
        r[c] = wind[((size_t) (ix + (c >> 2)) * NY + (iy + ((c >> 1) & 1))) * NP + iz + (c & 1)];
      cx = ix; cy = iy; cz = iz;
    }
    interpolate(r, xs - ix, ys - iy, zs - iz, 0.25 * k, u, v, w);
  }
  out[i] = u + v + w;
}

__global__ __launch_bounds__(256) void tile_kernel(const Rec *__restrict__ wind, const double *__restrict__ px,
                                                   const double *__restrict__ py, const double *__restrict__ pz,
                                                   long long n, double *__restrict__ out,
                                                   unsigned long long *__restrict__ stats) {
  __shared__ int s_lo[3], s_hi[3];
  __shared__ Rec s_tile[kCapCells];
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const bool live = i < n;
  if (threadIdx.x < 3) {
    s_lo[threadIdx.x] = 1 << 30;
    s_hi[threadIdx.x] = -1;
  }
  __syncthreads();
  const double x = live ? px[i] : 0, y = live ? py[i] : 0, z = live ? pz[i] : 0;
  int lo[3] = { 1 << 30, 1 << 30, 1 << 30 }, hi[3] = { -1, -1, -1 };
  if (live)
    for (int k = 0; k < 4; k++) {
      double xs, ys, zs;
      stage_pos(k, i, x, y, z, xs, ys, zs);
      const int b[3] = { (int) xs, (int) ys, (int) zs };
      for (int d = 0; d < 3; d++) {
        lo[d] = min(lo[d], b[d]);
        hi[d] = max(hi[d], b[d] + 1);
      }
    }
  for (int d = 0; d < 3; d++) {   // wave reduction, then one LDS atomic per wave and bound
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
  const int x0 = s_lo[0], y0 = s_lo[1], z0 = s_lo[2];
  int nx = s_hi[0] - x0 + 1, ny = s_hi[1] - y0 + 1, nz = s_hi[2] - z0 + 1;
  // too large for the tile: keep the first columns / levels that fit (the rest falls back to global loads)
  if (nx * ny * 2 > kCapCells) {
    nx = min(nx, 16);
    ny = min(ny, max(2, kCapCells / 2 / nx));
  }
  nz = min(nz, kCapCells / (nx * ny));
  const int ncell = nx * ny * nz;
  // cooperative load, 8-byte pieces, contiguous along the levels of a column
  {
    const float2 *src = (const float2 *) wind;
    float2 *dst = (float2 *) s_tile;
    const int per_col = nz * 3;
    for (int f = threadIdx.x; f < ncell * 3; f += 256) {
      const int col = f / per_col, within = f - col * per_col;
      const int cx = col / ny, cy = col - cx * ny;
      dst[f] = src[(((size_t) (x0 + cx) * NY + (y0 + cy)) * NP + z0) * 3 + within];
    }
  }
  __syncthreads();
  if (!live)
    return;
  Rec r[8];
  int cx = -1, cy = -1, cz = -1;
  double u = 0, v = 0, w = 0;
  unsigned from_global = 0, fetches = 0;
  for (int k = 0; k < 4; k++) {
    double xs, ys, zs;
    stage_pos(k, i, x, y, z, xs, ys, zs);
    const int ix = (int) xs, iy = (int) ys, iz = (int) zs;
    if (ix != cx || iy != cy || iz != cz) {
      const int tx = ix - x0, ty = iy - y0, tz = iz - z0;
      fetches++;
      if (tx + 1 < nx && ty + 1 < ny && tz + 1 < nz) {
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
    interpolate(r, xs - ix, ys - iy, zs - iz, 0.25 * k, u, v, w);
  }
  out[i] = u + v + w;
  if (stats) {
    if (threadIdx.x == 0) {
      atomicAdd(&stats[0], (unsigned long long) ncell);
      atomicAdd(&stats[1], (unsigned long long) ((s_hi[0] - x0 + 1) * (s_hi[1] - y0 + 1) * (s_hi[2] - z0 + 1)));
    }
    if (from_global)
      atomicAdd(&stats[2], (unsigned long long) from_global);
    atomicAdd(&stats[3], (unsigned long long) fetches);
  }
}

int main(int argc, char **argv) {
  const long long n = argc > 1 ? (long long) atof(argv[1]) : 10000000;
  const double age = argc > 2 ? atof(argv[2]) : 0.0;
  std::vector<double> x(n), y(n), z(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double) (s >> 11) / 9007199254740992.0; };
  for (long long i = 0; i < n; i++) {
    x[i] = rnd() * (NX - 1.001);
    y[i] = rnd() * (NY - 1.001);
    z[i] = 40.0 + rnd() * 70.0;
  }
  // locality order of the back end: (4 x 4 column tile, level, column inside the tile)
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
    // aged order: neighbours of the sorted order moved together (a smooth displacement field) and apart
    // (an individual part of a fifth of it)
    const double dx = age * (sin(0.05 * x[j]) * cos(0.07 * y[j]) + 0.2 * (rnd() - 0.5));
    const double dy = age * (cos(0.06 * x[j]) * sin(0.04 * y[j]) + 0.2 * (rnd() - 0.5));
    const double dz = 0.2 * age * (rnd() - 0.5);
    sx[i] = std::min(std::max(x[j] + dx, 0.0), NX - 1.001);
    sy[i] = std::min(std::max(y[j] + dy, 0.0), NY - 1.001);
    sz[i] = std::min(std::max(z[j] + dz, 0.0), NP - 1.001);
  }
  const size_t ncell = (size_t) NX * NY * NP;
  Rec *wind;
  double *dx_, *dy_, *dz_, *out;
  unsigned long long *stats;
  hipMalloc(&wind, ncell * sizeof(Rec));
  hipMemset(wind, 0x3c, ncell * sizeof(Rec));
  hipMalloc(&dx_, n * 8); hipMalloc(&dy_, n * 8); hipMalloc(&dz_, n * 8); hipMalloc(&out, n * 8);
  hipMalloc(&stats, 4 * sizeof(unsigned long long));
  hipMemset(stats, 0, 4 * sizeof(unsigned long long));
  hipMemcpy(dx_, sx.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dy_, sy.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dz_, sz.data(), n * 8, hipMemcpyHostToDevice);
  const int nb = (int) ((n + 255) / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_of = [&](auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; rep++)
      launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
  };
  const float tg = time_of([&] { hipLaunchKernelGGL(gather_kernel, dim3(nb), dim3(256), 0, 0, wind, dx_, dy_, dz_, n, out); });
  std::vector<double> ref(std::min<long long>(n, 100000));
  hipMemcpy(ref.data(), out, ref.size() * 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(tile_kernel, dim3(nb), dim3(256), 0, 0, wind, dx_, dy_, dz_, n, out, stats);
  const float tt = time_of([&] { hipLaunchKernelGGL(tile_kernel, dim3(nb), dim3(256), 0, 0, wind, dx_, dy_, dz_, n, out,
                                                  (unsigned long long *) nullptr); });
  std::vector<double> got(ref.size());
  hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < ref.size(); i++)
    bad += ref[i] != got[i];
  unsigned long long st[4];
  hipMemcpy(st, stats, sizeof(st), hipMemcpyDeviceToHost);
  printf("N %.3g  age %.1f cells  (%.2f particles per cell of the occupied levels)\n", (double) n, age,
         (double) n / ((double) NX * NY * 70));
  printf("  gather kernel %8.3f ms\n", tg);
  printf("  tile   kernel %8.3f ms   staged cells per workgroup %.0f (bounding box %.0f), %.2f stencil fetches per "
         "particle, %.2f %% of them from global memory   results %s\n",
         tt, (double) st[0] / nb, (double) st[1] / nb, (double) st[3] / (double) n, 100.0 * (double) st[2] / (double) st[3],
         bad ? "DIFFER" : "identical");
  return bad != 0;
}
