// Issue cost of single VALU instructions on gfx950 with the SIMDs full: W waves per SIMD (workgroups of
// 256 threads on every CU), each wave running CH independent chains of one instruction, inline assembly so
// that the instruction is the one named.  Reports shader cycles per wave-instruction per SIMD
// (= time * clock / instructions issued on one SIMD): the number the step kernel's cost model needs.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/micro/valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CH = 8;
constexpr int UNROLL = 16;

enum Op {
  FMA_F64, ADD_F64, MUL_F64, FMA_F32, PK_FMA_F32, PK_ADD_F32, ADD_U32, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32,
  MAD_U32_U24, CVT_F64_F32, CVT_F32_F64, CVT_F64_U32, CVT_I32_F64, CNDMASK, MOV_B32, CMP_F64, RCP_F64, RSQ_F64,
  RCP_F32, SQRT_F32, LSHL_B64, MIN_F64, LDEXP_F64, XOR_B32, ALIGNBIT, ADD_CO, FMA_F64_SGPR, PERM_B32, NOPS,
  CNDMASK_SGPR, CNDMASK_AFTER_CMP, CNDMASK_OTHER_DST, CNDMASK_FMA_MIX, CMP_CNDMASK_PAIR, CNDMASK_VCC_SET, MAX_F64_PAIR, CMP_F32, CVT_F32_U32, MUL_F32, BFE_U32, LSHL_ADD_U32, AND_OR_B32, MOV_B64,
  CND_E64_VCC, CND2_FMA, CND3_FMA, CND4_FMA, SEL64_VCC, SEL64_SGPR, CND2_NOP, CND2_ADD32, CND_CND_DIFFMASK, SEL64_VCC_X2
};

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int n, double a, double b) {
  double x[CH];
  float f[CH];
  float g[CH];
  uint32_t u[CH];
  uint32_t w[CH];
  for (int c = 0; c < CH; c++) {
    x[c] = 1.0 + 1e-3 * (threadIdx.x + c);
    f[c] = (float) x[c];
    g[c] = f[c] + 1.f;
    u[c] = threadIdx.x * 7u + c + 1u;
    w[c] = u[c] * 3u;
  }
  const float af = (float) a, bf = (float) b;
  const uint64_t smask = __builtin_amdgcn_read_exec() & 0x5555555555555555ull;
  if (OP == CNDMASK_AFTER_CMP) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x[0]), "v"(a) : "vcc");
  if (OP == CNDMASK_VCC_SET) asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc");
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int r = 0; r < UNROLL; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        if (OP == FMA_F64_SGPR) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "s"(a), "v"(b));
        if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b));
        if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
        if (OP == MIN_F64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
        if (OP == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[c]));
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(af), "v"(bf));
        if (OP == PK_FMA_F32) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 v = { f[c], g[c] };
          const f2 av = { af, af }, bv = { bf, bf };
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
          f[c] = v[0];
          g[c] = v[1];
        }
        if (OP == PK_ADD_F32) {
          typedef float f2 __attribute__((ext_vector_type(2)));
          f2 v = { f[c], g[c] };
          const f2 bv = { bf, bf };
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(bv));
          f[c] = v[0];
          g[c] = v[1];
        }
        if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(u[c]) : "v"(w[c]));
        if (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u[c]) : "v"(w[c]) : "vcc");
        if (OP == MAD_U64_U32) {
          uint64_t acc = ((uint64_t) w[c] << 32) | u[c];
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(u[c]), "v"(w[c]) : "vcc");
          u[c] = (uint32_t) acc;
          w[c] = (uint32_t) (acc >> 32);
        }
        if (OP == LSHL_B64) {
          uint64_t acc = ((uint64_t) w[c] << 32) | u[c];
          asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(acc));
          u[c] = (uint32_t) acc;
          w[c] = (uint32_t) (acc >> 32);
        }
        if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[c]) : "v"(f[c]));
        if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(x[c]));
        if (OP == CVT_F64_U32) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x[c]) : "v"(u[c]));
        if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[c]) : "v"(x[c]));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
        if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(w[c]));
        if (OP == CMP_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x[c]), "v"(a) : "vcc");
        if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c]));
        if (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[c]));
        if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
        if (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
        if (OP == NOPS) asm volatile("s_nop 0");
        if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[c]) : "v"(w[c]), "s"(smask));
        if (OP == CNDMASK_AFTER_CMP) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
        if (OP == CNDMASK_VCC_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
        if (OP == CNDMASK_OTHER_DST) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[c]) : "v"(w[c]), "v"(w[(c + 1) % CH]));
        if (OP == CNDMASK_FMA_MIX) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        }
        if (OP == CMP_CNDMASK_PAIR) {
          asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[c]) : "v"(x[c]), "v"(a), "v"(w[c]) : "vcc");
        }
        if (OP == MAX_F64_PAIR) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a));
        if (OP == CMP_F32) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[c]), "v"(af) : "vcc");
        if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[c]) : "v"(u[c]));
        if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[c]) : "v"(af));
        if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(u[c]));
        if (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[c]) : "v"(w[c]));
        if (OP == CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
        if (OP == CND2_FMA || OP == CND3_FMA || OP == CND4_FMA) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[c]) : "v"(u[c]));
          if (OP != CND2_FMA) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
          if (OP == CND4_FMA) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[c]) : "v"(u[c]));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
        }
        if (OP == SEL64_VCC) {   // the compiler's 64-bit select
          asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc"
                       : "+v"(u[c]), "+v"(w[c]) : "v"(x[c]), "v"(a), "v"(u[(c + 1) % CH]) : "vcc");
        }
        if (OP == SEL64_VCC_X2) {   // two selects in a row
          asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\t"
                       "v_cmp_gt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc"
                       : "+v"(u[c]), "+v"(w[c]) : "v"(x[c]), "v"(a), "v"(u[(c + 1) % CH]) : "vcc");
        }
        if (OP == SEL64_SGPR) {
          uint64_t m;
          asm volatile("v_cmp_lt_f64 %2, %3, %4\n\tv_cndmask_b32_e64 %0, %0, %5, %2\n\tv_cndmask_b32_e64 %1, %1, %5, %2"
                       : "+v"(u[c]), "+v"(w[c]), "=&s"(m) : "v"(x[c]), "v"(a), "v"(u[(c + 1) % CH]));
        }
        if (OP == CND2_NOP) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\ts_nop 0" : "+v"(u[c]) : "v"(w[c]));
        }
        if (OP == CND2_ADD32) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[c]) : "v"(u[c]));
        }
        if (OP == CND_CND_DIFFMASK) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(w[c]));
          asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(w[c]) : "v"(u[c]), "s"(smask));
        }
        if (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(x[c]) : "v"(x[(c + 1) % CH]));
      }
  }
  double s = 0;
  for (int c = 0; c < CH; c++)
    s += x[c] + f[c] + g[c] + u[c] + w[c];
  out[(size_t) blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_clock_hz = 2.4e9;

template <int OP>
void run(const char *name, int waves_per_simd) {
  int dev = 0, cus = 0;
  CHK(hipGetDevice(&dev));
  CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; W blocks per CU
  double *out;
  CHK(hipMalloc(&out, (size_t) blocks * 256 * sizeof(double)));
  const int n = 2000;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, n, 0.999999, 1e-7);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best)
      best = ms;
  }
  const double per_simd = (double) n * UNROLL * CH * waves_per_simd;   // wave-instructions issued on one SIMD
  printf("%-16s %d waves/SIMD: %6.2f cycles per wave-instruction per SIMD at %.2f GHz (%.3f ms)\n", name, waves_per_simd,
         best * 1e-3 * g_clock_hz / per_simd, g_clock_hz * 1e-9, best);
  CHK(hipFree(out));
}

#define RUN(op) run<op>(#op, 1); run<op>(#op, 4);

int main() {
  int clk = 0;
  CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  g_clock_hz = clk * 1e3;
  printf("shader clock %d kHz (nominal: the numbers assume it is sustained)\n", clk);
  // bring the clocks up
  for (int i = 0; i < 6; i++)
    run<FMA_F64>("warm-up", 4);
  RUN(FMA_F64) RUN(FMA_F64_SGPR) RUN(ADD_F64) RUN(MUL_F64) RUN(MIN_F64) RUN(LDEXP_F64) RUN(CMP_F64)
  RUN(FMA_F32) RUN(PK_FMA_F32) RUN(PK_ADD_F32)
  RUN(ADD_U32) RUN(XOR_B32) RUN(ADD_CO) RUN(ALIGNBIT) RUN(PERM_B32) RUN(CNDMASK) RUN(MOV_B32) RUN(LSHL_B64)
  RUN(MUL_LO_U32) RUN(MUL_HI_U32) RUN(MAD_U64_U32) RUN(MAD_U32_U24)
  RUN(CVT_F64_F32) RUN(CVT_F32_F64) RUN(CVT_F64_U32) RUN(CVT_I32_F64)
  RUN(RCP_F64) RUN(RSQ_F64) RUN(RCP_F32) RUN(SQRT_F32) RUN(NOPS)
  RUN(CNDMASK_SGPR) RUN(CNDMASK_AFTER_CMP) RUN(CNDMASK_VCC_SET) RUN(CNDMASK_OTHER_DST) RUN(CNDMASK_FMA_MIX) RUN(CMP_CNDMASK_PAIR)
  RUN(MAX_F64_PAIR) RUN(CMP_F32) RUN(CVT_F32_U32) RUN(MUL_F32) RUN(BFE_U32) RUN(LSHL_ADD_U32) RUN(AND_OR_B32) RUN(MOV_B64)
  printf("-- sequences: cycles per SEQUENCE (all instructions of one element)\n");
  RUN(CND_E64_VCC) RUN(CND2_FMA) RUN(CND3_FMA) RUN(CND4_FMA) RUN(SEL64_VCC) RUN(SEL64_VCC_X2) RUN(SEL64_SGPR) RUN(CND2_NOP) RUN(CND2_ADD32) RUN(CND_CND_DIFFMASK)
  return 0;
}
