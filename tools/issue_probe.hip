// How do the VALU and the matrix pipe share a SIMD on gfx950? Issue-rate probe for the attention kernels:
// cycles per instruction of the VALU ops their softmax uses, of v_mfma_f32_32x32x16_bf16, of both in ONE wave
// (1 MFMA : k VALU, independent registers) and of both on one SIMD from DIFFERENT waves (even waves MFMA, odd VALU).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define V1(op) asm volatile(op " %0, %0" : "+v"(r[0])); asm volatile(op " %0, %0" : "+v"(r[1])); \
               asm volatile(op " %0, %0" : "+v"(r[2])); asm volatile(op " %0, %0" : "+v"(r[3])); \
               asm volatile(op " %0, %0" : "+v"(r[4])); asm volatile(op " %0, %0" : "+v"(r[5])); \
               asm volatile(op " %0, %0" : "+v"(r[6])); asm volatile(op " %0, %0" : "+v"(r[7]));
#define V2(op) asm volatile(op " %0, %0, %0" : "+v"(r[0])); asm volatile(op " %0, %0, %0" : "+v"(r[1])); \
               asm volatile(op " %0, %0, %0" : "+v"(r[2])); asm volatile(op " %0, %0, %0" : "+v"(r[3])); \
               asm volatile(op " %0, %0, %0" : "+v"(r[4])); asm volatile(op " %0, %0, %0" : "+v"(r[5])); \
               asm volatile(op " %0, %0, %0" : "+v"(r[6])); asm volatile(op " %0, %0, %0" : "+v"(r[7]));
#define V3(op) asm volatile(op " %0, %0, %0, %0" : "+v"(r[0])); asm volatile(op " %0, %0, %0, %0" : "+v"(r[1])); \
               asm volatile(op " %0, %0, %0, %0" : "+v"(r[2])); asm volatile(op " %0, %0, %0, %0" : "+v"(r[3])); \
               asm volatile(op " %0, %0, %0, %0" : "+v"(r[4])); asm volatile(op " %0, %0, %0, %0" : "+v"(r[5])); \
               asm volatile(op " %0, %0, %0, %0" : "+v"(r[6])); asm volatile(op " %0, %0, %0, %0" : "+v"(r[7]));
#define P2(op) asm volatile(op " %0, %0, %0" : "+v"(q[0])); asm volatile(op " %0, %0, %0" : "+v"(q[1])); \
               asm volatile(op " %0, %0, %0" : "+v"(q[2])); asm volatile(op " %0, %0, %0" : "+v"(q[3])); \
               asm volatile(op " %0, %0, %0" : "+v"(q[4])); asm volatile(op " %0, %0, %0" : "+v"(q[5])); \
               asm volatile(op " %0, %0, %0" : "+v"(q[6])); asm volatile(op " %0, %0, %0" : "+v"(q[7]));
#define P3(op) asm volatile(op " %0, %0, %0, %0" : "+v"(q[0])); asm volatile(op " %0, %0, %0, %0" : "+v"(q[1])); \
               asm volatile(op " %0, %0, %0, %0" : "+v"(q[2])); asm volatile(op " %0, %0, %0, %0" : "+v"(q[3])); \
               asm volatile(op " %0, %0, %0, %0" : "+v"(q[4])); asm volatile(op " %0, %0, %0, %0" : "+v"(q[5])); \
               asm volatile(op " %0, %0, %0, %0" : "+v"(q[6])); asm volatile(op " %0, %0, %0, %0" : "+v"(q[7]));
#define CVT    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[0]) : "v"(r[0]), "v"(r[1])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[1]) : "v"(r[2]), "v"(r[3])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[2]) : "v"(r[4]), "v"(r[5])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[3]) : "v"(r[6]), "v"(r[7])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[4]) : "v"(r[1]), "v"(r[0])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[5]) : "v"(r[3]), "v"(r[2])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[6]) : "v"(r[5]), "v"(r[4])); \
               asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[7]) : "v"(r[7]), "v"(r[6]));
#define MF(i)  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#define MFAB(i, x, y) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
#define EXP1(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define FMA1(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));

// MODE: 0 exp, 1 fma, 2 pk_mul, 3 pk_fma, 4 cvt_pk_bf16, 5 mul_lo_u32, 6 xor, 7 mfma only, 8 exp2 via ldexp-free fma poly (skipped),
// 10 same wave 1 MFMA : 4 fma, 11 same wave 1 MFMA : 8 fma, 12 same wave 1 MFMA : 4 exp, 13 same wave 1 MFMA : 2 exp + 4 fma,
// 20 waves alternate by SIMD slot: (wave>>2)&1 ? VALU fma : MFMA, 21 ... exp : MFMA
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, float* sink, unsigned long long* clk, float seed) {
  float r[8];
  f32x2 q[8];
  uint32_t u[8];
  for (int i = 0; i < 8; ++i) { r[i] = seed * (i + 1) + threadIdx.x * 1e-6f; q[i] = f32x2{r[i], r[i] * 0.5f}; u[i] = 0; }
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + j + (threadIdx.x & 7)); b[j] = (short)(0x3F00 + 3 * j); }
  bf16x8 a2 = a, a3 = a, a4 = a, b2 = b, b3 = b, b4 = b;
  a2[0] += 1; a3[1] += 2; a4[2] += 3; b2[3] += 1; b3[4] += 2; b4[5] += 3;
  asm volatile("" : "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b2), "+v"(b3), "+v"(b4));
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const int wave = threadIdx.x >> 6;
  const bool odd = (wave >> 2) & 1;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { V1("v_exp_f32") V1("v_exp_f32") }
    if (MODE == 1) { V3("v_fma_f32") V3("v_fma_f32") }
    if (MODE == 2) { P2("v_pk_mul_f32") P2("v_pk_mul_f32") }
    if (MODE == 3) { P3("v_pk_fma_f32") P3("v_pk_fma_f32") }
    if (MODE == 4) { CVT CVT }
    if (MODE == 5) { asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[0])); asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[1]));
                     asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[2])); asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[3]));
                     asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[4])); asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[5]));
                     asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[6])); asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[7])); }
    if (MODE == 6) { V2("v_xor_b32") V2("v_xor_b32") }
    if (MODE == 7) { MF(0) MF(1) MF(2) MF(3) }
    if (MODE == 30) { MFAB(0, a, b) MFAB(1, a2, b) MFAB(2, a3, b) MFAB(3, a4, b) }            // distinct A, shared B
    if (MODE == 31) { MFAB(0, a, b) MFAB(1, a2, b2) MFAB(2, a3, b3) MFAB(3, a4, b4) }        // distinct A and B
    if (MODE == 32) { MFAB(0, a, b) MFAB(1, a, b2) MFAB(2, a2, b) MFAB(3, a2, b2) }          // 2x2 register tile
    if (MODE == 10) { MF(0) FMA1(0) FMA1(1) FMA1(2) FMA1(3) MF(1) FMA1(4) FMA1(5) FMA1(6) FMA1(7)
                      MF(2) FMA1(0) FMA1(1) FMA1(2) FMA1(3) MF(3) FMA1(4) FMA1(5) FMA1(6) FMA1(7) }
    if (MODE == 11) { MF(0) V3("v_fma_f32") MF(1) V3("v_fma_f32") MF(2) V3("v_fma_f32") MF(3) V3("v_fma_f32") }
    if (MODE == 12) { MF(0) EXP1(0) EXP1(1) EXP1(2) EXP1(3) MF(1) EXP1(4) EXP1(5) EXP1(6) EXP1(7)
                      MF(2) EXP1(0) EXP1(1) EXP1(2) EXP1(3) MF(3) EXP1(4) EXP1(5) EXP1(6) EXP1(7) }
    if (MODE == 13) { MF(0) EXP1(0) FMA1(1) FMA1(2) EXP1(3) FMA1(4) FMA1(5) MF(1) EXP1(6) FMA1(7) FMA1(0) EXP1(1) FMA1(2) FMA1(3)
                      MF(2) EXP1(4) FMA1(5) FMA1(6) EXP1(7) FMA1(0) FMA1(1) MF(3) EXP1(2) FMA1(3) FMA1(4) EXP1(5) FMA1(6) FMA1(7) }
    if (MODE == 20) { if (odd) { V3("v_fma_f32") V3("v_fma_f32") V3("v_fma_f32") V3("v_fma_f32") } else { MF(0) MF(1) MF(2) MF(3) } }
    if (MODE == 21) { if (odd) { V1("v_exp_f32") V1("v_exp_f32") } else { MF(0) MF(1) MF(2) MF(3) } }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += r[i] + q[i][0] + q[i][1] + (float)u[i];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) { clk[(blockIdx.x * 16 + wave) * 2] = c0; clk[(blockIdx.x * 16 + wave) * 2 + 1] = c1; }
}

template <int MODE>
static void run(const char* what, int waves, double valu_per_iter, double mfma_per_iter) {
  const int blocks = 256, iters = 4000;
  float* sink; unsigned long long* clk;
  (void)hipMalloc(&sink, 4); (void)hipMalloc(&clk, blocks * 32 * 8);
  (void)hipMemset(clk, 0, blocks * 32 * 8);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64 * waves), 0, 0, 100, sink, clk, 0.37f);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64 * waves), 0, 0, iters, sink, clk, 0.37f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[256 * 32];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  // per CU: first start to last end over its waves = the SIMDs' busy time (the arbiter is oldest-first, so the waves of
  // one SIMD need not progress at the same rate; per-wave durations are printed too)
  double span = 0, per_wave = 0;
  for (int bI = 0; bI < blocks; ++bI) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < waves; ++w) {
      const unsigned long long c0 = h[(bI * 16 + w) * 2], c1 = h[(bI * 16 + w) * 2 + 1];
      lo = c0 < lo ? c0 : lo; hi = c1 > hi ? c1 : hi;
      per_wave += (double)(c1 - c0) / iters / waves / blocks;
    }
    span += (double)(hi - lo) / iters / blocks;
  }
  const double wps = waves / 4.0;  // waves per SIMD
  const bool split = MODE >= 20 && MODE < 30;
  printf("%-40s %d waves/SIMD: %7.1f cycles/iter per SIMD (one wave: %7.1f)", what, waves / 4, span, per_wave);
  if (valu_per_iter > 0) printf("  %5.2f cycles per VALU instr", span / (valu_per_iter * (split ? wps / 2 : wps)));
  if (mfma_per_iter > 0) printf("  %5.1f cycles per MFMA", span / (mfma_per_iter * (split ? wps / 2 : wps)));
  printf("  [%.0f MHz]\n", span * iters / (ms * 1e3));
  (void)hipFree(sink); (void)hipFree(clk);
}

int main() {
  for (int waves : {4, 8, 12, 16}) {
    run<0>("v_exp_f32 x16", waves, 16, 0);
    run<1>("v_fma_f32 x16", waves, 16, 0);
    run<2>("v_pk_mul_f32 x16", waves, 16, 0);
    run<3>("v_pk_fma_f32 x16", waves, 16, 0);
    run<4>("v_cvt_pk_bf16_f32 x16", waves, 16, 0);
    run<5>("v_mul_lo_u32 x8", waves, 8, 0);
    run<6>("v_xor_b32 x16", waves, 16, 0);
    run<7>("mfma 32x32x16 x4", waves, 0, 4);
    run<31>("mfma x4, distinct A and B", waves, 0, 4);
    run<10>("same wave: 4 x (MFMA + 4 fma)", waves, 16, 4);
    run<11>("same wave: 4 x (MFMA + 8 fma)", waves, 32, 4);
    run<12>("same wave: 4 x (MFMA + 4 exp)", waves, 16, 4);
    run<13>("same wave: 4 x (MFMA + 2 exp + 4 fma)", waves, 24, 4);
    if (waves >= 8) {
      run<20>("split waves: 4 MFMA | 32 fma", waves, 32, 4);
      run<21>("split waves: 4 MFMA | 16 exp", waves, 16, 4);
    }
  }
  return 0;
}
