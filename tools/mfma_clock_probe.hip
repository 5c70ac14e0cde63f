// What does the MFMA pipe sustain on real data? Runs back-to-back v_mfma_f32_16x16x32_bf16 from registers
// (no memory traffic) on every CU, 2 waves per SIMD, with (a) zero operands and (b) random operands, and
// reports the achieved TFLOP/s and the shader clock during the run (s_memtime cycles / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(const uint32_t* seed, int iters, float* sink, unsigned long long* clk) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      const uint32_t s = seed[(tid * 4 + i) * 8 + j];
      a[i][j] = (short)(s & 0xFFFF);
      b[i][j] = (short)(s >> 16);
    }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// same operand bytes, the 32x32x16 shape: 4 accumulators of 16 registers, 16 MFMAs per iteration = the same FLOPs
__global__ __launch_bounds__(512) void mfma_loop32(const uint32_t* seed, int iters, float* sink, unsigned long long* clk) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      const uint32_t s = seed[(tid * 4 + i) * 8 + j];
      a[i][j] = (short)(s & 0xFFFF);
      b[i][j] = (short)(s >> 16);
    }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i & 3], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) sink[0] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks = 256, iters = 20000;
  uint32_t* h = (uint32_t*)malloc(512 * 32 * 4);
  uint32_t *dz, *dr; float* sink; unsigned long long* clk;
  hipMalloc(&dz, 512 * 32 * 4); hipMalloc(&dr, 512 * 32 * 4); hipMalloc(&sink, 4); hipMalloc(&clk, blocks * 16);
  hipMemset(dz, 0, 512 * 32 * 4);
  srand(1);
  for (int i = 0; i < 512 * 32; ++i) {
    // two random bf16 in [-2, 2): sign + exponent 0x3F/0x40 region + random mantissa
    const uint32_t lo = (uint32_t)(rand() & 0x80FF) | 0x3F00 | ((rand() & 1) << 7);
    const uint32_t hi = (uint32_t)(rand() & 0x80FF) | 0x3F00 | ((rand() & 1) << 7);
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(dr, h, 512 * 32 * 4, hipMemcpyHostToDevice);
  unsigned long long hc[2 * blocks];
  for (int pass = 0; pass < 8; ++pass) {
    const bool rnd = pass & 1;
    const bool big = pass & 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, rnd ? dr : dz, 100, sink, clk);
    hipEventRecord(e0, 0);
    if (big) hipLaunchKernelGGL(mfma_loop32, dim3(blocks), dim3(512), 0, 0, rnd ? dr : dz, iters, sink, clk);
    else hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, rnd ? dr : dz, iters, sink, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
    double mhz = 0;
    for (int i = 0; i < blocks; ++i) mhz += (double)hc[2 * i] / ((double)hc[2 * i + 1] / 100.0);  // realtime = 100 MHz
    mhz /= blocks;
    const double flops = (double)blocks * 8 * iters * 16 * 16384.0;
    printf("%s %-7s operands: %8.1f us  %7.1f TFLOP/s  shader clock %.0f MHz (cycle counter / 100 MHz real time)\n",
           big ? "32x32x16" : "16x16x32", rnd ? "random" : "zero", ms * 1e3, flops / (ms * 1e-3) / 1e12, mhz);
  }
  return 0;
}
