// L2 -> LDS DMA stream probe: replays the operand staging of the 128x384 NT GEMM tile (no MFMA, no
// LDS reads) to measure what the memory system delivers for that address pattern, and how the rate
// moves with the K order / operand mix.   hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint16_t bf16_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int HT = 128 * 64;
constexpr int RING = 10;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

struct Args {
  const bf16_t* A; const bf16_t* B;
  int lda, ldb, M, N, K;
  int nah, nbh;   // half-tiles of A / B per K-tile (1,3 = 128x384; 2,2 = 256x256)
  int mode;       // bit0: skip A, bit1: skip B, bit2: rotate K start per tile, bit3: no xcd remap
  int inflight;   // DMA instructions allowed in flight at the per-K-tile wait (each wave: 2 per half-tile)
};

template <int INF>
__global__ __launch_bounds__(512) void probe(Args p) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[RING * HT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int TM = p.nah * 128, TN = p.nbh * 128;
  const int nbn = p.N / TN;
  const int logical = (p.mode & 8) ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int bm = logical / nbn, bn = logical % nbn;
  const int nk = p.K >> 6;
  const int drow = lane >> 3;
  const int ch0 = ((lane & 7) ^ ((lane >> 4) & 7)) * 8;
  const int ch1 = ((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 8;
  const bf16_t* gA0 = p.A + (size_t)(bm * TM + (2 * uw) * 8 + drow) * p.lda + ch0;
  const bf16_t* gA1 = p.A + (size_t)(bm * TM + (2 * uw + 1) * 8 + drow) * p.lda + ch1;
  const bf16_t* gB0 = p.B + (size_t)(bn * TN + (2 * uw) * 8 + drow) * p.ldb + ch0;
  const bf16_t* gB1 = p.B + (size_t)(bn * TN + (2 * uw + 1) * 8 + drow) * p.ldb + ch1;
  const size_t hA = (size_t)128 * p.lda, hB = (size_t)128 * p.ldb;
  const int dst0 = (2 * uw) * 8 * 64, dst1 = (2 * uw + 1) * 8 * 64;
  const int rot = (p.mode & 4) ? (int)((logical * 5) % nk) : 0;
  int slot = 0;
  for (int t = 0; t < nk; ++t) {
    int kt = t + rot;
    kt = kt >= nk ? kt - nk : kt;
    if (!(p.mode & 1))
      for (int h = 0; h < p.nah; ++h) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gA0 + h * hA + (size_t)kt * 64), (lptr_t)&smem[slot * HT + dst0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gA1 + h * hA + (size_t)kt * 64), (lptr_t)&smem[slot * HT + dst1], 16, 0, 0);
        slot = slot + 1 == RING ? 0 : slot + 1;
      }
    if (!(p.mode & 2))
      for (int h = 0; h < p.nbh; ++h) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gB0 + h * hB + (size_t)kt * 64), (lptr_t)&smem[slot * HT + dst0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gB1 + h * hB + (size_t)kt * 64), (lptr_t)&smem[slot * HT + dst1], 16, 0, 0);
        slot = slot + 1 == RING ? 0 : slot + 1;
      }
    if (INF == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (INF == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (INF == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static void run(const char* name, Args a, int inflight, int iters) {
  const int TM = a.nah * 128, TN = a.nbh * 128;
  dim3 grid((a.M / TM) * (a.N / TN)), block(512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (inflight == 12) hipLaunchKernelGGL(probe<12>, grid, block, 0, 0, a);
    else if (inflight == 8) hipLaunchKernelGGL(probe<8>, grid, block, 0, 0, a);
    else if (inflight == 4) hipLaunchKernelGGL(probe<4>, grid, block, 0, 0, a);
    else hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, a);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const int halves = ((a.mode & 1) ? 0 : a.nah) + ((a.mode & 2) ? 0 : a.nbh);
  const double bytes = (double)grid.x * (a.K / 64) * halves * 16384.0;
  printf("%-34s M %6d N %5d K %5d tile %dx%d mode %2d inflight %2d grid %4u: %8.1f us  %6.1f GB/s/CU  %5.2f TB/s\n", name, a.M,
         a.N, a.K, TM, TN, a.mode, inflight, grid.x, ms * 1e3, bytes / (ms * 1e-3) / 1e9 / (grid.x < 256 ? grid.x : 256),
         bytes / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t elemsA = (size_t)16384 * 2304, elemsB = (size_t)2304 * 2304;
  bf16_t *A, *B;
  hipMalloc(&A, elemsA * 2);
  hipMalloc(&B, elemsB * 2);
  hipMemset(A, 0, elemsA * 2);
  hipMemset(B, 0, elemsB * 2);
  struct Shape { int M, N, K, nah, nbh; } shapes[] = {
      {16384, 768, 768, 1, 3}, {16384, 2304, 768, 1, 3}, {16384, 768, 2048, 1, 3}, {16384, 2048, 768, 2, 2}};
  for (auto s : shapes) {
    for (int mode : {0, 4, 1, 2, 8, 12}) {
      for (int inf : {12, 4}) {
        Args a{A, B, s.K, s.K, s.M, s.N, s.K, s.nah, s.nbh, mode, inf};
        run("stage-only", a, inf, 20);
      }
    }
  }
  hipFree(A); hipFree(B);
  return 0;
}
