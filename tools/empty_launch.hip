#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(int* p) { __shared__ int s[40000]; if (p && threadIdx.x == 9999) { s[0] = 1; p[0] = s[1]; } }
int main() { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
 for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, nullptr);
 hipEventRecord(a, 0); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, nullptr); hipEventRecord(b, 0); hipEventSynchronize(b);
 float ms; hipEventElapsedTime(&ms, a, b); printf("empty 256x512 160KB-LDS kernel back-to-back: %.2f us each\n", ms * 10); return 0; }
