// Hardware-assumption probes for gfx950: MFMA operand/accumulator lane maps, ds_read_b64_tr_b16
// gather semantics and the accumulator-as-operand k permutation.  Prints PASS/FAIL per probe.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_hw.hip -o gpurun_out/probe_hw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static inline float bf2f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

// C[16x16] = A[16x32] * B[32x16]; A row-major [16][32], Bt row-major [16][32] (= B^T)
__global__ void k_mfma16(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 15) * 32 + 8 * (l >> 4));
  bf16x8 b = *(const bf16x8*)(Bt + (l & 15) * 32 + 8 * (l >> 4));
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];
}
// C[32x32] = A[32x16] * B[16x32]; A [32][16], Bt [32][16]
__global__ void k_mfma32(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  bf16x8 a = *(const bf16x8*)(A + (l & 31) * 16 + 8 * (l >> 5));
  bf16x8 b = *(const bf16x8*)(Bt + (l & 31) * 16 + 8 * (l >> 5));
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// LDS holds M[R=16][Cc=64] u16 with value r*64+c. Each 16-lane group g reads the 4x16 block at
// rows 4g..4g+3, cols 16g'.. where g' = g (so blocks differ per group). Lane 4q+p supplies &M[r0+q][c0+4p].
__global__ void k_tr(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short M[16 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) M[i] = (unsigned short)i;
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  int r0 = 4 * g, c0 = 16 * g;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(M + (r0 + q) * 64 + c0 + 4 * p));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
// accumulator-as-operand: X^T[32 keys][32 q] = K[32x16]·Q^T ; then O[32 q][32 dv] = X^T^T · V  (sum over keys)
// K [32][16], Q [32][16], V [32 keys][32 dv] all row-major bf16. O fp32 [32][32] row-major.
__global__ void k_acc_operand(const unsigned short* K, const unsigned short* Q, const unsigned short* V, float* O) {
  __shared__ __attribute__((aligned(16))) unsigned short Vs[32 * 32];
  int l = threadIdx.x;
  for (int i = l; i < 32 * 32; i += 64) Vs[i] = V[i];
  __syncthreads();
  bf16x8 a = *(const bf16x8*)(K + (l & 31) * 16 + 8 * (l >> 5));
  bf16x8 b = *(const bf16x8*)(Q + (l & 31) * 16 + 8 * (l >> 5));
  f32x16 x;
  for (int i = 0; i < 16; ++i) x[i] = 0;
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);   // x: col q = l&31, rows = keys
  f32x16 o;
  for (int i = 0; i < 16; ++i) o[i] = 0;
  int h = l >> 5, g = l >> 4, li = l & 15, qq = li >> 2, p = li & 3;
  for (int s = 0; s < 2; ++s) {
    bf16x8 pa;
    for (int j = 0; j < 8; ++j) {   // values here are small integers: exact in bf16
      float f = x[8 * s + j];
      unsigned u = __float_as_uint(f);
      pa[j] = (short)(u >> 16);
    }
    // B operand: lane (col dv = l&31, half h) element j <- V[key = 16s + 8(j>>2) + 4h + (j&3)][dv]
    int c0 = 16 * (g & 1);
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Vs + (16 * s + 4 * h + qq) * 32 + c0 + 4 * p));
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Vs + (16 * s + 8 + 4 * h + qq) * 32 + c0 + 4 * p));
    bf16x8 vb = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, o, 0, 0, 0);  // A = X^T (rows q), B = V -> O[q][dv]
  }
  for (int r = 0; r < 16; ++r) O[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = o[r];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

int main() {
  int fails = 0;
  srand(1);
  {  // mfma16
    std::vector<unsigned short> A(16 * 32), Bt(16 * 32);
    std::vector<float> C(256), R(256, 0.f);
    for (auto& v : A) v = f2bf((float)(rand() % 7 - 3));
    for (auto& v : Bt) v = f2bf((float)(rand() % 5 - 2));
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += bf2f(A[i * 32 + k]) * bf2f(Bt[j * 32 + k]);
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("probe mfma_16x16x32 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // mfma32
    std::vector<unsigned short> A(32 * 16), Bt(32 * 16);
    std::vector<float> C(1024), R(1024, 0.f);
    for (auto& v : A) v = f2bf((float)(rand() % 7 - 3));
    for (auto& v : Bt) v = f2bf((float)(rand() % 5 - 2));
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += bf2f(A[i * 16 + k]) * bf2f(Bt[j * 16 + k]);
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("probe mfma_32x32x16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // tr read
    std::vector<unsigned short> out(256);
    unsigned short* d; CK(hipMalloc(&d, 512));
    k_tr<<<1, 64>>>(d); CK(hipMemcpy(out.data(), d, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15;
      for (int j = 0; j < 4; ++j) { int want = (4 * g + j) * 64 + 16 * g + i; bad += out[l * 4 + j] != want; }
    }
    printf("probe ds_read_b64_tr_b16: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, out[l*4], out[l*4+1], out[l*4+2], out[l*4+3]);
  }
  {  // accumulator as operand
    std::vector<unsigned short> K(32 * 16), Q(32 * 16), V(32 * 32);
    std::vector<float> O(1024), R(1024, 0.f), X(1024, 0.f);
    for (auto& v : K) v = f2bf((float)(rand() % 3 - 1));
    for (auto& v : Q) v = f2bf((float)(rand() % 3 - 1));
    for (auto& v : V) v = f2bf((float)(rand() % 5 - 2));
    for (int key = 0; key < 32; ++key) for (int q = 0; q < 32; ++q) for (int k = 0; k < 16; ++k) X[key * 32 + q] += bf2f(K[key * 16 + k]) * bf2f(Q[q * 16 + k]);
    for (int q = 0; q < 32; ++q) for (int dv = 0; dv < 32; ++dv) for (int key = 0; key < 32; ++key) R[q * 32 + dv] += X[key * 32 + q] * bf2f(V[key * 32 + dv]);
    unsigned short *dK, *dQ, *dV; float* dO;
    CK(hipMalloc(&dK, 1024)); CK(hipMalloc(&dQ, 1024)); CK(hipMalloc(&dV, 2048)); CK(hipMalloc(&dO, 4096));
    CK(hipMemcpy(dK, K.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dQ, Q.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dV, V.data(), 2048, hipMemcpyHostToDevice));
    k_acc_operand<<<1, 64>>>(dK, dQ, dV, dO); CK(hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += O[i] != R[i];
    printf("probe acc-as-A-operand + tr-read B: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d LDS/block %zu clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock, prop.clockRate);
  return fails ? 1 : 0;
}
