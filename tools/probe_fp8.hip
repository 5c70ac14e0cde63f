// Operand lane maps of the block-scaled fp8 MFMA on gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 / e5m2
// operands and unit E8M0 scales), checked with exact small-integer data, plus the f32 -> fp8 conversion builtins.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_fp8.hip -o gpurun_out/probe_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// e4m3 (OCP) encode of small integers / powers of two (exact)
static unsigned char e4m3(float f) {
  if (f == 0) return 0;
  unsigned s = f < 0 ? 0x80 : 0;
  f = fabsf(f);
  int e; float m = frexpf(f, &e);  // f = m * 2^e, m in [0.5,1)
  int E = e - 1 + 7;               // biased exponent of 1.xxx * 2^(e-1)
  int man = (int)roundf((m * 2 - 1) * 8);
  if (man == 8) { man = 0; E++; }
  if (E <= 0) { man = (int)roundf(f / ldexpf(1.f, -9)); return (unsigned char)(s | man); }
  return (unsigned char)(s | (E << 3) | man);
}
static unsigned char e5m2(float f) {
  if (f == 0) return 0;
  unsigned s = f < 0 ? 0x80 : 0;
  f = fabsf(f);
  int e; float m = frexpf(f, &e);
  int E = e - 1 + 15;
  int man = (int)roundf((m * 2 - 1) * 4);
  if (man == 4) { man = 0; E++; }
  return (unsigned char)(s | (E << 2) | man);
}

// mode: 0 A e4m3 x B e4m3 ; 1 A e5m2 x B e4m3.  A row-major [16][128] bytes, Bt row-major [16][128] (= B^T).
// hyp 0: lane l holds k = 32*(l>>4) + j          (j = byte 0..31 of its 8 VGPRs)
// hyp 1: lane l holds k = 16*(l>>4) + (j&15) + 64*(j>>4)
template <int MODE>
__global__ void k_mfma(const unsigned char* A, const unsigned char* Bt, float* C, int hyp, int swap) {
  int l = threadIdx.x;
  unsigned char ab[32], bb[32];
  for (int j = 0; j < 32; ++j) {
    int k = hyp == 0 ? 32 * (l >> 4) + j : 16 * (l >> 4) + (j & 15) + 64 * (j >> 4);
    ab[j] = A[(l & 15) * 128 + k];
    bb[j] = Bt[(l & 15) * 128 + k];
  }
  i32x8 a, b;
  memcpy(&a, ab, 32);
  memcpy(&b, bb, 32);
  f32x4 c = {0, 0, 0, 0};
  const int one = 0x7F7F7F7F;  // E8M0 127 = 2^0 in every byte
  if (!swap) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, MODE == 1 ? 1 : 0, 0, 0, one, 0, one);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, MODE == 1 ? 1 : 0, 0, one, 0, one);
  // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int j = 0; j < 4; ++j) C[((l >> 4) * 4 + j) * 16 + (l & 15)] = c[j];
}

__global__ void k_cvt(const float* x, unsigned* out_e4, unsigned* out_e5, int n) {
  int i = threadIdx.x;
  if (i * 4 + 3 < n) {
    unsigned v = 0, w = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], v, true);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(x[4 * i], x[4 * i + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(x[4 * i + 2], x[4 * i + 3], w, true);
    out_e4[i] = v; out_e5[i] = w;
  }
}

int main() {
  std::vector<float> Af(16 * 128), Bf(16 * 128);
  srand(3);
  for (auto& v : Af) v = (float)(rand() % 9 - 4);          // -4..4 exact in both formats
  for (auto& v : Bf) v = (float)(rand() % 7 - 3) * 0.5f;   // asymmetric with A
  unsigned char *dA, *dB; float* dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<unsigned char> A8(2048), B8(2048);
    for (int i = 0; i < 2048; ++i) { A8[i] = mode ? e5m2(Af[i]) : e4m3(Af[i]); B8[i] = e4m3(Bf[i]); }
    hipMemcpy(dA, A8.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dB, B8.data(), 2048, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp)
      for (int swap = 0; swap < 2; ++swap) {
        if (mode == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp, swap);
        else hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp, swap);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            float ref = 0;
            for (int k = 0; k < 128; ++k) ref += Af[i * 128 + k] * Bf[j * 128 + k];
            // not swapped: D[i][j] = sum_k A[i][k] B[k][j]; swapped operands give D^T
            float got = swap ? C[j * 16 + i] : C[i * 16 + j];
            if (got != ref) ++bad;
          }
        printf("mode %d (A %s) hyp %d swap %d: %s (%d wrong)\n", mode, mode ? "e5m2" : "e4m3", hyp, swap, bad ? "FAIL" : "PASS", bad);
      }
  }
  // conversions: round-to-nearest-even, saturation, byte order
  float xs[16] = {0.f, 1.f, -1.f, 448.f, 1000.f, -1000.f, 0.0625f, 3.3f, 1e-4f, 57344.f, 1e6f, 0.3f, 17.f, 19.f, 0.0019f, -0.f};
  float* dx; unsigned *d4, *d5;
  hipMalloc(&dx, 64); hipMalloc(&d4, 16); hipMalloc(&d5, 16);
  hipMemcpy(dx, xs, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cvt, dim3(1), dim3(4), 0, 0, dx, d4, d5, 16);
  unsigned o4[4], o5[4];
  hipMemcpy(o4, d4, 16, hipMemcpyDeviceToHost); hipMemcpy(o5, d5, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i)
    printf("x %12g -> e4m3 0x%02x  e5m2 0x%02x\n", xs[i], (o4[i / 4] >> (8 * (i % 4))) & 0xFF, (o5[i / 4] >> (8 * (i % 4))) & 0xFF);
  return 0;
}
