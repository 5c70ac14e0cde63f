// gfx950 probes for the fp8 token-major (weight-gradient) GEMM:
//  (1) ds_read_b64_tr_b8: which LDS byte lands in (lane, byte j) when lane l supplies the 8-byte address A(l);
//  (2) a 16x16x128 block-scaled fp8 MFMA fed by transposed reads of two TOKEN-MAJOR images
//      (D[n][k] = sum_t A[t][n] * B[t][k], t = 0..127), checked against a host sum.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr8.hip -o gpurun_out/probe_tr8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i32x2 lds_i32x2;

// LDS = 2048 bytes; byte i holds (i & 255) in pass 0 and (i >> 8) in pass 1. Lane l reads address addr[l].
__global__ void k_tr8_dump(const int* addr, unsigned char* out, int pass) {
  __shared__ __attribute__((aligned(16))) unsigned char M[2048];
  const int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) M[i] = pass ? (unsigned char)(i >> 8) : (unsigned char)(i & 255);
  __syncthreads();
  const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(M + addr[l]));
  unsigned int w[2] = {(unsigned)v[0], (unsigned)v[1]};
  for (int j = 0; j < 8; ++j) out[l * 8 + j] = (unsigned char)(w[j >> 2] >> (8 * (j & 3)));
}

// Token-major images At[128 t][16 n], Bt[128 t][16 k] (fp8 e4m3 bytes, row stride 16 B). Hypothesis under test (from the
// 16-bit form: per 16-lane group a block of 8-byte row pieces is delivered column-major): lane 2q + p of a 16-lane group
// supplies the address of row q (0..7), columns 8p..8p+7, and lane i of the group receives column i's 8 rows.
// One fragment = 32 t per lane = 4 reads; lane group g = lane >> 4 takes t in [32 g, 32 g + 32): read r covers
// t = 32 g + 8 r .. + 7. A and B use the same assignment, so the k order inside the MFMA does not matter.
__global__ void k_tn_fp8(const unsigned char* At, const unsigned char* Bt, float* D, int variant) {
  __shared__ __attribute__((aligned(16))) unsigned char sA[128 * 16], sB[128 * 16];
  const int l = threadIdx.x;
  for (int i = l; i < 128 * 16; i += 64) { sA[i] = At[i]; sB[i] = Bt[i]; }
  __syncthreads();
  const int g = l >> 4, i = l & 15;
  int q, p;
  if (variant == 0) { q = i >> 1; p = i & 1; }        // lane 2q + p: row q, 8-byte column piece p
  else { q = i & 7; p = i >> 3; }                      // lane 8p + q
  i32x8 a, b;
  for (int r = 0; r < 4; ++r) {
    const int t = 32 * g + 8 * r + q;
    const i32x2 va = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(sA + t * 16 + 8 * p));
    const i32x2 vb = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(sB + t * 16 + 8 * p));
    a[2 * r] = va[0]; a[2 * r + 1] = va[1];
    b[2 * r] = vb[0]; b[2 * r + 1] = vb[1];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  // D[row = from first operand's lane&15][col = second operand's lane&15]: c[j] of lane l = D[4 (l >> 4) + j][l & 15]
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int j = 0; j < 4; ++j) D[(4 * g + j) * 16 + i] = c[j];
}

static float e4m3_to_float(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

int main() {
  // ---- (1) dump
  int h_addr[64];
  unsigned char *d_out, h0[512], h1[512];
  int* d_addr;
  hipMalloc(&d_addr, sizeof(h_addr));
  hipMalloc(&d_out, 512);
  const char* names[3] = {"addr = 8*lane", "addr = 128*(lane&15) + 8*(lane>>4)  (row = lane&15, piece = lane>>4)",
                          "addr = 16*(lane>>1) + 8*(lane&1)   (16-byte rows: row = lane>>1, piece = lane&1)"};
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l)
      h_addr[l] = pat == 0 ? 8 * l : pat == 1 ? 128 * (l & 15) + 8 * (l >> 4) : 16 * (l >> 1) + 8 * (l & 1);
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tr8_dump, dim3(1), dim3(64), 0, 0, d_addr, d_out, 0);
    hipMemcpy(h0, d_out, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_tr8_dump, dim3(1), dim3(64), 0, 0, d_addr, d_out, 1);
    hipMemcpy(h1, d_out, 512, hipMemcpyDeviceToHost);
    printf("== ds_read_b64_tr_b8, %s: (lane: source byte index of bytes 0..7 | as (supplying lane, byte in its piece))\n", names[pat]);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 8; ++j) printf(" %4d", h0[l * 8 + j] + 256 * h1[l * 8 + j]);
      printf("  |");
      for (int j = 0; j < 8; ++j) {
        const int src = h0[l * 8 + j] + 256 * h1[l * 8 + j];
        int sl = -1, sb = -1;
        for (int m = 0; m < 64; ++m)
          if (src >= h_addr[m] && src < h_addr[m] + 8) { sl = m; sb = src - h_addr[m]; }
        printf(" (%2d,%d)", sl, sb);
      }
      printf("\n");
    }
  }
  // ---- (2) microkernel
  std::vector<unsigned char> At(128 * 16), Bt(128 * 16);
  srand(7);
  for (auto& v : At) { v = (unsigned char)(rand() & 0xFF); if ((v & 0x7F) == 0x7F) v &= 0xFE; if (((v >> 3) & 15) > 9) v &= 0xBF; }
  for (auto& v : Bt) { v = (unsigned char)(rand() & 0xFF); if ((v & 0x7F) == 0x7F) v &= 0xFE; if (((v >> 3) & 15) > 9) v &= 0xBF; }
  unsigned char *dA, *dB;
  float* dD;
  hipMalloc(&dA, At.size()); hipMalloc(&dB, Bt.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, At.data(), At.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
  std::vector<double> ref(256, 0.0);
  for (int n = 0; n < 16; ++n)
    for (int k = 0; k < 16; ++k)
      for (int t = 0; t < 128; ++t) ref[n * 16 + k] += (double)e4m3_to_float(At[t * 16 + n]) * (double)e4m3_to_float(Bt[t * 16 + k]);
  for (int variant = 0; variant < 2; ++variant) {
    float hD[256];
    hipLaunchKernelGGL(k_tn_fp8, dim3(1), dim3(64), 0, 0, dA, dB, dD, variant);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double e_nk = 0, e_kn = 0, mag = 0;
    for (int n = 0; n < 16; ++n)
      for (int k = 0; k < 16; ++k) {
        e_nk = fmax(e_nk, fabs(hD[n * 16 + k] - ref[n * 16 + k]));
        e_kn = fmax(e_kn, fabs(hD[k * 16 + n] - ref[n * 16 + k]));
        mag = fmax(mag, fabs(ref[n * 16 + k]));
      }
    printf("tn fp8 microkernel, variant %d: max |D[n][k] - ref| = %.4g, max |D[k][n] - ref| = %.4g (|ref| up to %.4g) -> %s\n", variant,
           e_nk, e_kn, mag, (e_nk < 1e-3 * mag) ? "D[row=A][col=B] PASS" : (e_kn < 1e-3 * mag) ? "D[row=B][col=A] PASS" : "FAIL");
  }
  return 0;
}
