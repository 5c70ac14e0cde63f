// Semantics of gfx950's scaled fp8 conversions, measured on the hardware (the ISA text is not in this image):
//   v_cvt_scalef32_pk_fp8_bf16 / _pk_bf8_bf16 (2 packed bf16 -> 2 fp8) and v_cvt_scalef32_pk_fp8_f32 / _pk_bf8_f32,
// each with an f32 "scale" operand. Questions: is the result x * scale, x / scale, or x / 2^exponent(scale)? Does the
// mantissa of the scale matter? Do out-of-range values saturate to the largest finite value or become NaN / inf (the
// unscaled v_cvt_pk_fp8_f32 does NOT saturate: tools/probe_fp8.hip)? Round to nearest even?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_scale.hip -o gpurun_out/probe_cvt && gpurun_out/probe_cvt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* x, const float* scale, int n, int ns, uint8_t* out) {
  // out[(form * ns + si) * n + i]: form 0 fp8<-bf16, 1 bf8<-bf16, 2 fp8<-f32, 3 bf8<-f32
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= n) return;
  for (int si = 0; si < ns; ++si) {
    const float s = scale[si];
    const float v = x[i];
    bf16x2 b = {(__bf16)v, (__bf16)v};
    s16x2 old = {0, 0};
    s16x2 r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, b, s, false);
    s16x2 r1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(old, b, s, false);
    s16x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, v, v, s, false);
    s16x2 r3 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(old, v, v, s, false);
    out[(0 * ns + si) * n + i] = (uint8_t)(r0[0] & 0xff);
    out[(1 * ns + si) * n + i] = (uint8_t)(r1[0] & 0xff);
    out[(2 * ns + si) * n + i] = (uint8_t)(r2[0] & 0xff);
    out[(3 * ns + si) * n + i] = (uint8_t)(r3[0] & 0xff);
  }
}

static float dec_e4m3(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) v = ldexpf((float)m, -9); else v = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}
static float dec_e5m2(uint8_t b) {
  const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v;
  if (e == 31) return m ? NAN : (s ? -INFINITY : INFINITY);
  if (e == 0) v = ldexpf((float)m, -16); else v = ldexpf(1.f + m / 4.f, e - 15);
  return s ? -v : v;
}

int main() {
  std::vector<float> xs = {0.f, 1.f, 1.0625f, 1.1875f, 1.5f, 3.f, 100.f, 448.f, 464.f, 480.f, 1000.f, 1e6f, -1000.f, 57344.f, 61440.f, 65536.f,
                           1e9f, 0.001953125f, 0.0009765625f, 0.0014f, 1.52587890625e-05f, 7e-6f, INFINITY, NAN, -0.f, 0.3f, 17.3f};
  std::vector<float> sc = {1.f, 2.f, 0.5f, 3.f, 1.5f, 0.125f, 1.9999f, 8.f};
  const int n = (int)xs.size(), ns = (int)sc.size();
  float *dx, *ds; uint8_t* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, ns * 4); hipMalloc(&dout, 4 * ns * n);
  hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(ds, sc.data(), ns * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, ds, n, ns, dout);
  std::vector<uint8_t> out(4 * ns * n);
  if (hipMemcpy(out.data(), dout, out.size(), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  const char* names[4] = {"fp8<-bf16", "bf8<-bf16", "fp8<-f32", "bf8<-f32"};
  for (int f = 0; f < 4; ++f) {
    printf("=== %s: rows = input, columns = scale operand; entries = decoded result (raw byte)\n%14s", names[f], "x \\ scale");
    for (int si = 0; si < ns; ++si) printf(" %16g", sc[si]);
    printf("\n");
    for (int i = 0; i < n; ++i) {
      printf("%14g", xs[i]);
      for (int si = 0; si < ns; ++si) {
        const uint8_t b = out[(f * ns + si) * n + i];
        const float v = (f & 1) ? dec_e5m2(b) : dec_e4m3(b);
        printf(" %11g (0x%02x)", v, b);
      }
      printf("\n");
    }
  }
  return 0;
}
