// Round 3, fp8 go / no-go: lane maps and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 and of the
// scaled f32 <-> fp8 conversions on gfx950, established with exact integer data before any kernel relies on them.
//   hipcc --offload-arch=gfx950 -O2 fp8_probe.hip -o fp8_probe && ./fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));

__host__ __device__ static float e4m3_to_f(uint8_t v) {      // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}
static uint8_t f_to_e4m3_exact(float f) {                   // exact values only (small integers)
  for (int v = 0; v < 256; ++v) if (e4m3_to_f((uint8_t)v) == f) return (uint8_t)v;
  fprintf(stderr, "not representable: %f\n", f); exit(1);
}

// A: [16][128] bytes (row-major), B: [128][16] bytes (k-major), per hypothesis `hyp` the (lane, byte) -> k map
__device__ int kmap(int hyp, int g, int byte) {
  if (hyp == 0) return 32 * g + byte;
  if (hyp == 1) return byte < 16 ? 16 * g + byte : 64 + 16 * g + (byte - 16);
  return 8 * g + (byte & 7) + 32 * (byte >> 3);
}
__global__ void mm(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* C, int hyp) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  union { v8i v; uint8_t b[32]; } a, b;
  for (int j = 0; j < 32; ++j) { const int k = kmap(hyp, g, j); a.b[j] = A[r * 128 + k]; b.b[j] = B[k * 16 + r]; }
  // scale of (row r, k block g) / (k block g, col r) in byte 0 of the scale registers
  const int sa = SA[r * 4 + g], sb = SB[g * 16 + r];
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, sa, 0, sb);
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = c[j];
}
__global__ void cvt(const float* in, float scale, int* packed, float* back) {
  const int i = threadIdx.x;
  v2s p = {0, 0};
  p = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(p, in[2 * i], in[2 * i + 1], scale, false);
  const int pi = __builtin_bit_cast(int, p);
  packed[i] = pi;
  v2f b = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(pi, scale, false);
  back[2 * i] = b[0]; back[2 * i + 1] = b[1];
}

int main() {
  uint8_t hA[16 * 128], hB[128 * 16], hSA[64], hSB[64];
  float fA[16 * 128], fB[128 * 16];
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) { fA[i] = (float)(rand() % 7 - 3); hA[i] = f_to_e4m3_exact(fA[i]); }
  for (int i = 0; i < 128 * 16; ++i) { fB[i] = (float)(rand() % 5 - 2); hB[i] = f_to_e4m3_exact(fB[i]); }
  uint8_t *dA, *dB, *dSA, *dSB; float* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: all scales 127 (2^0); pass 1: per-block scales 125 .. 130
    for (int i = 0; i < 64; ++i) { hSA[i] = pass ? 125 + (i * 5) % 6 : 127; hSB[i] = pass ? 125 + (i * 7 + 3) % 6 : 127; }
    hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 3; ++hyp) {
      mm<<<1, 64>>>(dA, dB, dSA, dSB, dC, hyp);
      float hC[256];
      hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
      int bad = 0; double worst = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)fA[i * 128 + k] * ldexp(1.0, hSA[i * 4 + k / 32] - 127) * (double)fB[k * 16 + j] * ldexp(1.0, hSB[(k / 32) * 16 + j] - 127);
        const double d = fabs(ref - hC[i * 16 + j]);
        if (d > 1e-3) ++bad;
        if (d > worst) worst = d;
      }
      printf("pass %d (scales %s) hypothesis %d: %d of 256 wrong, worst |diff| %.4f\n", pass, pass ? "per block" : "unit", hyp, bad, worst);
    }
  }
  // conversions: what does `scale` do?
  float hin[128], *din, *dback; int* dp;
  for (int i = 0; i < 128; ++i) hin[i] = (i - 64) * 0.37f;
  hipMalloc(&din, sizeof hin); hipMalloc(&dback, sizeof hin); hipMalloc(&dp, 64 * 4);
  hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
  for (float sc : {1.0f, 4.0f, 0.25f}) {
    cvt<<<1, 64>>>(din, sc, dp, dback);
    int hp[64]; float hb[128];
    hipMemcpy(hp, dp, sizeof hp, hipMemcpyDeviceToHost); hipMemcpy(hb, dback, sizeof hb, hipMemcpyDeviceToHost);
    printf("scale %.2f:", sc);
    for (int i : {0, 40, 70, 100, 127}) {
      const uint8_t byte = (hp[i / 2] >> (8 * (i & 1))) & 0xff;
      printf("  in %.3f -> fp8 0x%02x (= %.4f) -> back %.4f", hin[i], byte, e4m3_to_f(byte), hb[i]);
    }
    printf("\n");
  }
  return 0;
}
