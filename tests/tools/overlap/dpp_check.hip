#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
  int i = __float_as_int(v), r;
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(i, i, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    default: r = __builtin_amdgcn_update_dpp(i, i, 0x140, 0xF, 0xF, false); break; // row_mirror
  }
  return __int_as_float(r);
}
__device__ __forceinline__ float row_sum16(float t) {
  t += dpp_f(t, 0); t += dpp_f(t, 1); t += dpp_f(t, 2); t += dpp_f(t, 3);
  return t;
}
__device__ __forceinline__ float ref16(float t) { t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8); return t; }
__global__ void k(const float* in, float* out) {
  const float v = in[threadIdx.x];
  float a = row_sum16(v), b = ref16(v);
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  float a32 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  float b32 = b + __shfl_xor(b, 16);
  const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a32), __float_as_uint(a32), false, false);
  float a64 = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  float b64 = b32 + __shfl_xor(b32, 32);
  // partner values (not sums): lane ^ 16 and lane ^ 32
  const int g = threadIdx.x >> 4;
  float p16 = (g & 1) ? __uint_as_float(r[0]) : __uint_as_float(r[1]), q16 = __shfl_xor(a, 16);
  float p32 = (threadIdx.x & 32) ? __uint_as_float(r2[0]) : __uint_as_float(r2[1]), q32 = __shfl_xor(a32, 32);
  out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = a32; out[192 + threadIdx.x] = b32;
  out[256 + threadIdx.x] = a64; out[320 + threadIdx.x] = b64; out[384 + threadIdx.x] = p16; out[448 + threadIdx.x] = q16;
  out[512 + threadIdx.x] = p32; out[576 + threadIdx.x] = q32;
}
int main() {
  float h[64], o[640]; srand(3);
  for (int i = 0; i < 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.7f;
  float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 2560);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 2560, hipMemcpyDeviceToHost);
  int bad16 = 0, bad32 = 0;
  for (int i = 0; i < 64; ++i) { if (memcmp(&o[i], &o[64 + i], 4)) ++bad16; if (memcmp(&o[128 + i], &o[192 + i], 4)) ++bad32; }
  int bad64 = 0, badp16 = 0, badp32 = 0;
  for (int i = 0; i < 64; ++i) { if (memcmp(&o[256 + i], &o[320 + i], 4)) ++bad64; if (memcmp(&o[384 + i], &o[448 + i], 4)) ++badp16; if (memcmp(&o[512 + i], &o[576 + i], 4)) ++badp32; }
  printf("wave64 sum mismatches %d, lane^16 partner mismatches %d, lane^32 partner mismatches %d\n", bad64, badp16, badp32);
  printf("row16 mismatches %d, xor16 mismatches %d (%.9g %.9g | %.9g %.9g)\n", bad16, bad32, o[0], o[64], o[128], o[192]);
  return 0;
}
