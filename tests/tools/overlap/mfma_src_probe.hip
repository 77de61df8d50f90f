// Does v_mfma_f32_16x16x32_bf16 issue at 16 cycles when its A operand (or B) comes from the AGPR half of the register file?  (round 6: the one-wave-per-SIMD
// block kernels keep filter and accumulators in AGPRs and run at ~25 cycles per MFMA; the 512-thread kernels - all VGPR - share a SIMD between two waves.)
// One wave per SIMD (256 threads, one workgroup per CU), 8 independent accumulators, 4096 MFMAs per wave; cycles by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 mfma_src_probe.hip -o mfma_src_probe && ./mfma_src_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE> __global__ void __launch_bounds__(256, 1) probe(unsigned long long* out, int iters) {
  // MODE 0: A v, B v, C/D a   1: A a, B v, C/D a   2: A v, B a, C/D a   3: A a, B a, C/D a   4: everything v
  unsigned long long t0 = 0, t1 = 0;
  asm volatile(
      "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n"
      "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
      "v_accvgpr_write_b32 a40, v10\n v_accvgpr_write_b32 a41, v10\n v_accvgpr_write_b32 a42, v10\n v_accvgpr_write_b32 a43, v10\n"
      "v_accvgpr_write_b32 a44, v10\n v_accvgpr_write_b32 a45, v10\n v_accvgpr_write_b32 a46, v10\n v_accvgpr_write_b32 a47, v10\n"
      ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
  t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#define MF(j) \
    if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%0+3], v[10:13], v[14:17], a[%0:%0+3]" :: "n"(4 * j) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"); \
    if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%0+3], a[40:43], v[14:17], a[%0:%0+3]" :: "n"(4 * j) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"); \
    if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%0+3], v[10:13], a[44:47], a[%0:%0+3]" :: "n"(4 * j) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"); \
    if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%0+3], a[40:43], a[44:47], a[%0:%0+3]" :: "n"(4 * j) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"); \
    if (MODE == 4) asm volatile("v_mfma_f32_16x16x32_bf16 v[%0:%0+3], v[10:13], v[14:17], v[%0:%0+3]" :: "n"(100 + 4 * j) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131");
    REP8(MF) REP8(MF)
  }
  t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 4 * 8);
  unsigned long long h[1024];
  const int iters = 256;
  const char* names[5] = {"A vgpr, B vgpr, C/D agpr", "A AGPR, B vgpr, C/D agpr", "A vgpr, B AGPR, C/D agpr", "A AGPR, B AGPR, C/D agpr", "A, B, C/D vgpr"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, d, iters);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, d, iters);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, d, iters);
      if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, d, iters);
      if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
    printf("%-28s %.2f cycles per MFMA\n", names[mode], s / 1024 / (iters * 16.0));
  }
  return 0;
}
