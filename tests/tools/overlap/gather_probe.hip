// How fast can 256 workgroups pull the PixelShuffle^T gather of conv4d_kernel (csrc/conv_dgrad4.hip) out of HBM with LDS-DMA, and does it matter
// that a stage reads ONE shuffle phase - 128-byte pieces at a 256-byte stride - instead of both phases of a row (256 contiguous bytes per pixel,
// 4.6 KB contiguous per halo row)?  No MFMAs, no fragment reads: only the DMA issue, the counted waits and the stage hand-over of that kernel.
//   pattern A: stage = (tile, dy, dx): 10 x 18 halo pixels x 128 B, 23 pieces of 8 pixels, ring of 6 stages, 4 ahead          (the product)
//   pattern B: stage = (tile, dy): both dx phases, 10 x 18 x 256 B, 45 pieces of 4 pixels, ring of 3 stages, 2 ahead           (same bytes in flight)
// input [32][192][192][64] bf16 (151 MB), 2304 tiles of 8 x 16 LR pixels, 9 per workgroup.
// build + run (GPU box): make -C tests/tools/overlap gather_probe && tests/tools/overlap/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int TH = 8, TW = 16, HW_ = 18, HPIX = 180;
__device__ __attribute__((aligned(256))) uint4 zero_page[16];

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int PAT>
__global__ void __launch_bounds__(256, 1) gather(const uint16_t* x, int N, int H, int W, int tiles_x, int tiles_y, unsigned* sink) {
  constexpr int PIECES = PAT == 0 ? 23 : 45, PW = (PIECES + 3) / 4, STAGE = PIECES * 1024, NST = PAT == 0 ? 6 : 3, AHEAD = PAT == 0 ? 4 : 2;
  constexpr int SPT = PAT == 0 ? 4 : 2;                     // stages per tile
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = N * tiles_x * tiles_y, tile0 = blockIdx.x, stride = gridDim.x;
  if (tile0 >= ntiles) return;
  const int nt = (ntiles - tile0 + stride - 1) / stride, nstage = SPT * nt;
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto issue = [&](int u) {
    const int tile = tile0 + (u / SPT) * stride, ph = u % SPT;
    const int n = tile / (tiles_x * tiles_y), rem = tile - n * tiles_x * tiles_y, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const unsigned dst = ring + (unsigned)(u % NST) * STAGE;
#pragma unroll
    for (int k = 0; k < PW; ++k) {
      const int piece = (q + 4 * k < PIECES) ? q + 4 * k : PIECES - 1;
      int pix, dy, dx, slot;
      if (PAT == 0) { pix = piece * 8 + (lane >> 3); dy = ph >> 1; dx = ph & 1; slot = lane & 7; }
      else { pix = piece * 4 + (lane >> 4); dy = ph; dx = (lane >> 3) & 1; slot = lane & 7; }
      const int r = pix / HW_, c = pix - r * HW_;
      const int y = ty * TH + r - 1, xx = tx * TW + c - 1;
      const bool ok = pix < HPIX && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
      const size_t e = (((size_t)n * 2 * H + 2 * y + dy) * (2 * W) + 2 * xx + dx) * 64 + ((slot ^ (pix & 7)) * 8);
      const void* src = ok ? (const void*)(x + e) : (const void*)zero_page;
      dma16(src, __builtin_amdgcn_readfirstlane(dst + piece * 1024));
    }
  };
  for (int u = 0; u < AHEAD; ++u) if (u < nstage) issue(u);
  unsigned acc = 0;
  for (int u = 0; u < nstage; ++u) {
    const int younger = (nstage - 1 - u < AHEAD - 1) ? nstage - 1 - u : AHEAD - 1;
    if (younger >= 3) vmwait<3 * PW>(); else if (younger == 2) vmwait<2 * PW>(); else if (younger == 1) vmwait<PW>(); else vmwait<0>();
    __syncthreads();
    acc += *reinterpret_cast<const unsigned*>(lds + (u % NST) * STAGE + tid * 16);      // touch the stage
    if (u + AHEAD < nstage) issue(u + AHEAD);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT> static void run(const uint16_t* x, unsigned* sink, int N, int H, int W) {
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather<PAT>, dim3(256), dim3(256), 0, 0, x, N, H, W, tiles_x, tiles_y, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)N * 4 * H * W * 128;
  printf("pattern %c  %d x %d x %d: %7.1f us  %5.2f TB/s of input (%.0f MB; halo re-reads not counted)\n", PAT == 0 ? 'A' : 'B', N, H, W, best * 1e3, bytes / best / 1e9, bytes / 1e6);
}

int main() {
  const int N = 32, H = 96, W = 96;
  uint16_t* x; unsigned* sink;
  hipMalloc(&x, (size_t)N * 4 * H * W * 128); hipMemset(x, 1, (size_t)N * 4 * H * W * 128);
  hipMalloc(&sink, 64);
  run<0>(x, sink, N, H, W); run<1>(x, sink, N, H, W); run<0>(x, sink, N, H, W); run<1>(x, sink, N, H, W);
  run<0>(x, sink, N, 48, 48); run<1>(x, sink, N, 48, 48);
  return 0;
}
