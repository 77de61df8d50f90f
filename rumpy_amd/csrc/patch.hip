// Device-side training-patch pipeline: random crop + horizontal / vertical flip + transpose ("rot90")
// + uint8 -> float / 255 on the GPU, from uint8 HWC images that stay resident in HBM.  Replaces, per batch,
// SuperResImages.__getitem__ -> image_augment_crop -> random_flip_rotate -> image_patch_selection -> extract_image_patch
// (rumpy/sr_tools/data_handler.py:570-645, rumpy/image_tools/image_manipulation/image_functions.py:245-362) and the
// torchvision ToTensor in front of them (data_handler.py:472-486).  The random numbers are drawn on the host in the
// reference's order (rumpy_amd/sr_tools/device_patches.py); the kernel is a pure gather and is bit-exact: the reference
// augments the whole image A = T(V(H(I))) and then slices A[:, y:y+crop, x:x+crop], here each output element reads
//   I[c][vflip ? Hi-1-y' : y'][hflip ? Wi-1-x' : x']   with (y', x') = rot ? (x+j, y+i) : (y+i, x+j)
// and divides by 255 in fp32 (IEEE division, as Tensor.div does).
#include "common.hpp"

__global__ void __launch_bounds__(256) patch_gather_kernel(const uint8_t* __restrict__ images, const rumpy_patch_item* __restrict__ items,
                                                           float* __restrict__ out, int C, int crop, int use_hr, int scale) {
  const rumpy_patch_item it = items[blockIdx.y];
  const int Hi = use_hr ? it.lr_h * scale : it.lr_h, Wi = use_hr ? it.lr_w * scale : it.lr_w;
  const uint8_t* img = images + (use_hr ? it.hr_off : it.lr_off);
  const int y0 = use_hr ? it.y * scale : it.y, x0 = use_hr ? it.x * scale : it.x;
  const int plane = crop * crop;
  float* o = out + (size_t)blockIdx.y * C * plane;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < C * plane; e += gridDim.x * blockDim.x) {
    const int c = e / plane, r = e - c * plane, i = r / crop, j = r - i * crop;
    const int ya = y0 + i, xa = x0 + j;                 // coordinates in the augmented image
    const int yb = it.rot ? xa : ya, xb = it.rot ? ya : xa;
    const int ys = it.vflip ? Hi - 1 - yb : yb, xs = it.hflip ? Wi - 1 - xb : xb;
    o[e] = (float)img[((size_t)ys * Wi + xs) * C + c] / 255.0f;
  }
}

extern "C" int rumpy_patch_gather(const rumpy_patch_args* p, void* stream) {
  if (!p || !p->images || !p->items || !p->out_lr || p->N <= 0 || p->C <= 0 || p->crop <= 0 || p->scale <= 0) {
    rumpy_set_error("rumpy_patch_gather: bad argument"); return RUMPY_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int lr_elems = p->C * p->crop * p->crop;
  int gx = (lr_elems + 255) / 256; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(patch_gather_kernel, dim3(gx, p->N), dim3(256), 0, s, p->images, p->items, p->out_lr, p->C, p->crop, 0, p->scale);
  if (p->out_hr) {
    const int hc = p->crop * p->scale;
    int gh = (p->C * hc * hc + 255) / 256; if (gh > 64) gh = 64;
    hipLaunchKernelGGL(patch_gather_kernel, dim3(gh, p->N), dim3(256), 0, s, p->images, p->items, p->out_hr, p->C, hc, 1, p->scale);
  }
  return rumpy_check_launch("rumpy_patch_gather");
}
