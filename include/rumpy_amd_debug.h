/*
 * rumpy_amd_debug.h - measurement and test hooks exported by librumpy_amd.so BESIDE the drop-in boundary (include/rumpy_amd.h).
 * Nothing here is called by the handlers or the engine on the product path: `rumpy_probe_*` is how bench.py takes the dominant kernel's
 * launch durations in-process (HIP events attached to the dispatches themselves, on the stream the kernel is launched on),
 * `rumpy_debug_*` serve tests/ and tests/tools/.  A reference-side integration binds rumpy_amd.h only.
 */
#ifndef RUMPY_AMD_DEBUG_H
#define RUMPY_AMD_DEBUG_H
#include "rumpy_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- timing probe: HIP events around every launch of one kernel family on its own stream ----
 * kernel_id: 1 = rumpy_conv3x3 with cin_chunks==1 and cout_tiles==1 ; 2 = rumpy_wgrad_grouped ; 3 = any rumpy_conv3x3 ;
 * 4 = (experimental chain kernels, tests/tools/csrc) ; 5 = rumpy_conv_block, rumpy_rcab_* */
int rumpy_probe_begin(int kernel_id, int max_records);
/* synchronises the recorded events; returns launches seen; *total_ms = summed duration */
int rumpy_probe_end(double* total_ms);

/* stamped build of the Cin=64 strip kernel; a->pool receives grid_x*4*8 u64 of s_memrealtime (100 MHz) phase stamps of each wave's
 * first strip (tests/tools/kbench.py) */
int rumpy_debug_conv_stamps(const rumpy_conv_args* a, void* stream);
/* `blocks` workgroups holding 80 KiB of LDS each spin for about `microseconds` on `stream` - a stand-in for a foreign kernel (an RCCL
 * collective on a side stream) that occupies CUs while the product kernels run (tests/test_network_gpu.py) */
int rumpy_debug_occupy(int32_t blocks, float microseconds, void* stream);

/* out[2 i] / out[2 i + 1] = the OCP e4m3 / e5m2 byte the hardware conversion gives for in[i] / scale (tests/test_fp8_gpu.py pins rounding and
 * saturation, which the delayed scaling of precision 'fp8' relies on) */
int rumpy_fp8_convert(const float* in, float scale, void* out, int32_t n, int32_t ovfl, void* stream);   /* ovfl: with MODE.FP16_OVFL set */

#ifdef __cplusplus
}
#endif
#endif
