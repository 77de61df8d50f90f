// Error reporting, device query and the timing probe of the C ABI.
#include "common.hpp"
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <vector>

static thread_local char g_err[512] = "";

void rumpy_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int rumpy_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rumpy_set_error("%s: %s", what, hipGetErrorString(e));
    return RUMPY_E_LAUNCH;
  }
  return RUMPY_OK;
}
extern "C" const char* rumpy_last_error(void) { return g_err; }
// A whole launch list in one call: every kernel entry point of this library is `int fn(const <args>*, void* stream)`, so a pass of the
// engine (60 launches for an EDSR step, 1000 for RCAN) is a table of {entry point, argument block}.  Walking it here instead of from
// the host language removes the per-launch interpreter / FFI cost (measured from Python: 0.98 -> 0.6 ms of host time per EDSR step,
// where the GPU needs 1.27 ms - the host was within 25 % of becoming the bottleneck).  Returns 0, or -(index + 1) of the first entry
// that failed (its message is in rumpy_last_error()).
extern "C" int rumpy_run_list(const rumpy_op* ops, int32_t n, void* stream) {
  if (!ops || n < 0) { rumpy_set_error("rumpy_run_list: bad argument"); return RUMPY_E_ARG; }
  for (int32_t i = 0; i < n; ++i) {
    typedef int (*entry_fn)(const void*, void*);
    const int rc = reinterpret_cast<entry_fn>(const_cast<void*>(ops[i].fn))(ops[i].args, stream);
    if (rc != 0) return -(i + 1);
  }
  return RUMPY_OK;
}
extern "C" int rumpy_abi_version(void) { return 6; }

// diagnostic (tests only): `blocks` workgroups that each hold 80 KiB of LDS (at most two per CU) and spin for about `microseconds` -
// a stand-in for a foreign kernel (an RCCL collective on the side stream) that occupies CUs while the product kernels run
__global__ void __launch_bounds__(256) occupy_kernel(unsigned long long ticks, unsigned* sink) {
  __shared__ unsigned hold[80 * 1024 / 4];
  hold[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink && hold[(threadIdx.x * 7) & 255] == 0xffffffffu) *sink = 1u;
}
extern "C" int rumpy_debug_occupy(int32_t blocks, float microseconds, void* stream) {
  if (blocks <= 0 || !(microseconds > 0.f) || microseconds > 5e6f) { rumpy_set_error("rumpy_debug_occupy: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)(microseconds * 100.f), (unsigned*)nullptr);
  return rumpy_check_launch("rumpy_debug_occupy");
}
extern "C" int rumpy_device_cus(void) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;  // MI355X
  }
  return cus;
}

// XCDs of the current device (conv_chain.hip assigns whole images to XCDs): MI355X = 8 dies of 32 CUs; RUMPY_XCDS overrides
extern "C" int rumpy_device_xcds(void) {
  static int n = 0;
  if (n == 0) {
    const char* e = getenv("RUMPY_XCDS");
    n = e ? atoi(e) : 0;
    if (n <= 0 || n > 16) { const int cus = rumpy_device_cus(); n = cus >= 64 ? 8 : 1; }
  }
  return n;
}

// ---- timing probe ---------------------------------------------------------------------------------------
// Start / stop events of the launches of ONE kernel id, summed by rumpy_probe_end.  The block kernels (id 5) attach the pair to the
// dispatch itself (hipExtLaunchKernelGGL through rumpy_probe_slot: the events take the kernel's own begin / end timestamps, as
// rocprofv3 does, and no marker packet enters the stream); the other ids bracket the launch with hipEventRecord, which costs about
// 2.5 us of marker / dispatch time per launch (it made this probe read 17 % above rocprofv3 on the 17 us block kernel).
static int g_probe_id = 0;
static int g_probe_max = 0;
static std::vector<hipEvent_t> g_probe_ev;   // start/stop pairs
static int g_probe_n = 0;

static bool probe_wants(int kernel_id) { return g_probe_id != 0 && (kernel_id == g_probe_id || (g_probe_id == 3 && kernel_id == 1)); }
bool rumpy_probe_slot(int kernel_id, hipEvent_t* start, hipEvent_t* stop) {
  if (!probe_wants(kernel_id) || g_probe_n >= g_probe_max) return false;
  *start = g_probe_ev[2 * g_probe_n];
  *stop = g_probe_ev[2 * g_probe_n + 1];
  ++g_probe_n;
  return true;
}
void rumpy_probe_pre(int kernel_id, hipStream_t s) {
  if (!probe_wants(kernel_id) || g_probe_n >= g_probe_max) return;
  (void)hipEventRecord(g_probe_ev[2 * g_probe_n], s);
}
void rumpy_probe_post(int kernel_id, hipStream_t s) {
  if (!probe_wants(kernel_id) || g_probe_n >= g_probe_max) return;
  (void)hipEventRecord(g_probe_ev[2 * g_probe_n + 1], s);
  ++g_probe_n;
}
extern "C" int rumpy_probe_begin(int kernel_id, int max_records) {
  if (kernel_id < 1 || kernel_id > 5 || max_records <= 0) { rumpy_set_error("rumpy_probe_begin: bad argument"); return RUMPY_E_ARG; }
  while ((int)g_probe_ev.size() < 2 * max_records) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { rumpy_set_error("rumpy_probe_begin: hipEventCreate failed"); return RUMPY_E_LAUNCH; }
    g_probe_ev.push_back(e);
  }
  g_probe_max = max_records;
  g_probe_n = 0;
  g_probe_id = kernel_id;
  return RUMPY_OK;
}
extern "C" int rumpy_probe_end(double* total_ms) {
  double tot = 0.0;
  const int n = g_probe_n;
  for (int i = 0; i < n; ++i) {
    (void)hipEventSynchronize(g_probe_ev[2 * i + 1]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_probe_ev[2 * i], g_probe_ev[2 * i + 1]) == hipSuccess) tot += ms;
  }
  g_probe_id = 0;
  g_probe_n = 0;
  if (total_ms) *total_ms = tot;
  return n;
}
