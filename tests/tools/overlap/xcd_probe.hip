// Which store / load scope lets two workgroups ON THE SAME XCD hand an 8-byte record {value, tag} to each other through that XCD's L2 (not
// through the memory side), and what does a round trip cost?  The RCAB kernels' pool exchange (csrc/rcab_common.hpp::strip_allsum) writes its
// records with sc1 stores and polls them with sc1 loads: right for strips on different XCDs (their L2s are not coherent with each other), two
// fabric round trips when - as at the headline shape - all strips of an image sit on one XCD.
//
// Ping-pong between workgroup pairs (same XCD: b and b + 8; different XCDs: b and b + 1), one wave each, 2000 round trips, per mode:
//   mode 0: store sc1        / load sc1                    (the product's form)
//   mode 1: store sc0        / load sc0                    (workgroup scope)
//   mode 2: store sc0 sc1    / load sc0 sc1                (system scope)
//   mode 3: store plain      / load sc0
//   mode 4: store sc0        / load sc1
//   mode 5: atomic swap x2 (no scope bit) / atomic or x2 with return (no scope bit)        - read-modify-write in the L2
//   mode 6: atomic swap x2 sc1            / atomic or x2 with return sc1
//   mode 7: store sc0        / atomic or x2 with return (no scope bit)
// Reports ns per round trip (wall_clock64, 100 MHz) and how many pairs gave up (spin limit) or saw a wrong value.
// build + run (GPU box): make -C tests/tools/overlap xcd_probe && tests/tools/overlap/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int SC0 = 1, SC1 = 16;
constexpr unsigned SPIN = 1u << 16;

template <int AUX> __device__ __forceinline__ void st(rsrc_t r, unsigned byte, unsigned v, unsigned tag) {
  __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, tag}, r, byte, 0, AUX);
}
template <int AUX> __device__ __forceinline__ u32x2 ld(rsrc_t r, unsigned byte) { return __builtin_amdgcn_raw_buffer_load_b64(r, byte, 0, AUX); }

template <bool S1> __device__ __forceinline__ void at_swap(unsigned long long* p, unsigned v, unsigned tag) {
  const unsigned long long d = ((unsigned long long)tag << 32) | v;
  if (S1) asm volatile("global_atomic_swap_x2 %0, %1, off sc1" :: "v"(p), "v"(d) : "memory");
  else asm volatile("global_atomic_swap_x2 %0, %1, off" :: "v"(p), "v"(d) : "memory");
}
template <bool S1> __device__ __forceinline__ u32x2 at_or(unsigned long long* p) {
  unsigned long long r;
  const unsigned long long z = 0ull;
  if (S1) asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory");
  else asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory");
  return (u32x2){(unsigned)r, (unsigned)(r >> 32)};
}

template <int MODE> __device__ __forceinline__ void send(rsrc_t r, unsigned long long* base, unsigned byte, unsigned v, unsigned tag) {
  if (MODE == 0) st<SC1>(r, byte, v, tag);
  else if (MODE == 1 || MODE == 4 || MODE == 7) st<SC0>(r, byte, v, tag);
  else if (MODE == 2) st<SC0 | SC1>(r, byte, v, tag);
  else if (MODE == 3) st<0>(r, byte, v, tag);
  else if (MODE == 5) at_swap<false>(base + byte / 8, v, tag);
  else at_swap<true>(base + byte / 8, v, tag);
}
template <int MODE> __device__ __forceinline__ u32x2 recv(rsrc_t r, unsigned long long* base, unsigned byte) {
  if (MODE == 0 || MODE == 4) return ld<SC1>(r, byte);
  if (MODE == 1 || MODE == 3) return ld<SC0>(r, byte);
  if (MODE == 2) return ld<SC0 | SC1>(r, byte);
  if (MODE == 5 || MODE == 7) return at_or<false>(base + byte / 8);
  return at_or<true>(base + byte / 8);
}

// out[pair] = {ticks, failures, xcc of a, xcc of b}
template <int MODE>
__global__ void __launch_bounds__(64) pingpong(unsigned long long* buf, unsigned bytes, int stride, int rounds, unsigned epoch, unsigned* out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  // pairs: stride 8 -> (b, b + 8) inside blocks of 16; stride 1 -> (b, b + 1)
  const int grp = b / (2 * stride), in = b - grp * 2 * stride;
  const bool first = in < stride;
  const int pair = grp * stride + (first ? in : in - stride);
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  // record slots: [pair][direction][lane] 8 bytes
  const unsigned mine = (unsigned)(((pair * 2 + (first ? 0 : 1)) * 64 + lane) * 8), theirs = (unsigned)(((pair * 2 + (first ? 1 : 0)) * 64 + lane) * 8);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned fails = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= rounds; ++i) {
    const unsigned tag = (epoch << 16) + (unsigned)i;
    if (first) send<MODE>(r, buf, mine, (unsigned)(i * 3 + lane), tag);
    u32x2 v = recv<MODE>(r, buf, theirs);
    unsigned spins = 0;
    while (!__all(v.y == tag)) {
      if (++spins > SPIN) { fails |= 1u; break; }
      __builtin_amdgcn_s_sleep(1);
      v = recv<MODE>(r, buf, theirs);
    }
    if (fails) break;
    if (v.x != (unsigned)(i * 3 + lane)) fails |= 2u;
    if (!first) send<MODE>(r, buf, mine, (unsigned)(i * 3 + lane), tag);
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) {
    unsigned* o = out + (pair * 2 + (first ? 0 : 1)) * 4;
    o[0] = (unsigned)(t1 - t0); o[1] = fails; o[2] = xcc & 0xf; o[3] = (unsigned)b;
  }
}

template <int MODE> static void run(int stride, unsigned long long* buf, unsigned bytes, unsigned* out, int nwg, int rounds, unsigned& epoch) {
  hipMemset(out, 0, nwg * 4 * sizeof(unsigned));
  ++epoch;
  hipLaunchKernelGGL(pingpong<MODE>, dim3(nwg), dim3(64), 0, 0, buf, bytes, stride, rounds, epoch, out);
  hipError_t e = hipDeviceSynchronize();
  std::vector<unsigned> h(nwg * 4);
  hipMemcpy(h.data(), out, nwg * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
  double ticks = 0; int gave_up = 0, wrong = 0, same = 0, n = 0;
  for (int p = 0; p < nwg / 2; ++p) {
    const unsigned* a = &h[(p * 2) * 4]; const unsigned* b = &h[(p * 2 + 1) * 4];
    gave_up += ((a[1] | b[1]) & 1) ? 1 : 0; wrong += ((a[1] | b[1]) & 2) ? 1 : 0; same += a[2] == b[2];
    if (!((a[1] | b[1]) & 1)) { ticks += a[0]; ++n; }
  }
  printf("mode %d  pairs (b, b + %d): %3d of %3d pairs on one XCD | %7.1f ns per round trip | gave up %3d, wrong value %3d%s\n", MODE, stride, same, nwg / 2,
         n ? ticks / n / rounds * 10.0 : 0.0, gave_up, wrong, e == hipSuccess ? "" : "  (launch error)");
}

int main() {
  const int nwg = 256, rounds = 2000;
  const unsigned bytes = (nwg / 2) * 2 * 64 * 8;
  unsigned long long* buf; unsigned* out;
  hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes);
  hipMalloc(&out, nwg * 4 * sizeof(unsigned));
  unsigned epoch = 0;
  for (int stride : {8, 1}) {
    run<0>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<1>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<2>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<3>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<4>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<5>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<6>(stride, buf, bytes, out, nwg, rounds, epoch);
    run<7>(stride, buf, bytes, out, nwg, rounds, epoch);
  }
  return 0;
}
