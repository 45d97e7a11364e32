// nsx_grid.hpp — device-side pieces shared by the persistent (one launch, grid-wide exchanges) kernels:
// the Gram-Schmidt sweep k_mgs (nsx_blas.hip) and the Schur-complement CG k_cg_schur (nsx_cg.hip).
//
// A grid-wide sum without atomics on shared counters: workgroup b stores the BIT PATTERN of its partial sum in
// mailbox[b] (initially GX_EMPTY), workgroup 0 waits for all of them, adds them in a fixed order and stores the total in
// one word that every workgroup picks up.  Mailbox and total words are written and read with agent-scope relaxed atomics
// (global_store / global_load sc1: write-through, L1-bypassing — the hand-off form of /opt/skills/guides/MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility").  Every wait is bounded by a wall-clock timeout, so a grid
// that is not co-resident ends loudly instead of hanging the GPU.
#pragma once
#include "nsx_internal.hpp"

namespace nsx {

constexpr unsigned long long GX_EMPTY = ~0ull;
constexpr unsigned long long GX_TIMEOUT_TICKS = 200000000ull;  // wall_clock64 runs at 100 MHz: 2 s

__device__ __forceinline__ unsigned long long gx_bits(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return b == GX_EMPTY ? 0x7ff8000000000000ull : b;  // a NaN with the sentinel's bit pattern becomes the canonical NaN
}
__device__ __forceinline__ void gx_post(unsigned long long *p, double v) {
  __hip_atomic_store(p, gx_bits(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gx_clear(unsigned long long *p) {
  __hip_atomic_store(p, GX_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait for a mailbox word; *err is raised (and 0 returned) after the timeout.  One poll in flight at a time: keeping three in
// flight (bunched or evenly spaced) measured 0.2 - 0.6 us SLOWER per exchange (tools/exchange_bench.hip): the extra requests
// queue in front of the answer.
__device__ __forceinline__ unsigned long long gx_load(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long gx_wait(const unsigned long long *p, int *err) {
  unsigned long long b = gx_load(p);
  if (b != GX_EMPTY) return b;
  unsigned long long t0 = 0;
  for (unsigned int spin = 1;; ++spin) {
    __builtin_amdgcn_s_sleep(1);
    b = gx_load(p);
    if (b != GX_EMPTY) return b;
    if ((spin & 255u) == 0) {  // the clock is a scalar memory read: look at it rarely
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > GX_TIMEOUT_TICKS) {
        *err = 1;
        return 0;
      }
    }
  }
}
// the same for N words `stride` apart, all polled together: one round trip per poll round, not one per word
template <int N>
__device__ __forceinline__ void gx_wait_n(const unsigned long long *p, size_t stride, double (&out)[N], int *err) {
  unsigned long long b[N], t0 = 0;
  for (unsigned int spin = 1;; ++spin) {
    bool all = true;
#pragma unroll
    for (int v = 0; v < N; ++v) b[v] = gx_load(p + (size_t)v * stride);
#pragma unroll
    for (int v = 0; v < N; ++v) all = all && b[v] != GX_EMPTY;
    if (all) break;
    __builtin_amdgcn_s_sleep(1);
    if ((spin & 255u) == 0) {
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > GX_TIMEOUT_TICKS) {
        *err = 1;
#pragma unroll
        for (int v = 0; v < N; ++v) b[v] = 0;
        break;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < N; ++v) out[v] = __longlong_as_double((long long)b[v]);
}
__device__ __forceinline__ double gx_wait_value(const unsigned long long *p, int *err) { return __longlong_as_double((long long)gx_wait(p, err)); }

// sum over the 64 lanes of the wave in a fixed order, result wave-uniform: four DPP stages inside each row of 16 lanes,
// then the four row sums are read out of lanes 0/16/32/48
__device__ __forceinline__ double gx_wave_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * k), __builtin_amdgcn_readlane(lo, 16 * k));
  return (r[0] + r[1]) + (r[2] + r[3]);
}
// fixed-order sum over the 256 threads of the block, result in every thread (sh: 4 doubles, reusable after the next barrier)
__device__ __forceinline__ double gx_block_sum(double v, double *sh) {
  v = gx_wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

}  // namespace nsx
