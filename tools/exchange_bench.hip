// exchange_bench.hip — what does one grid-wide sum cost inside a persistent kernel on MI355X, and which shape is fastest?
// Development tool behind the choices in csrc/nsx_grid.hpp (k_mgs, k_cg_schur).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/exchange_bench tools/exchange_bench.hip && gpurun_out/exchange_bench
// Every workgroup contributes one double per exchange; all workgroups need the fixed-order total before they go on.
// Variants: polling (single load + s_sleep / three loads in flight, bunched / evenly spaced), reducer shape (workgroup 0 /
// every workgroup reads all mailboxes / 8 sub-reducers then 8 words per workgroup), with and without an HBM stream
// (loads of the next "basis vector" issued before the wait, as the Gram-Schmidt sweep does).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                           \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

typedef unsigned long long u64;
constexpr u64 EMPTY = ~0ull;
constexpr int MAX_WG = 1024, RING = 4;

__device__ __forceinline__ u64 ld(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int POLL>
__device__ __forceinline__ u64 wait_word(const u64 *p) {
  u64 b0 = ld(p);
  if (b0 != EMPTY) return b0;
  if (POLL == 0) {
    for (long k = 0; k < 50000000; ++k) {
      __builtin_amdgcn_s_sleep(1);
      b0 = ld(p);
      if (b0 != EMPTY) return b0;
    }
    return 0;
  }
  if (POLL == 2) __builtin_amdgcn_s_sleep(10);
  u64 b1 = ld(p);
  if (POLL == 2) __builtin_amdgcn_s_sleep(10);
  u64 b2 = ld(p);
  if (POLL == 2) __builtin_amdgcn_s_sleep(10);
  b0 = ld(p);
  for (long k = 0; k < 50000000; ++k) {
    if (b1 != EMPTY) return b1;
    b1 = b2;
    b2 = b0;
    b0 = ld(p);
  }
  return 0;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// SHAPE 0: workgroup 0 reduces, everybody polls one total.  1: everybody reads all mailboxes.  2: workgroups 0..7 reduce the
// mailboxes of the workgroups congruent to them mod 8, everybody polls those 8 sub-totals.
template <int POLL, int SHAPE>
__global__ __launch_bounds__(256) void k_bench(int iters, u64 *box, const double *stream, size_t stream_len, int stream_e, double *out) {
  __shared__ double sh[2][4];
  __shared__ u64 bc;
  const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  u64 *rows = box, *totals = box + (size_t)RING * MAX_WG;  // totals: RING x 8
  double acc_stream = 0.0, check = 0.0;
  size_t pos = ((size_t)wg * 256 + tid);
  for (int it = 0; it < iters; ++it) {
    u64 *row = rows + (size_t)(it % RING) * MAX_WG, *row2 = rows + (size_t)((it + 2) % RING) * MAX_WG;
    u64 *tot = totals + (size_t)(it % RING) * 8, *tot2 = totals + (size_t)((it + 2) % RING) * 8;
    // the stream: loads issued before the exchange, consumed after it
    double sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sv[k] = 0.0;
    if (stream_e > 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < stream_e) {
          sv[k] = stream[pos & (stream_len - 1)];  // stream_len is a power of two
          pos += (size_t)nwg * 256;
        }
    }
    const double part = block_sum(1.0 + 1e-3 * (it & 7), sh[0]);
    if (tid == 0) {
      st(row + wg, (u64)__double_as_longlong(part));
      st(row2 + wg, EMPTY);
    }
    double total;
    if (SHAPE == 0) {
      if (wg == 0) {
        double a = 0.0;
        for (int q = tid; q < nwg; q += 256) a += __longlong_as_double((long long)wait_word<POLL>(row + q));
        const double t = block_sum(a, sh[1]);
        if (tid == 0) st(tot, (u64)__double_as_longlong(t));
        if (tid == 1) st(tot2, EMPTY);
      }
      if (tid == 0) bc = wait_word<POLL>(tot);
      __syncthreads();
      total = __longlong_as_double((long long)bc);
    } else if (SHAPE == 1) {
      double a = 0.0;
      for (int q = tid; q < nwg; q += 256) a += __longlong_as_double((long long)wait_word<POLL>(row + q));
      total = block_sum(a, sh[1]);
      __syncthreads();
    } else {
      if (wg < 8) {
        double a = 0.0;
        for (int q = wg + 8 * tid; q < nwg; q += 8 * 256) a += __longlong_as_double((long long)wait_word<POLL>(row + q));
        const double t = block_sum(a, sh[1]);
        if (tid == 0) st(tot + wg, (u64)__double_as_longlong(t));
        if (tid == 1) st(tot2 + wg, EMPTY);
      }
      double a = 0.0;
      if (tid < 8) a = __longlong_as_double((long long)wait_word<POLL>(tot + tid));
      // lanes 0..7 of wave 0 hold the sub-totals: fixed-order sum, broadcast through LDS
      a = wave_sum(a);
      if (tid == 0) bc = (u64)__double_as_longlong(a);
      __syncthreads();
      total = __longlong_as_double((long long)bc);
    }
    check += total;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc_stream += sv[k] * total;
  }
  if (tid == 0) out[wg] = check + 1e-300 * acc_stream;
}

// ORDER: where the stream loads of the next link are issued.  0: before the partial sums (they queue in front of the
// exchange's own traffic); 1: after the workgroup has posted its partial sum and BEFORE it polls the total (workgroup 0: after it
// has gathered the mailboxes and posted the total) — the stream then moves while the exchange's round trips are in flight.
template <int ORDER>
__global__ __launch_bounds__(256) void k_bench_order(int iters, u64 *box, const double *stream, size_t stream_len, int stream_e, double *out) {
  __shared__ double sh[2][4];
  __shared__ u64 bc;
  const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  u64 *rows = box, *totals = box + (size_t)RING * MAX_WG;
  double acc_stream = 0.0, check = 0.0;
  size_t pos = ((size_t)wg * 256 + tid);
  for (int it = 0; it < iters; ++it) {
    u64 *row = rows + (size_t)(it % RING) * MAX_WG, *row2 = rows + (size_t)((it + 2) % RING) * MAX_WG;
    u64 *tot = totals + (size_t)(it % RING) * 8, *tot2 = totals + (size_t)((it + 2) % RING) * 8;
    double sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sv[k] = 0.0;
    auto issue = [&]() {
      if (stream_e > 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < stream_e) {
            sv[k] = stream[pos & (stream_len - 1)];
            pos += (size_t)nwg * 256;
          }
      }
    };
    if (ORDER == 0) issue();
    const double part = block_sum(1.0 + 1e-3 * (it & 7), sh[0]);
    if (tid == 0) {
      st(row + wg, (u64)__double_as_longlong(part));
      st(row2 + wg, EMPTY);
    }
    if (wg == 0) {
      double a = 0.0;
      for (int q = tid; q < nwg; q += 256) a += __longlong_as_double((long long)wait_word<0>(row + q));
      const double t = block_sum(a, sh[1]);
      if (tid == 0) st(tot, (u64)__double_as_longlong(t));
      if (tid == 1) st(tot2, EMPTY);
    }
    u64 first = EMPTY;
    if (ORDER == 1) {
      if (tid == 0) first = ld(tot);  // the first poll goes out in front of this wave's own stream loads
      issue();
    }
    if (tid == 0) bc = (ORDER == 1 && first != EMPTY) ? first : wait_word<0>(tot);
    __syncthreads();
    const double total = __longlong_as_double((long long)bc);
    check += total;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc_stream += sv[k] * total;
  }
  if (tid == 0) out[wg] = check + 1e-300 * acc_stream;
}

template <int ORDER>
static void run_order(const char *name, int nwg, int iters, int stream_e, u64 *box, const double *stream, size_t stream_len, double *out) {
  const size_t words = (size_t)RING * MAX_WG + RING * 8;
  CK(hipMemset(box, 0xff, words * sizeof(u64)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_bench_order<ORDER>), dim3(nwg), dim3(256), 0, 0, 20, box, stream, stream_len, stream_e, out);
  CK(hipDeviceSynchronize());
  CK(hipMemset(box, 0xff, words * sizeof(u64)));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((k_bench_order<ORDER>), dim3(nwg), dim3(256), 0, 0, iters, box, stream, stream_len, stream_e, out);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<double> h(nwg);
  CK(hipMemcpy(h.data(), out, nwg * sizeof(double), hipMemcpyDeviceToHost));
  double expect = 0;
  for (int it = 0; it < iters; ++it) expect += nwg * 256.0 * (1.0 + 1e-3 * (it & 7));
  bool ok = true;
  for (int w = 0; w < nwg; ++w) ok = ok && std::abs(h[w] - expect) < 1e-6 * expect;
  printf("%-34s nwg %4d stream %2d x8B/thread : %7.3f us per exchange  %s\n", name, nwg, stream_e, 1e3 * ms / iters, ok ? "ok" : "WRONG TOTALS");
  fflush(stdout);
}

// SERVICE: a fifth wave per workgroup posts and polls; it never has stream loads of its own in flight (a wave's loads return
// in order, so a poll issued behind ten prefetch loads waits for all of them).  PURE0: workgroup 0 does not stream either.
template <int PURE0>
__global__ __launch_bounds__(320) void k_bench_service(int iters, u64 *box, const double *stream, size_t stream_len, int stream_e, double *out) {
  __shared__ double sh[2][4];
  __shared__ u64 bc;
  const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  const bool service = tid >= 256;
  const int lane = tid & 63;
  u64 *rows = box, *totals = box + (size_t)RING * MAX_WG;
  double acc_stream = 0.0, check = 0.0;
  size_t pos = ((size_t)wg * 256 + tid);
  if (PURE0 && wg == 0) stream_e = 0;
  for (int it = 0; it < iters; ++it) {
    u64 *row = rows + (size_t)(it % RING) * MAX_WG, *row2 = rows + (size_t)((it + 2) % RING) * MAX_WG;
    u64 *tot = totals + (size_t)(it % RING) * 8, *tot2 = totals + (size_t)((it + 2) % RING) * 8;
    double sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sv[k] = 0.0;
    if (stream_e > 0 && !service) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < stream_e) {
          sv[k] = stream[pos & (stream_len - 1)];
          pos += (size_t)nwg * 256;
        }
    }
    double v = service ? 0.0 : 1.0 + 1e-3 * (it & 7);
    v = wave_sum(v);
    if (!service && lane == 0) sh[it & 1][tid >> 6] = v;
    __syncthreads();
    if (service) {
      if (lane == 0) {
        const double part = (sh[it & 1][0] + sh[it & 1][1]) + (sh[it & 1][2] + sh[it & 1][3]);
        st(row + wg, (u64)__double_as_longlong(part));
        st(row2 + wg, EMPTY);
      }
      if (wg == 0) {
        // 8 mailboxes per lane, all loads in flight together
        u64 w[16];
        bool all = false;
        for (long spin = 0; spin < 50000000 && !all; ++spin) {
          all = true;
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int q = lane + 64 * k;
            w[k] = q < nwg ? ld(row + q) : 0;
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) all = all && w[k] != EMPTY;
        }
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += (lane + 64 * k < nwg) ? __longlong_as_double((long long)w[k]) : 0.0;
        a = wave_sum(a);
        if (lane == 0) {
          st(tot, (u64)__double_as_longlong(a));
          st(tot2, EMPTY);
        }
      }
      if (lane == 0) bc = wait_word<0>(tot);
    }
    __syncthreads();
    const double total = __longlong_as_double((long long)bc);
    check += total;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc_stream += sv[k] * total;
  }
  if (tid == 0) out[wg] = check + 1e-300 * acc_stream;
}

template <int PURE0>
static void run_service(const char *name, int nwg, int iters, int stream_e, u64 *box, const double *stream, size_t stream_len, double *out) {
  const size_t words = (size_t)RING * MAX_WG + RING * 8;
  CK(hipMemset(box, 0xff, words * sizeof(u64)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_bench_service<PURE0>), dim3(nwg), dim3(320), 0, 0, 20, box, stream, stream_len, stream_e, out);
  CK(hipDeviceSynchronize());
  CK(hipMemset(box, 0xff, words * sizeof(u64)));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((k_bench_service<PURE0>), dim3(nwg), dim3(320), 0, 0, iters, box, stream, stream_len, stream_e, out);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<double> h(nwg);
  CK(hipMemcpy(h.data(), out, nwg * sizeof(double), hipMemcpyDeviceToHost));
  double expect = 0;
  for (int it = 0; it < iters; ++it) expect += nwg * 256.0 * (1.0 + 1e-3 * (it & 7));
  bool ok = true;
  for (int w = 0; w < nwg; ++w) ok = ok && std::abs(h[w] - expect) < 1e-6 * expect;
  printf("%-34s nwg %4d stream %2d x8B/thread : %7.3f us per exchange  %s\n", name, nwg, stream_e, 1e3 * ms / iters, ok ? "ok" : "WRONG TOTALS");
  fflush(stdout);
}

template <int POLL, int SHAPE>
static void run(const char *name, int nwg, int iters, int stream_e, u64 *box, const double *stream, size_t stream_len, double *out) {
  const size_t words = (size_t)RING * MAX_WG + RING * 8;
  CK(hipMemset(box, 0xff, words * sizeof(u64)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_bench<POLL, SHAPE>), dim3(nwg), dim3(256), 0, 0, 20, box, stream, stream_len, stream_e, out);  // warm-up
  CK(hipDeviceSynchronize());
  CK(hipMemset(box, 0xff, words * sizeof(u64)));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((k_bench<POLL, SHAPE>), dim3(nwg), dim3(256), 0, 0, iters, box, stream, stream_len, stream_e, out);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<double> h(nwg);
  CK(hipMemcpy(h.data(), out, nwg * sizeof(double), hipMemcpyDeviceToHost));
  double expect = 0;
  for (int it = 0; it < iters; ++it) expect += nwg * 256.0 * (1.0 + 1e-3 * (it & 7));
  bool ok = true;
  for (int w = 0; w < nwg; ++w) ok = ok && std::abs(h[w] - expect) < 1e-6 * expect;
  printf("%-34s nwg %4d stream %2d x8B/thread : %7.3f us per exchange  %s\n", name, nwg, stream_e, 1e3 * ms / iters, ok ? "ok" : "WRONG TOTALS");
  fflush(stdout);
}

int main() {
  u64 *box;
  double *stream, *out;
  const size_t stream_len = (size_t)64 << 20;  // 512 MB of doubles: nothing stays in the caches
  CK(hipMalloc(&box, ((size_t)RING * MAX_WG + RING * 8) * sizeof(u64)));
  CK(hipMalloc(&stream, stream_len * sizeof(double)));
  CK(hipMemset(stream, 0, stream_len * sizeof(double)));
  CK(hipMalloc(&out, MAX_WG * sizeof(double)));
  const int iters = 2000;
  for (int nwg : {256, 512}) {
    for (int se : {0, 8, 16}) {
      run<0, 0>("poll single+sleep, wg0 reduces", nwg, iters, se, box, stream, stream_len, out);
      run_order<0>("stream issued before the sums", nwg, iters, se, box, stream, stream_len, out);
      run_order<1>("stream issued after the post", nwg, iters, se, box, stream, stream_len, out);

    }
  }
  return 0;
}
