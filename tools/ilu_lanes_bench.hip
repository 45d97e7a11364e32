// ilu_lanes_bench.hip — development tool: what does one tick of the lane-owner triangular solve cost, and why?
// Runs the device code of csrc/nsx_ilu_lanes.hpp (the very functions k_ilu_solve_lanes uses) on a synthetic stream:
//   W waves x T ticks, random columns among R rows, a row every ~6 ticks per lane.
// Variants: HOT = every wave reads the SAME T slabs (they stay in L2: no HBM latency), COLD = its own slabs (the real case);
// prefetch depth PF in {4, 8}.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ilu_lanes_bench tools/ilu_lanes_bench.hip && tools/ilu_lanes_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../navierstokes_project_nm4pde_amd/csrc/nsx_ilu_lanes.hpp"

#define CK(e)                                                                  \
  do {                                                                         \
    hipError_t r_ = (e);                                                       \
    if (r_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

template <int NCOMP, int E, int PF>
__global__ __launch_bounds__(64) void k_bench(int ticks, int rows, int hot, const uint32_t *__restrict__ meta, const double *__restrict__ val, double *out) {
  extern __shared__ double xs[];
  const unsigned lane = threadIdx.x;
  for (int t = lane; t < (rows + 64) * NCOMP; t += 64) xs[t] = 1.0 + 1e-3 * t;
  __builtin_amdgcn_wave_barrier();
  const int s0 = hot ? 0 : blockIdx.x * ticks;
  nsx::LaneSlot<E> A[PF];
  nsx::lane_load<E, PF>(A, s0, meta, val, lane);
  nsx::lane_sweep<NCOMP, E, PF>(A, s0, s0 + ticks, meta, val, lane, (uint32_t)(rows + lane) * 8u * NCOMP);
  double acc = 0.0;
  for (int t = lane; t < rows * NCOMP; t += 64) acc += xs[t];
  if (acc == 123.456) out[blockIdx.x] = acc;
}

template <int E>
static void run(int T, int R) {
  constexpr int MW = (E + 2) / 2;
  const int maxW = 2048;
  std::mt19937 rng(7);
  std::vector<uint32_t> meta((size_t)(maxW * T + 64) * 64 * MW, 0u);
  std::vector<double> val((size_t)(maxW * T + 64) * 64 * E, 1e-3);
  auto put = [&](size_t slot, int k, uint32_t v) {
    uint32_t &wd = meta[slot * MW + k / 2];
    wd = (k & 1) ? ((wd & 0xffffu) | (v << 16)) : ((wd & 0xffff0000u) | v);
  };
  for (int w = 0; w < maxW; ++w) {
    int left[64] = {0}, row[64] = {0};
    for (int t = 0; t < T; ++t)
      for (int l = 0; l < 64; ++l) {
        uint32_t first = 0;
        if (left[l] == 0) {
          left[l] = 1 + rng() % (12 / E + 1);
          row[l] = rng() % R;
          first = 1;
        }
        const size_t slot = ((size_t)w * T + t) * 64 + l;
        for (int e = 0; e < E; ++e) put(slot, e, (uint32_t)(rng() % R) * 24u);
        meta[slot * MW] |= first | (left[l] == 1 ? 2u : 0u);
        put(slot, E, (uint32_t)row[l] * 24u);
        --left[l];
      }
  }
  uint32_t *dm;
  double *dv, *dout;
  CK(hipMalloc(&dm, meta.size() * 4));
  CK(hipMalloc(&dv, val.size() * 8));
  CK(hipMalloc(&dout, maxW * 8));
  CK(hipMemcpy(dm, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dv, val.data(), val.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const size_t shm = (size_t)(R + 64) * 24;
  for (int hot = 1; hot >= 0; --hot)
    for (int pf : {4, 8})
      for (int W : {64, 256, 512, 1024, 2048}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(a));
          if (pf == 4) hipLaunchKernelGGL((k_bench<3, E, 4>), dim3(W), dim3(64), shm, 0, T, R, hot, dm, dv, dout);
          else hipLaunchKernelGGL((k_bench<3, E, 8>), dim3(W), dim3(64), shm, 0, T, R, hot, dm, dv, dout);
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float ms;
          CK(hipEventElapsedTime(&ms, a, b));
          best = ms < best ? ms : best;
        }
        printf("E %d %s PF %d waves %4d: %7.1f us  %6.1f ns/tick  %7.1f GB/s\n", E, hot ? "hot " : "cold", pf, W, best * 1e3, best * 1e6 / T,
               (double)W * T * 64 * (8 * E + 4 * MW) / (best * 1e-3) * 1e-9);
      }
  CK(hipFree(dm));
  CK(hipFree(dv));
  CK(hipFree(dout));
}

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 96, R = argc > 2 ? atoi(argv[2]) : 750;
  printf("# T = %d ticks per wave, %d rows in LDS; us per launch, ns per tick, GB/s of stream\n", T, R);
  run<1>(T, R);
  run<2>(T, R);
  run<3>(T, R);
  run<4>(T, R);
  return 0;
}
