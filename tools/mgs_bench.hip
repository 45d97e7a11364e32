// mgs_bench.hip — development tool: the Gram-Schmidt sweep kernels of csrc/nsx_blas.hip on their own, with wall-clock stamps
// of workgroup 0 and of the last workgroup (NSX_MGS_TRACE), to see where a sweep's microseconds go.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -DNSX_MGS_TRACE -Dnsx=nsx_tool -o gpurun_out/mgs_bench tools/mgs_bench.hip
// (-Dnsx=nsx_tool: the kernels get names of their own, libnsx.so is not linked)
#include "../navierstokes_project_nm4pde_amd/csrc/nsx_blas.hip"

namespace nsx {  // the two functions of nsx_comm.hip the host code of nsx_blas.hip refers to (never called here)
void comm_allreduce_partials(nsx_handle *, double *, int) {}
void comm_allreduce_scalars(nsx_handle *, int, int) {}
int cg_dirty_words(nsx_handle *) { return 0; }
}  // namespace nsx

#include <cmath>
#include <random>
#include <string>

using namespace nsx;
typedef unsigned long long u64;

__global__ void k_set_trace(u64 *p) { g_mgs_trace = p; }

template <int E, int M, bool PF>
static void run(const char *name, int n, int dim, int nwg, double *w, double *w0, MgsArgs V, u64 *box, double *scal, int *err, u64 *tail,
                double *pub, u64 *trace_dev, bool trace) {
  const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipMemset(box, 0xff, 2 * region * sizeof(u64)));
  const int nv = M + M * (M - 1) / 2 + 1;
  const int words = M == 1 ? 0 : (cdiv(dim, M) + 1) * nv * nwg;
  int parity = 0;
  u64 seq = 1;
  u64 *null_trace = nullptr;
  hipLaunchKernelGGL(k_set_trace, dim3(1), dim3(1), 0, 0, null_trace);
  const int reps = 50;
  float ms = 0;
  for (int rep = 0; rep < reps + 3; ++rep) {
    if (rep == 3) HIP_CHECK(hipEventRecord(a, 0));
    if (rep == reps + 2 && trace) hipLaunchKernelGGL(k_set_trace, dim3(1), dim3(1), 0, 0, trace_dev);
    HIP_CHECK(hipMemcpyAsync(w, w0, (size_t)n * 8, hipMemcpyDeviceToDevice, 0));
    u64 *bx = box + (size_t)parity * region, *bn = box + (size_t)(1 - parity) * region;
    if constexpr (M == 1)
      hipLaunchKernelGGL((k_mgs<E>), dim3(nwg), dim3(256), 0, 0, n, n, 0, w, V, dim, bx, bn, nwg, rep ? dim + 2 : 0, scal, err, tail, 1, 0, pub,
                         (u64 *)(pub + 64), seq, -1);
    else
      hipLaunchKernelGGL((k_mgs_blk<E, M, PF>), dim3(nwg), dim3(256), 0, 0, n, n, 0, w, V, dim, bx, bn, rep ? words : 0, scal, err, tail, 1, 0, pub,
                         (u64 *)(pub + 64), seq, -1);
    parity ^= 1;
    ++seq;
  }
  HIP_CHECK(hipEventRecord(b, 0));
  HIP_CHECK(hipEventSynchronize(b));
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  // subtract the copy of w0 (measured separately below)
  std::vector<double> hs(dim + 1);
  HIP_CHECK(hipMemcpy(hs.data(), scal, (dim + 1) * 8, hipMemcpyDeviceToHost));
  printf("%-28s dim %2d nwg %3d : %7.2f us per (copy + sweep)   h0 %.12e h_last %.12e |w|^2 %.12e err %d\n", name, dim, nwg, 1e3 * ms / reps, hs[0],
         hs[dim - 1], hs[dim], *err);
  if (trace) {
    std::vector<u64> t(128);
    HIP_CHECK(hipMemcpy(t.data(), trace_dev, 128 * 8, hipMemcpyDeviceToHost));
    for (int g = 0; g < 2; ++g) {
      printf("   %s:", g == 0 ? "wg 0   " : "wg last");
      for (int k = 0; k < 64 && t[g * 64 + k]; ++k) printf(" %5.2f", 0.01 * (double)(t[g * 64 + k] - t[0]));
      printf("  (us since wg 0 started; stamps: start | per exchange: sums done, posted, picked up | end)\n");
    }
    HIP_CHECK(hipMemset(trace_dev, 0, 128 * 8));
  }
}

// the one-exchange sweep (k_mgs_one): cold = a 1 GB buffer is read between the sweeps so that the basis comes from HBM, not from the
// Infinity Cache (inside a GMRES iteration the product and the triangular solve push it out)
__global__ void k_flush(const double *__restrict__ p, size_t n, double *out) {
  double a = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
  if (a == 1.2345e300) *out = a;
}
template <int E, int DMAX>
static void run_one(const char *name, int n, int dim, double *w, double *w0, MgsArgs V, u64 *box, double *scal, int *err, u64 *tail, double *pub,
                    u64 *trace_dev, double *gram, const double *flush, size_t n_flush) {
  const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
  const int nwg = cdiv(n, E * 256);
  HIP_CHECK(hipMemset(box, 0xff, 2 * region * sizeof(u64)));
  HIP_CHECK(hipMemset(gram, 0, 1024 * 8));
  const int words = (2 * dim + 2) * nwg;
  int parity = 0;
  u64 seq = 1;
  u64 *null_trace = nullptr;
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  float tot = 0;
  const int reps = 12;
  for (int rep = 0; rep < reps + 2; ++rep) {
    hipLaunchKernelGGL(k_set_trace, dim3(1), dim3(1), 0, 0, rep == reps + 1 ? trace_dev : null_trace);
    HIP_CHECK(hipMemcpyAsync(w, w0, (size_t)n * 8, hipMemcpyDeviceToDevice, 0));
    if (flush) hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, flush, n_flush, scal + 63);
    u64 *bx = box + (size_t)parity * region, *bn = box + (size_t)(1 - parity) * region;
    HIP_CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_mgs_one<E, DMAX>), dim3(nwg), dim3(256), 0, 0, n, n, 0, w, V, dim, gram, bx, bn, rep ? words : 0, scal, err, tail, 1, 0, pub,
                       (u64 *)(pub + 64), seq, -1, 1e-2);
    HIP_CHECK(hipEventRecord(b, 0));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 2) tot += ms;
    parity ^= 1;
    ++seq;
  }
  printf("%-22s dim %2d nwg %3d %s: %7.2f us per sweep (%.0f MB, %.2f TB/s) err %d\n", name, dim, nwg, flush ? "cold" : "hot ", 1e3 * tot / reps,
         8e-6 * n * (dim + 2), 8e-6 * n * (dim + 2) / (1e3 * tot / reps), *err);
  std::vector<u64> t(128);
  HIP_CHECK(hipMemcpy(t.data(), trace_dev, 128 * 8, hipMemcpyDeviceToHost));
  for (int g = 0; g < 2; ++g) {
    printf("   %s:", g == 0 ? "wg 0   " : "wg last");
    for (int k = 0; k < 64 && t[g * 64 + k]; ++k) printf(" %5.2f", 0.01 * (double)(t[g * 64 + k] - t[0]));
    printf("  (us since wg 0 started: start | loads + sums | posted | totals | solved | updated | stores issued)\n");
  }
  HIP_CHECK(hipMemset(trace_dev, 0, 128 * 8));
}

int main(int argc, char **argv) {
  const int n = 1043658, nvec = 30;
  double *w, *w0, *vs, *scal, *pub;
  int *err;
  u64 *box, *tail, *trace;
  const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
  HIP_CHECK(hipMalloc(&w, (size_t)n * 8));
  HIP_CHECK(hipMalloc(&w0, (size_t)n * 8));
  HIP_CHECK(hipMalloc(&vs, (size_t)n * 8 * nvec));
  HIP_CHECK(hipMalloc(&scal, 64 * 8));
  HIP_CHECK(hipMalloc(&box, 2 * region * 8));
  HIP_CHECK(hipMalloc(&tail, MGS_TAIL * 8));
  HIP_CHECK(hipMalloc(&trace, 128 * 8));
  HIP_CHECK(hipMemset(trace, 0, 128 * 8));
  HIP_CHECK(hipHostMalloc(&pub, 128 * 8, hipHostMallocMapped));
  HIP_CHECK(hipHostMalloc(&err, 64, hipHostMallocMapped));
  *err = 0;
  std::vector<double> host((size_t)n * (nvec + 1));
  std::mt19937_64 rng(7);
  std::normal_distribution<double> g;
  // an almost (not exactly) orthonormal basis: random vectors scaled to unit norm, correlations ~ n^-1/2
  for (auto &x : host) x = g(rng) / std::sqrt((double)n);
  HIP_CHECK(hipMemcpy(vs, host.data(), (size_t)n * 8 * nvec, hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(w0, host.data() + (size_t)n * nvec, (size_t)n * 8, hipMemcpyHostToDevice));
  if (argc > 1 && std::string(argv[1]) == "one") {
    double *gram, *flush;
    const size_t n_flush = (size_t)128 << 20;  // 1 GB
    HIP_CHECK(hipMalloc(&gram, 1024 * 8));
    HIP_CHECK(hipMalloc(&flush, n_flush * 8));
    HIP_CHECK(hipMemset(flush, 0, n_flush * 8));
    for (int dim : {2, 5, 8, 10, 14}) {
      MgsArgs V;
      for (int i = 0; i < MGS_STEPS; ++i) V.v[i] = i < dim ? vs + (size_t)i * n : nullptr;
      run_one<8, 10>("one exchange <8,10>", n, dim, w, w0, V, box, scal, err, tail, pub, trace, gram, nullptr, 0);
      run_one<8, 10>("one exchange <8,10>", n, dim, w, w0, V, box, scal, err, tail, pub, trace, gram, flush, n_flush);
    }
    return 0;
  }
  const int dims[] = {4, 10, 16};
  for (int dim : dims) {
    MgsArgs V;
    for (int i = 0; i < MGS_STEPS; ++i) V.v[i] = i < dim ? vs + (size_t)i * n : nullptr;
    run<10, 1, true>("chain, link by link", n, dim, 512, w, w0, V, box, scal, err, tail, pub, trace, false);
    run<8, 2, true>("2 links per exchange", n, dim, 512, w, w0, V, box, scal, err, tail, pub, trace, false);
    run<8, 3, true>("3 links per exchange", n, dim, 512, w, w0, V, box, scal, err, tail, pub, trace, false);
    run<8, 4, true>("4 links per exchange", n, dim, 512, w, w0, V, box, scal, err, tail, pub, trace, dim == 10);
    run<20, 4, false>("4 links, 256 wg x 16", n, dim, 256, w, w0, V, box, scal, err, tail, pub, trace, false);
  }
  return 0;
}
