// development tool (round 4): which XCD / shader engine / CU does bit i of a stream's CU mask (hipExtStreamCreateWithCUMask) stand for?
// For every bit of the first mask words a stream with ONLY that bit set runs 64 workgroups that record where they ran.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/cu_mask_probe tools/cu_mask_probe.hip && gpurun_out/cu_mask_probe > gpurun_out/cu_mask_probe.txt
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

__global__ void where(unsigned *out) {
  if (threadIdx.x == 0) {
    unsigned xcc, id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = id;
  }
}

// co-residency on a masked stream: n workgroups that each take 72 KB of LDS (two fit a CU) meet at a counter; returns how many arrived
// before the first one gave up after 50 ms
__global__ void meet(unsigned *count, unsigned n, unsigned *result, unsigned *where_) {
  extern __shared__ double hog[];
  if (threadIdx.x == 0) {
    hog[0] = 1.0;
    unsigned xcc, id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    where_[blockIdx.x] = (xcc << 16) | (((id >> 13) & 0x7) << 8) | ((id >> 8) & 0xf);
    __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 5000000ull) break;
    }
    if (blockIdx.x == 0) *result = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static void residency(const char *what, hipStream_t st) {
  unsigned *d;
  hipMalloc(&d, (2 + 1024) * sizeof(unsigned));
  hipFuncSetAttribute((const void *)meet, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  for (unsigned n : {256u, 448u, 480u, 496u, 504u, 512u}) {
    hipMemsetAsync(d, 0, (2 + 1024) * sizeof(unsigned), st);
    hipLaunchKernelGGL(meet, dim3(n), dim3(256), 72 * 1024, st, d, n, d + 1, d + 2);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(2 + 1024);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<unsigned> cus(h.begin() + 2, h.begin() + 2 + n);
    printf("# %s: %u workgroups (72 KB LDS each): %u met; they ran on %zu distinct CUs\n", what, n, h[1], cus.size());
  }
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, words = (cus + 31) / 32;
  printf("# %s, %d CUs, %d mask words\n# bit -> distinct (xcc, se, sa, cu) the 64 workgroups of a stream with only that bit ran on\n", p.name, cus, words);
  {
    hipStream_t plain, masked;
    hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    residency("plain stream", plain);
    std::vector<uint32_t> mask(words, 0xffffffffu);
    mask[0] &= ~0xffu;  // all CUs but bits 0..7
    if (hipExtStreamCreateWithCUMask(&masked, words, mask.data()) == hipSuccess) residency("stream without mask bits 0..7", masked);
    if (getenv("CU_MASK_RESIDENCY_ONLY")) return 0;
  }
  unsigned *out;
  hipMalloc(&out, 2 * 64 * sizeof(unsigned));
  std::vector<unsigned> host(2 * 64);
  for (int bit = 0; bit < cus; ++bit) {
    std::vector<uint32_t> mask(words, 0u);
    mask[bit / 32] = 1u << (bit % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, words, mask.data()) != hipSuccess) {
      printf("%d: stream creation failed\n", bit);
      continue;
    }
    hipMemsetAsync(out, 0xff, 2 * 64 * sizeof(unsigned), st);
    hipLaunchKernelGGL(where, dim3(64), dim3(64), 0, st, out);
    hipStreamSynchronize(st);
    hipMemcpy(host.data(), out, host.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<std::vector<unsigned>> seen;
    for (int b = 0; b < 64; ++b) {
      const unsigned id = host[2 * b + 1];
      seen.insert({host[2 * b] & 0xf, (id >> 13) & 0x7, (id >> 16) & 0x1, (id >> 8) & 0xf});
    }
    printf("%d:", bit);
    for (const auto &s : seen) printf(" (%u,%u,%u,%u)", s[0], s[1], s[2], s[3]);
    printf("\n");
    hipStreamDestroy(st);
  }
  return 0;
}
