// development tool (round 4): which XCD / shader engine / CU does bit i of a stream's CU mask (hipExtStreamCreateWithCUMask) stand for?
// For every bit of the first mask words a stream with ONLY that bit set runs 64 workgroups that record where they ran.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/cu_mask_probe tools/cu_mask_probe.hip && gpurun_out/cu_mask_probe > gpurun_out/cu_mask_probe.txt
#include <hip/hip_runtime.h>

#include <cstdio>
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

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, words = (cus + 31) / 32;
  printf("# %s, %d CUs, %d mask words\n# bit -> distinct (xcc, se, sa, cu) the 64 workgroups of a stream with only that bit ran on\n", p.name, cus, words);
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
