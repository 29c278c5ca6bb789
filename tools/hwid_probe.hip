// Where do the wavefronts of a workgroup land?  Prints, for workgroups of 256 / 768 / 1024 threads, the
// SIMD and CU of every wavefront (HW_REG_HW_ID, gfx9 layout: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
// and the cost of an s_barrier loop at 4 / 12 / 16 wavefronts.  Information for step_solve.hip's GROUP
// variants; HIP promises none of it.  Build: make -C tools hwid_probe ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(unsigned *out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}

__global__ void barrier_loop(long long *cyc, int n) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  unsigned *d; long long *c;
  CHK(hipMalloc(&d, sizeof(unsigned) * 16 * 4));
  CHK(hipMalloc(&c, sizeof(long long)));
  std::printf("{");
  bool first = true;
  for (int threads : {256, 768, 1024}) {
    CHK(hipMemset(d, 0xff, sizeof(unsigned) * 16 * 4));
    hipLaunchKernelGGL(probe, dim3(3), dim3(threads), 0, 0, d);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned> h(16 * 4);
    CHK(hipMemcpy(h.data(), d, sizeof(unsigned) * 16 * 4, hipMemcpyDeviceToHost));
    std::printf("%s\"simd_of_wave_%d\": [", first ? "" : ", ", threads);
    first = false;
    for (int b = 0; b < 3; ++b) {
      std::printf("%s[", b ? ", " : "");
      for (int w = 0; w < threads / 64; ++w) std::printf("%s%u", w ? ", " : "", (h[b * 16 + w] >> 4) & 3);
      std::printf("]");
    }
    std::printf("], \"cu_of_block_%d\": [", threads);
    for (int b = 0; b < 3; ++b) std::printf("%s%u", b ? ", " : "", (h[b * 16] >> 8) & 0xff);
    std::printf("]");
    const int n = 100000;
    long long hc = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(barrier_loop, dim3(1), dim3(threads), 0, 0, c, n);
      CHK(hipDeviceSynchronize());
    }
    CHK(hipMemcpy(&hc, c, sizeof(hc), hipMemcpyDeviceToHost));
    std::printf(", \"barrier_ticks_%d_threads\": %.1f", threads, (double)hc / n);
  }
  std::printf("}\n");
  return 0;
}
