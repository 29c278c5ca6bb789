// microbench.hip -- measures, on the MI355X it runs on, the latencies the solve kernels'
// latency model (bench.py `roofline`) is built from:
//   * dependent / independent v_fma_f64 (cycles per instruction, one wavefront),
//   * dependent ds_read_b32 (LDS round trip),
//   * ONE projected Gauss-Seidel constraint update of the 1-lane tile kernel -- ticket poll,
//     accumulator loads, the three row residuals, the projected 3x3 block solve, both
//     accumulator updates, stores, ticket stores -- executed back to back by one lane that is
//     always ready (the device functions are the kernels' own: solve_device.h), alone on a CU
//     and with 1..4 wavefronts per SIMD each running its own chain (issue sharing).
// Prints one JSON object.  Build: make -C tools ; run on the GPU box: tools/microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../eggshell_amd/csrc/solve_device.h"

using namespace egs;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void fma_dep(double *out, long long *cyc, int n, double a, double b) {
  double x = out[threadIdx.x];
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = __builtin_fma(x, a, b);
  }
  asm volatile("" :: "v"(x));
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

__global__ void fma_indep(double *out, long long *cyc, int n, double a, double b) {
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = out[threadIdx.x] + k;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = __builtin_fma(x[k], a, b);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

__global__ void lds_chase(int *out, long long *cyc, int n) {
  __shared__ int next[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) next[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  int p = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) p = next[p];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}

// One always-ready constraint update per wavefront and iteration, `active` lanes per wavefront.
template <bool ISO, int MAXT>
__global__ void __launch_bounds__(MAXT) update_chain(double *out, long long *cyc, int n, int active, double cfm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int waves = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  double *s_acc = reinterpret_cast<double *>(smem);                      // [waves][64][2][6]
  unsigned *s_tick = reinterpret_cast<unsigned *>(s_acc + (size_t)waves * 64 * 12);
  const int slot0 = (wave * 64 + lane) * 2, slot1 = slot0 + 1;
  for (int k = 0; k < 6; ++k) { s_acc[slot0 * 6 + k] = 0.01 * k; s_acc[slot1 * 6 + k] = -0.02 * k; }
  s_tick[slot0] = 0; s_tick[slot1] = 0;
  Cons<double> c;
  for (int k = 0; k < 18; ++k) {
    c.J0[k] = 0.1 * ((k * 7 + lane) % 11) - 0.5; c.J1[k] = 0.1 * ((k * 5 + lane) % 13) - 0.6;
    c.B0[k] = 0.9 * c.J0[k]; c.B1[k] = 1.1 * c.J1[k];
  }
  for (int k = 0; k < 9; ++k) c.D[k] = (k % 4 == 0) ? 2.5 : 0.1;
  c.wl0 = 1.0; c.wa0 = 10.0; c.wl1 = 1.0; c.wa1 = 10.0;
  for (int r = 0; r < 3; ++r) { c.inv[r] = 0.4; c.rhs[r] = 0.3 + r; c.lo[r] = -1.0; c.hi[r] = 1.0; c.eq[r] = false; }
  double x[3] = {0.1, 0.2, 0.3};
  __syncthreads();
  const unsigned tk0 = lds_addr(s_tick + slot0), tk1 = lds_addr(s_tick + slot1);
  const unsigned ac0 = lds_addr(s_acc + slot0 * 6), ac1 = lds_addr(s_acc + slot1 * 6);
  unsigned want = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (lane < active) {
    for (int i = 0; i < n; ++i) {
      unsigned t0v, t1v;
      double a0[6], a1[6];
      poll_ticks(tk0, tk1, t0v, t1v);
      const bool ready = t0v == want && t1v == want;
      if (ready) {
        load12(ac0, ac1, a0, a1);
        double res[3], dx[3] = {0, 0, 0};
        row_residuals(c, a0, a1, x, cfm, res);
        update_rows<double, 1>(c, res, x, dx);
        acc_add_side0<ISO>(a0, c, dx); store6(ac0, a0);
        acc_add_side1<ISO>(a1, c, dx); store6(ac1, a1);
        store_tick(tk0, want + 1); store_tick(tk1, want + 1);
        ++want;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + want;
  if (lane == 0) cyc[blockIdx.x * waves + wave] = t1 - t0;
}

int main() {
  double *d_out; long long *d_cyc; int *d_iout;
  CHK(hipMalloc(&d_out, sizeof(double) * 1 << 16));
  CHK(hipMalloc(&d_iout, sizeof(int) * 1024));
  CHK(hipMalloc(&d_cyc, sizeof(long long) * 64));
  CHK(hipMemset(d_out, 0, sizeof(double) * 1 << 16));
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  const int n = 2048;
  auto rd = [&](int k) { std::vector<long long> h(64); CHK(hipMemcpy(h.data(), d_cyc, sizeof(long long) * 64, hipMemcpyDeviceToHost)); return h[k]; };
  std::printf("{\"device\": \"%s\", \"cu\": %d, \"clock_mhz\": %d", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
  for (int rep = 0; rep < 2; ++rep) {   // second pass is the warm one
    hipLaunchKernelGGL(fma_dep, dim3(1), dim3(64), 0, 0, d_out, d_cyc, n, 1.0000001, 1e-9); CHK(hipDeviceSynchronize());
    if (rep) std::printf(", \"fma_f64_dependent_cycles\": %.2f", (double)rd(0) / (n * 16.0));
    hipLaunchKernelGGL(fma_indep, dim3(1), dim3(64), 0, 0, d_out, d_cyc, n, 1.0000001, 1e-9); CHK(hipDeviceSynchronize());
    if (rep) std::printf(", \"fma_f64_issue_cycles_one_wave\": %.2f", (double)rd(0) / (n * 16.0));
    hipLaunchKernelGGL(fma_indep, dim3(1), dim3(1024), 0, 0, d_out, d_cyc, n, 1.0000001, 1e-9); CHK(hipDeviceSynchronize());
    if (rep) std::printf(", \"fma_f64_issue_cycles_4_waves_per_simd\": %.2f", (double)rd(0) / (n * 16.0));
    hipLaunchKernelGGL(lds_chase, dim3(1), dim3(64), 0, 0, d_iout, d_cyc, n); CHK(hipDeviceSynchronize());
    if (rep) std::printf(", \"ds_read_b32_dependent_cycles\": %.2f", (double)rd(0) / (n * 16.0));
  }
  // s_memtime ticks at 100 MHz on gfx9?  report both raw ticks and the ratio to shader cycles
  // registers: 232 VGPRs (stored B) allow 2 wavefronts per SIMD, 164 (isotropic variant) 3
  for (int iso = 0; iso < 2; ++iso)
    for (int waves : {1, 4, 8, 12}) {
      if (!iso && waves > 8) continue;
      for (int active : {1, 8, 64}) {
        const size_t lds = (size_t)waves * 64 * 12 * sizeof(double) + (size_t)waves * 64 * 2 * sizeof(unsigned);
        for (int rep = 0; rep < 2; ++rep) {
          if (iso) hipLaunchKernelGGL((update_chain<true, 768>), dim3(1), dim3(64 * waves), lds, 0, d_out, d_cyc, n, active, 0.01);
          else hipLaunchKernelGGL((update_chain<false, 512>), dim3(1), dim3(64 * waves), lds, 0, d_out, d_cyc, n, active, 0.01);
          CHK(hipDeviceSynchronize());
        }
        long long worst = 0;
        for (int w = 0; w < waves; ++w) worst = rd(w) > worst ? rd(w) : worst;
        std::printf(", \"update_%s_w%d_a%d_ticks\": %.1f", iso ? "iso" : "reg", waves, active, (double)worst / n);
      }
    }
  // wall-clock calibration of the s_memtime tick: a long dependent chain timed with hipEvents
  {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int big = 1 << 18;
    hipLaunchKernelGGL(fma_dep, dim3(1), dim3(64), 0, 0, d_out, d_cyc, big, 1.0000001, 1e-9); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(fma_dep, dim3(1), dim3(64), 0, 0, d_out, d_cyc, big, 1.0000001, 1e-9);
    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::printf(", \"memtime_ticks_per_us\": %.2f, \"fma_f64_dependent_ns\": %.3f", (double)rd(0) / (ms * 1e3), ms * 1e6 / (big * 16.0));
  }
  std::printf("}\n");
  return 0;
}
