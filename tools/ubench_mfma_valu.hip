// Microbenchmark: how much VALU work hides behind one fp32 MFMA on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o tools/ubench_mfma_valu
// For each MFMA shape and for NV = VALU ops per MFMA, runs a dependent MFMA
// chain with NV independent VALU ops (v_fma or v_exp) issued between MFMAs,
// with W waves per SIMD, and prints cycles per MFMA (s_memtime, wave 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE, int NV, int KIND>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float a, float b) {
  __shared__ unsigned long long wmax, wmin;
  if (threadIdx.x == 0) { wmax = 0; wmin = ~0ull; }
  f32x16 acc = {0};
  f32x4 acc4a = {0}, acc4b = {0};
  float v[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = a * (i + 1) + threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (SHAPE == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      else if (SHAPE == 1) {
        acc4a = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4a, 0, 0, 0);
        acc4b = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc4b, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 12]) : "v"(a), "v"(b));
        else v[i % 12] = __builtin_amdgcn_exp2f(v[i % 12]);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  s += acc4a[0] + acc4b[1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { atomicMax(&wmax, t1 - t0); atomicMin(&wmin, t1 - t0); }
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = wmax; cyc[1] = wmin; }
}

template <int SHAPE, int NV, int KIND>
void run(int waves_per_simd) {
  float *out;
  unsigned long long *cyc, h[2];
  const int threads = 256 * waves_per_simd;
  hipMalloc(&out, sizeof(float) * threads * 256);
  hipMalloc(&cyc, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  k<SHAPE, NV, KIND><<<256, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  hipEventRecord(e0, 0);
  k<SHAPE, NV, KIND><<<256, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  const double slots = iters * 8.0 * waves_per_simd;   // MFMA(+NV valu) slots executed per SIMD
  printf("shape=%-9s kind=%s NV=%2d waves/SIMD=%d  cyc/slot/SIMD: slowest wave %.1f fastest %.1f | kernel %.3f ms = %.1f cyc/slot @2.4GHz\n",
         SHAPE == 0 ? "32x32x2" : (SHAPE == 1 ? "2x16x16x4" : "none"), KIND == 0 ? "fma" : "exp", NV,
         waves_per_simd, h[0] / slots, h[1] / slots, ms, ms * 1e-3 * 2.4e9 / slots);
  hipFree(out);
  hipFree(cyc);
}

#define SWEEP(SHAPE, KIND, W)                                                              \
  run<SHAPE, 0, KIND>(W); run<SHAPE, 2, KIND>(W); run<SHAPE, 4, KIND>(W);  \
  run<SHAPE, 8, KIND>(W); run<SHAPE, 16, KIND>(W);

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    SWEEP(0, 0, w)
    SWEEP(0, 1, w)
    SWEEP(1, 0, w)
    SWEEP(2, 0, w)
    SWEEP(2, 1, w)
  }
  return 0;
}
