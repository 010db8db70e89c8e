// Microbenchmark 2: does an LDS-fed fp32 MFMA chain sustain 64 cycles/MFMA/SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_lds.hip -o tools/ubench_mfma_lds
// MODE 0: A operands constant registers, B from a 16-register tile (baseline)
// MODE 1: A operands from LDS (one ds_read_b128 per 4 MFMAs), like the decode kernel
// MODE 2: MODE 1 + the accumulator re-initialised from LDS every 32 MFMAs (bias init)
// MODE 3: MODE 2 + 16 v_med3/cmp/cndmask argmax ops per 16 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned long long wmax;
  if (threadIdx.x == 0) wmax = 0;
  for (int i = threadIdx.x; i < 12288; i += blockDim.x) lds[i] = seed * (i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 h, acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { h[r] = seed + r; acc[r] = 0.f; }
  float best = -1e30f; int bidx = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    int lo = lane; asm volatile("" : "+v"(lo)); lo *= 4;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {          // 4 x 32 MFMAs = 128 per iteration
      if (MODE >= 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(lds + 12000 + 8 * q + 4 * (lane >> 5));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[4 * q + e] = b[e];
        }
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        f32x4 a;
        if (MODE == 0) { a[0] = seed; a[1] = seed * 2; a[2] = seed * 3; a[3] = seed * 4; }
        else a = *reinterpret_cast<const f32x4 *>(lds + (blk * 8 + g) * 256 + lo);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], h[(4 * g + e) & 15], acc, 0, 0, 0);
        if (MODE == 3 && (g & 3) == 3) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float zc = __builtin_amdgcn_fmed3f(acc[r], -9.f, 9.f);
            const bool c = zc > best;
            best = c ? zc : best;
            bidx = c ? r + 16 * g : bidx;
          }
        }
      }
      if (MODE < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(acc[r]));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = best + bidx;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(&wmax, t1 - t0);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = wmax;
}

template <int MODE>
void run(int waves_per_simd) {
  float *out; unsigned long long *cyc, h;
  const int threads = 256 * waves_per_simd, iters = 500;
  hipMalloc(&out, sizeof(float) * threads * 256); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, threads, 49152>>>(out, cyc, iters, 0.001f);
  hipEventRecord(e0, 0);
  k<MODE><<<256, threads, 49152>>>(out, cyc, iters, 0.001f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = iters * 128.0 * waves_per_simd;
  printf("mode %d waves/SIMD %d: %.1f cyc/MFMA/SIMD (slowest wave), kernel %.3f ms = %.1f cyc @2.4GHz\n",
         MODE, waves_per_simd, h / n, ms, ms * 1e-3 * 2.4e9 / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) { run<0>(w); run<1>(w); run<2>(w); run<3>(w); }
  return 0;
}
