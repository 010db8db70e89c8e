// Microbenchmark 3: f16 32x32x16 MFMA (+ VALU co-execution) and packed-f32 VALU cost on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// KIND: 0 none, 1 v_exp_f32, 2 v_fma_f32 (asm), 3 v_pk_fma_f32 (asm, 2 values), 4 v_pk_mul_f32, 5 v_cvt_pkrtz_f16_f32
// MF: 0 no mfma, 1 f16 32x32x16, 2 f32 32x32x2, 3 f16 16x16x32
template <int MF, int NV, int KIND>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float a, float b) {
  __shared__ unsigned long long wmax;
  if (threadIdx.x == 0) wmax = 0;
  f32x16 acc = {0};
  __attribute__((ext_vector_type(4))) float acc4 = {0};
  f16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(a * i); fb[i] = (_Float16)(b + i); }
  float v[12]; f32x2 p[6];
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = a * (i + 1) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 6; ++i) { p[i][0] = v[2 * i]; p[i][1] = v[2 * i + 1]; }
  f32x2 pa = {a, a}, pb = {b, b};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MF == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
      if (MF == 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if (MF == 3) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc4, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (KIND == 1) v[i % 12] = __builtin_amdgcn_exp2f(v[i % 12]);
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i % 12]) : "v"(a), "v"(b));
        if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i % 6]) : "v"(pa), "v"(pb));
        if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i % 6]) : "v"(pa));
        if (KIND == 5) { unsigned r; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(v[i % 12]), "v"(v[(i + 1) % 12])); v[i % 12] = __uint_as_float(r); }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc4[0];
#pragma unroll
  for (int i = 0; i < 12; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) s += p[i][0] + p[i][1];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(&wmax, t1 - t0);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = wmax;
}

// mixed: even waves run f16 MFMA chains, odd waves run fp32 MFMA chains (co-execution of the two pipes?)
__global__ __launch_bounds__(1024) void kmix(float *out, unsigned long long *cyc, int iters, float a, float b) {
  __shared__ unsigned long long wmax;
  if (threadIdx.x == 0) wmax = 0;
  f32x16 acc = {0};
  f16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(a * i); fb[i] = (_Float16)(b + i); }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((wave >> 2) & 1) {
    for (int it = 0; it < iters * 8; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  } else {
    for (int it = 0; it < iters * 16; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(&wmax, t1 - t0);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = wmax;
}

template <int MF, int NV, int KIND>
void run(int w) {
  float *out; unsigned long long *cyc, h;
  const int threads = 256 * w, iters = 2000;
  hipMalloc(&out, sizeof(float) * threads * 256); hipMalloc(&cyc, 8);
  k<MF, NV, KIND><<<256, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  k<MF, NV, KIND><<<256, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const char *mf[] = {"none", "f16 32x32x16", "f32 32x32x2", "f16 16x16x32"};
  const char *kd[] = {"-", "exp", "fma", "pk_fma", "pk_mul", "cvt_pkrtz"};
  printf("mfma=%-13s valu=%-9s NV=%2d waves/SIMD=%d : %.1f cyc per slot per SIMD\n", mf[MF], kd[KIND], NV, w,
         (double)h / (iters * 8.0 * w));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<1, 0, 0>(w); run<3, 0, 0>(w);
    run<1, 2, 1>(w); run<1, 4, 1>(w); run<1, 8, 1>(w);
    run<1, 4, 2>(w); run<1, 8, 2>(w); run<1, 16, 2>(w);
    run<0, 8, 2>(w); run<0, 8, 3>(w); run<0, 8, 4>(w); run<0, 8, 5>(w);
    run<2, 8, 2>(w); run<2, 8, 3>(w); run<2, 8, 4>(w); run<2, 8, 5>(w);
    run<1, 8, 3>(w); run<1, 8, 5>(w);
  }
  {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) kmix<<<256, 512>>>(out, cyc, 2000, 1.0f, 0.5f);
    hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("mix: per SIMD one wave 16000 f32 MFMAs (x64 = %d cyc) + one wave 32000 f16 MFMAs: slowest wave %llu cycles\n", 16000 * 64, h);
  }
  return 0;
}
