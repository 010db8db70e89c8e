// Microbenchmark behind DESIGN section 5 (K5, priced ceiling; docs/HISTORY.md section 9 item 7): one LSTM(20) time step for 32 sequences as K5 issues it --
// 3 tiles x 11 dependent v_mfma_f32_32x32x2_f32 + the cell update of 10 units per lane (100 transcendentals) -- against
// the exact tiling a 16x16x4 layout would allow: 5 tiles x 2 column tiles x 6 dependent v_mfma_f32_16x16x4_f32, the same
// cell update, + the projection on the vector ALU (5 FMAs + 2 lane exchanges + 2 adds per column tile).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize tools/ubench_lstm_step.hip -o tools/ubench_lstm_step
// Arithmetic is NOT an LSTM (weights are arbitrary registers); only the instruction mix, the dependences and the
// register pressure are reproduced.  Prints ms per 512 steps for both shapes at 4 waves per SIMD on every CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float cell(float zi, float zj, float zf, float zo, float &c) {
  const float si = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zi));
  const float sf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zf));
  const float so = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zo));
  const float rj = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zj));
  const float tj = __builtin_fmaf(-5.77f, rj, 2.885f);
  const float cn = __builtin_fmaf(si, tj, sf * c);
  const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn));
  c = cn;
  return so * __builtin_fmaf(-2.0f, rc, 1.0f);
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float *win, const float *xin, float *out, int steps) {
  const int lane = threadIdx.x & 63;
  float w[33];
#pragma unroll
  for (int i = 0; i < 33; ++i) w[i] = win[i * 64 + lane] * 0.01f;
  float c[10], h[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) c[i] = 0.f, h[i] = 0.f;
  float acc_out = 0.f;
  for (int t = 0; t < steps; ++t) {
    const float x = xin[(size_t)t * 64 + lane];
    if (SHAPE == 0) {
      f32x16 acc[3];
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[T * 11], h[0], f32x16{0}, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 10; ++s) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[T * 11 + s], h[s], acc[T], 0, 0, 0);
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[T * 11 + 10], x, acc[T], 0, 0, 0);
      }
      acc_out += acc[2][8];   // the projection slot
#pragma unroll
      for (int u = 0; u < 10; ++u)
        h[u] = cell(acc[u / 16][u % 16], acc[(10 + u) / 16][(10 + u) % 16], acc[(20 + u) / 16][(20 + u) % 16],
                    acc[(30 + u) / 16][(30 + u) % 16], c[u]);
    } else {
      f32x4 acc[2][5];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int T = 0; T < 5; ++T) {
          acc[ct][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[T * 6], h[ct * 5 + 0], f32x4{0}, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 5; ++s) acc[ct][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[T * 6 + s], h[ct * 5 + s], acc[ct][T], 0, 0, 0);
          acc[ct][T] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[T * 6 + 5], x, acc[ct][T], 0, 0, 0);
        }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float p = 0.f;
#pragma unroll
        for (int T = 0; T < 5; ++T) {
          h[ct * 5 + T] = cell(acc[ct][T][0], acc[ct][T][1], acc[ct][T][2], acc[ct][T][3], c[ct * 5 + T]);
          p = __builtin_fmaf(w[30 + (T % 3)], h[ct * 5 + T], p);
        }
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        acc_out += p;
      }
    }
  }
  float s = acc_out;
#pragma unroll
  for (int i = 0; i < 10; ++i) s += h[i] + c[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
float run(const float *w, const float *x, float *out, int steps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, w, x, out, steps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float *w, *x, *out;
  const int steps = 512;
  hipMalloc(&w, 33 * 64 * 4); hipMalloc(&x, (size_t)steps * 64 * 4); hipMalloc(&out, (size_t)256 * 4 * 256 * 4);
  hipMemset(w, 0x3c, 33 * 64 * 4); hipMemset(x, 0x3c, (size_t)steps * 64 * 4);
  for (int rep = 0; rep < 2; ++rep) {
    const float a = run<0>(w, x, out, steps), b = run<1>(w, x, out, steps);
    printf("512 steps x 4 waves/SIMD x 256 CUs: 32x32x2 layout (33 MFMAs/step) %.3f ms; 16x16x4 layout (60 MFMAs/step + projection on the VALU) %.3f ms; ratio %.3f\n", a, b, b / a);
  }
  return 0;
}
