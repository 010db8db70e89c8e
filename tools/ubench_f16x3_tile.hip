// Microbenchmark (round 5): does another MFMA shape or a packed argmax shorten K1's f16x3 tile?
//
// One 32-row tile of the config-3 network in f16x3 mode is, per wave (PMC, profiles/r04/pmc_summary.json):
//   120 v_mfma_f32_32x32x16_f16 + 811 VALU instructions, of which 224 quarter-rate transcendentals
//   (112 v_exp_f32 + 112 v_rcp_f32) and 587 full-rate ones (112 v_add, 224 operand-split ops, ~136 argmax, rest).
// The kernel below issues exactly that mix (one eighth of a tile per loop iteration, MFMAs and VALU work interleaved
// with independent dependency chains, as in the real kernel) at 1..4 waves per SIMD, and the same mix with
//   B: the matrix work as 240 v_mfma_f32_16x16x32_f16 (same FLOPs, half the accumulator registers per MFMA),
//   C: 64 of the full-rate ops (the argmax max-tree: v_max3_f32 over 128 class values per lane) replaced by the
//      packed-f16 form (64 v_cvt_pk_f16_f32 + 64 v_pk_max_f16: what "packed class values" would issue),
//   D: no MFMAs at all (the VALU stream alone: the floor any MFMA shape can reach),
//   E: no VALU at all (the matrix pipe alone).
// Prints cycles per tile and SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_f16x3_tile tools/ubench_f16x3_tile.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// MF: 0 none, 1 32x32x16 (15 per eighth), 2 16x16x32 (30 per eighth).  VALU: 0 none, 1 today's mix, 2 packed argmax
template <int MF, int VALU>
__global__ __launch_bounds__(1024) void tile_mix(float *out, unsigned long long *cyc, int iters, float a, float b) {
  __shared__ unsigned long long wmax;
  if (threadIdx.x == 0) wmax = 0;
  f32x16 acc[2] = {{0}, {0}};
  f32x4 acc4[4] = {{0}, {0}, {0}, {0}};
  f16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(a * i); fb[i] = (_Float16)(b + i); }
  float t[14], s[16];
#pragma unroll
  for (int i = 0; i < 14; ++i) t[i] = a * 0.01f * (i + 1) + 1e-3f * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = b * (i + 1) + threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // one eighth of a tile: 15 (30) MFMAs, 28 transcendentals, 73 full-rate ops, interleaved 1 : ~2 : ~5
#pragma unroll
    for (int u = 0; u < 15; ++u) {
      if (MF == 1) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[u & 1], 0, 0, 0);
      if (MF == 2) {
        acc4[(2 * u) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc4[(2 * u) & 3], 0, 0, 0);
        acc4[(2 * u + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc4[(2 * u + 1) & 3], 0, 0, 0);
      }
      if (VALU != 0) {
        if (u < 14) {  // 14 activations per eighth: exp2, add, rcp (the hidden layers' r = 1 / (1 + 2^zeta))
          float e = __builtin_amdgcn_exp2f(t[u]);
          asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e));
          t[u] = __builtin_amdgcn_rcpf(e);
        }
        // 59 more full-rate ops per eighth (4 per slot, 3 in the last): split, argmax, addressing stand-ins
        const int nfull = u < 14 ? 4 : 3;
#pragma unroll
        for (int i = 0; i < nfull; ++i) {
          const int q = (4 * u + i) & 15;
          const bool tree = VALU == 2 && ((4 * u + i) % 9) == 0;   // 8 of the 73: the max-tree's share per eighth
          if (!tree) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[q]) : "v"(a), "v"(b));
          else {  // packed form: convert two class values to f16, packed max (2 instructions where v_max3 was 1)
            unsigned r;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(s[q]), "v"(s[(q + 1) & 15]));
            asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r) : "v"(r));
            s[q] = __uint_as_float(r);
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 14; ++i) sum += t[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += s[i] + acc[0][i] + acc[1][i];
#pragma unroll
  for (int i = 0; i < 4; ++i) sum += acc4[i][0] + acc4[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) atomicMax(&wmax, t1 - t0);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = wmax;
}

template <int MF, int VALU>
void run(int w, const char *what) {
  float *out; unsigned long long *cyc, h;
  const int threads = 256 * w, iters = 8 * 500;   // 500 tiles per wave
  hipMalloc(&out, sizeof(float) * threads * 256); hipMalloc(&cyc, 8);
  tile_mix<MF, VALU><<<256, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  tile_mix<MF, VALU><<<256, threads>>>(out, cyc, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md)
  printf("%-64s waves/SIMD=%d : %8.0f cycles per tile and SIMD\n", what, w, (double)h / (500.0 * w));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; ++w) {
    run<1, 1>(w, "A today: 120 x 32x32x16 + 811 VALU (224 quarter-rate)");
    run<2, 1>(w, "B 240 x 16x16x32 + the same VALU");
    run<1, 2>(w, "C 120 x 32x32x16 + VALU with the max-tree in packed f16");
    run<0, 1>(w, "D the VALU stream alone");
    run<1, 0>(w, "E 120 x 32x32x16 alone");
    run<2, 0>(w, "F 240 x 16x16x32 alone");
  }
  return 0;
}
