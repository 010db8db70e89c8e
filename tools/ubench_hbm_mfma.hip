// Do HBM streaming and the fp32 matrix pipe overlap at full rate on gfx950?  (Behind DESIGN section 5 K4d: the one-position
// bit-wise kernel's load side alone runs at 0.21 ms, its MFMA side alone at 0.18 ms, together 0.25 ms.)
// One persistent workgroup of 16 waves per CU: waves 0-7 stream a buffer from HBM (16-byte loads, fully coalesced, 8 loads
// in flight per wave), waves 8-15 run dependent v_mfma_f32_32x32x2_f32 chains -- independent work, nothing shared.
//   mode 0: the streaming waves alone; mode 1: the MFMA waves alone; mode 2: both.
// Variants: a streaming loop WITHOUT any vector-ALU instruction (inline asm); the two kinds of waves on separate SIMDs;
// 16-bit MFMA chains (v_mfma_f32_32x32x16_f16) instead of fp32 ones.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm_mfma.hip -o tools/ubench_hbm_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__global__ __launch_bounds__(1024) void k(const u32x4 *buf, size_t n_vec, int mfma_iters, int mode, unsigned *out, int novalu, int split, int f16, int prio) {
  const int wave0 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // split 0: waves 0-7 stream (two streaming + two MFMA waves on every SIMD: waves sit on SIMD wave % 4);
  // split 1: the waves of SIMDs 0 and 1 stream, those of SIMDs 2 and 3 run MFMAs (four of a kind per SIMD)
  const bool streams = split ? (wave0 & 3) < 2 : wave0 < 8;
  const int wave = split ? ((wave0 >> 2) * 2 + (wave0 & 1)) : (wave0 & 7);   // 0..7 within its kind
  if (streams) {
    if (mode == 1) return;
    if (prio) __builtin_amdgcn_s_setprio(3);   // does a higher issue priority get the streaming waves their turns?
    if (novalu) {
      // the same stream with NO vector-ALU instruction in the loop: scalar-addressed buffer loads whose results are
      // never read (inline asm), scalar loop counter -- the vector ALU port is what fp32 MFMAs do not share
      const size_t per_wave = n_vec * 16 / ((size_t)gridDim.x * 8);            // bytes, a multiple of 8 KB
      const char *base = reinterpret_cast<const char *>(buf) + ((size_t)blockIdx.x * 8 + wave) * per_wave;
      const unsigned long long ba = reinterpret_cast<unsigned long long>(base);
      u32x4 rsrc;   // raw buffer descriptor: base, stride 0, num_records (bytes), flags
      rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)ba);
      rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ba >> 32) & 0xffffu);
      rsrc[2] = __builtin_amdgcn_readfirstlane((unsigned)per_wave);
      rsrc[3] = 0x00020000u;
      const unsigned voff = lane * 16u;
      for (unsigned off = 0; off + 8192 <= per_wave;) {   // the asm advances `off` by 8192
        asm volatile(
            "buffer_load_dwordx4 v[40:43], %1, %2, %0 offen\n"
            "buffer_load_dwordx4 v[44:47], %1, %2, %0 offen offset:1024\n"
            "buffer_load_dwordx4 v[48:51], %1, %2, %0 offen offset:2048\n"
            "buffer_load_dwordx4 v[52:55], %1, %2, %0 offen offset:3072\n"
            "s_add_u32 %0, %0, 4096\n"
            "buffer_load_dwordx4 v[56:59], %1, %2, %0 offen\n"
            "buffer_load_dwordx4 v[60:63], %1, %2, %0 offen offset:1024\n"
            "buffer_load_dwordx4 v[64:67], %1, %2, %0 offen offset:2048\n"
            "buffer_load_dwordx4 v[68:71], %1, %2, %0 offen offset:3072\n"
            "s_add_u32 %0, %0, 4096\n"
            "s_waitcnt vmcnt(4)\n"
            : "+s"(off) : "v"(voff), "s"(rsrc)
            : "memory", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55",
              "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(out) + 1, wall_clock64());
      return;
    }
    // 8 streaming waves per CU: wave-contiguous 1 KB pieces, grid-strided
    const size_t stride = (size_t)gridDim.x * 8 * 64;
    size_t i = ((size_t)blockIdx.x * 8 + wave) * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + 7 * stride < n_vec; i += 8 * stride) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(buf + i + j * stride);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
    if (lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(out) + 1, wall_clock64());
  } else {
    if (mode == 0) return;
    f32x16 a = {0};
    float x = 1.0f + lane * 1e-9f, y = 0.5f;
    if (f16) {   // the same chains on the 16-bit matrix path (v_mfma_f32_32x32x16_f16: 8 passes, co-executes with the vector ALU)
      using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
      f16x8 hx, hy;
      for (int e = 0; e < 8; ++e) hx[e] = (_Float16)x, hy[e] = (_Float16)y;
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, a, 0, 0, 0);
      }
    } else
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
    }
    if (a[0] == 123.456f) out[1] = 1;
    if (lane == 0) atomicMax(reinterpret_cast<unsigned long long *>(out) + 2, wall_clock64());
  }
  if (threadIdx.x == 0) atomicMin(reinterpret_cast<unsigned long long *>(out) + 3, wall_clock64() - 0);
}

int main() {
  const size_t bytes = (size_t)4 << 30;   // 4 GiB: far beyond the 256 MB Infinity Cache
  u32x4 *buf; unsigned *out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64);
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int novalu = 0, split = 0, f16 = 0, prio = 0;
  auto run = [&](int mode, int mfma_iters) {
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, buf, bytes / 16, mfma_iters, mode, out, novalu, split, f16, prio);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, buf, bytes / 16, mfma_iters, mode, out, novalu, split, f16, prio);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
  };
  // 8 MFMA waves per CU = 2 per SIMD; one iteration = 16 MFMAs x 64 cycles; choose the count so that the MFMA side alone
  // takes about as long as the streaming side alone
  const float t_load = run(0, 0);
  int iters = 1000;
  float t_mfma = run(1, iters);
  iters = (int)(iters * t_load / t_mfma);
  for (int rep = 0; rep < 16; ++rep) {
    novalu = (rep & 3) >= 2;
    split = rep >= 4 && rep < 8;
    f16 = rep >= 8 && rep < 12;
    prio = rep >= 12;
    if (rep == 12) {  // fp32 chains, two of each kind per SIMD, the streaming waves at s_setprio 3: recalibrate
      const float tl = run(0, 0), tm = run(1, 1000);
      iters = (int)(1000 * tl / tm);
    }
    if (prio) printf("[streaming waves at s_setprio 3] ");
    if (rep == 8) {   // 16-bit chains, two of each kind per SIMD again: recalibrate
      const float tl = run(0, 0), tm = run(1, 1000);
      iters = (int)(1000 * tl / tm);
    }
    if (f16) printf("[16-bit MFMA chains] ");
    if (rep == 4) {   // four MFMA waves on two SIMDs: recalibrate the MFMA side to the streaming side
      const float tl = run(0, 0), tm = run(1, 1000);
      iters = (int)(1000 * tl / tm);
    }
    printf(split ? "[streaming waves on SIMDs 0-1, MFMA waves on SIMDs 2-3] " : "[two streaming + two MFMA waves on every SIMD] ");
    printf(novalu ? "[no vector-ALU instruction in the streaming loop] " : "[streaming loop with 4 v_xor per load + address adds] ");
    const float a = run(0, 0), b = run(1, iters);
    unsigned long long init[4] = {0, 0, 0, ~0ull}, got[4];
    hipMemcpy(out + 2, init + 1, 24, hipMemcpyHostToDevice);
    const float c = run(2, iters);   // (two launches: the maxima are over both, the minimum start is the first launch's)
    hipMemcpy(got + 1, out + 2, 24, hipMemcpyDeviceToHost);
    printf("streaming alone %.3f ms (%.2f TB/s) | MFMA alone %.3f ms (%.0f TFLOP/s) | both %.3f ms = %.2f x the slower side; in the second 'both' launch the last MFMA wave ended %.3f ms before the last streaming wave\n",
           a, bytes / a / 1e9, b, (256.0 * 8 * iters * 16 * (f16 ? 32768.0 : 4096.0)) / (b * 1e-3) / 1e12, c, c / (a > b ? a : b), ((double)got[1] - (double)got[2]) / 1e5);
  }
  return 0;
}
