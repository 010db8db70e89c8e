// K4c: the bit-wise decoder BANK with the received rows RESIDENT in LDS and every position's first layer STREAMED
// from L2 / Infinity Cache straight into the MFMA A operand (round 4).
//
//   src/fnn_viterbi_bitwise.py:326-399 build_network   2(6+L) -> n_hidden (30) -> 2, ReLU after both layers
//   src/fnn_viterbi_bitwise.py:503-505 argmax eval     one trained network per bit_position, evaluated one per run
//
// K4b (nnv_dense.hip, dense_tanh_kernel<..., HEAD>) runs the bank as one LDS-tiled GEMM over the stacked first
// layers: both operands staged global -> VGPR -> LDS behind two barriers per k-tile, the 128 x 128 tile passed
// through LDS once more for the heads; matrix pipe 85-87 % busy, 0.78 of the fp32-MFMA peak, 61 GB fetched into
// the L2s per launch.  What K6 and the stream variant of K1 measured this round -- the L2s deliver 16 B per cycle
// and CU to scalar-addressed buffer loads without the matrix pipe noticing -- allows the other assignment:
//  * a persistent workgroup of 16 waves keeps ONE tile of 32 received rows in LDS, in MFMA B-operand order
//    [k-quad][row][4 floats] (1036 inputs: 130 KB), staged once per 32 rows;
//  * wave w evaluates positions w, w + 16, ...: per position the 32-neuron first layer is 130 fragment groups
//    of 1 KB (packed on the host in A-operand order), pulled four groups ahead by buffer loads; per group one
//    conflict-free ds_read_b128 of the rows and four v_mfma_f32_32x32x2_f32 (same k order as K4b: identical hidden
//    values); no operand staging, no barrier between the staging of two row tiles;
//  * the head runs in registers: the other lane half's 16 hidden values come over by 16 lane exchanges, then the
//    sequential 32-term fmaf chain of K4b's head (identical logits), bias, ReLU, first-maximum over the 2 outputs,
//    one byte per (position, row).
// All 256 workgroups walk the positions in the same order, so a weight fragment is fetched into an XCD's L2 about
// once per pass over the rows (66 MB x 37 passes x 8 XCDs = 20 GB) instead of once per 128-row block.
#include <stdlib.h>

#include <mutex>

#include "nnv_internal.h"

namespace nnv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
#ifndef NNV_BANK_STREAM_DEPTH
#define NNV_BANK_STREAM_DEPTH 4
#endif
// weight fragment groups in flight per wave.  The 66 MB of first layers live in the Infinity Cache / HBM, not in the
// L2s (K6's 3.5 MB do): same box, 512 positions x 300 k rows: 1 group ahead 75.0 ms, 2: 72.9, 4: 69.5, 8: 69.5-70.2,
// 16: 69.2 ms (K4b: 77.7)
constexpr int kDepth = NNV_BANK_STREAM_DEPTH;

__global__ __launch_bounds__(kThreads) void bank_stream_kernel(const BankStreamParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [quad][32 rows][4 floats]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const unsigned lane_off = static_cast<unsigned>(lane) * 16u;
  const int groups = p.groups;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);

  for (int64_t blk = blockIdx.x; blk < p.n_blocks; blk += gridDim.x) {
    const int64_t row0 = blk * 32;
    // ---------------- 32 rows -> LDS in B-operand order -------------------------------------------------
    {
      const int nq = 2 * groups;
      for (int it = tid; it < nq * 32; it += kThreads) {
        const int q = it >> 5, rr = it & 31;       // consecutive threads: consecutive rows, one contiguous LDS span
        int64_t row = row0 + rr;
        row = row < p.n_rows ? row : p.n_rows - 1;  // tail rows: valid memory, results discarded
        const int k = 4 * q;
        const float *src = p.x + row * p.row_stride + k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k + 3 < p.K) {
          if (p.x16) {
            v = *reinterpret_cast<const f32x4 *>(src);
          } else {                                  // rows of an even length: 8-byte aligned
            const f32x2 t0 = *reinterpret_cast<const f32x2 *>(src);
            const f32x2 t1 = *reinterpret_cast<const f32x2 *>(src + 2);
            v[0] = t0[0], v[1] = t0[1], v[2] = t1[0], v[3] = t1[1];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < p.K) v[e] = src[e];
        }
        *reinterpret_cast<f32x4 *>(lds + static_cast<int64_t>(it) * 4) = v;
      }
    }
    __syncthreads();

    const int64_t row = row0 + r;
    const bool live = row < p.n_rows;
    for (int pos = wave; pos < p.n_pos; pos += kWaves) {     // wave-uniform
      // ---------------- first layer of position `pos`: 32 neurons x 32 rows ----------------------------
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int base = pos * groups * 1024;                   // byte offset of this position's fragment groups
      f32x4 a[kDepth];
#pragma unroll
      for (int i = 0; i < kDepth; ++i)
        a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, base + i * 1024, 0));
      const float *bp = lds + (h * 32 + r) * 4;               // quad 2g + h, row r
      f32x4 b = *reinterpret_cast<const f32x4 *>(bp);
      auto step = [&](const f32x4 av, int g) {
        const f32x4 c = b;
        b = *reinterpret_cast<const f32x4 *>(bp + (g + 1) * (2 * 32 * 4));   // one group ahead (past the end: valid LDS, unused)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c[e], acc, 0, 0, 0);
      };
      int g = 0;
      for (; g + kDepth <= groups; g += kDepth) {
#pragma unroll
        for (int i = 0; i < kDepth; ++i) {
          step(a[i], g + i);
          // refilled AFTER its consumers: the same registers (no copies); past this position's end the load runs into
          // the next position's first groups (unused), past the bank's end the range check returns 0
          a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, base + (g + i + kDepth) * 1024, 0));
        }
      }
#pragma unroll
      for (int i = 0; i < kDepth - 1; ++i)
        if (g + i < groups) step(a[i], g + i);

      // ---------------- + bias, ReLU (matmul, add, relu: src/fnn_viterbi_bitwise.py:392-393) ------------
      // accumulator register 4q + e <-> hidden neuron 8q + 4h + e
      const float *bl = p.bias + static_cast<int64_t>(pos) * 32 + 4 * h;
      float hid[32];   // hid[n] = hidden neuron n of this lane's row -- complete in lane half 0 only, which decides
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bl + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[4 * q + e] + bv[e];
          v = v > 0.f ? v : 0.f;
          hid[8 * q + e] = v;                           // half 0: its own neuron 8q + e
          hid[8 * q + 4 + e] = __shfl_xor(v, 32);       // half 0: the other half's neuron 8q + 4 + e (half 1: unused)
        }
      }
      // ---------------- head: relu(h . Wout + bout), first maximum (:396-397, :504-505) -------------------
      const float *hw = p.head_w + static_cast<int64_t>(pos) * 64;   // [32 hidden][2 outputs], wave-uniform
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int n = 0; n < 32; ++n) {
        s0 = __builtin_fmaf(hid[n], hw[2 * n], s0);
        s1 = __builtin_fmaf(hid[n], hw[2 * n + 1], s1);
      }
      float v0 = s0 + p.head_b[2 * pos], v1 = s1 + p.head_b[2 * pos + 1];
      v0 = v0 > 0.f ? v0 : 0.f;
      v1 = v1 > 0.f ? v1 : 0.f;
      const unsigned bit = v1 > v0 ? 1u : 0u;
      const bool mine = h == 0 && live;
      if (mine) {
        p.bits[static_cast<int64_t>(pos) * p.n_rows + row] = static_cast<uint8_t>(bit);
        if (p.head_act) {
          const f32x2 o = {v0, v1};
          *reinterpret_cast<f32x2 *>(p.head_act + (static_cast<int64_t>(pos) * p.n_rows + row) * 2) = o;
        }
      }
      if (p.errors) {
        const bool wrong = mine && p.labels[static_cast<int64_t>(pos) * p.n_rows + row] != bit;
        const unsigned long long m = __ballot(wrong);
        if (lane == 0 && m) atomicAdd(p.errors + pos, static_cast<unsigned long long>(__popcll(m)));
      }
    }
    __syncthreads();   // every wave is done with these rows: the buffer may be refilled
  }
}

}  // namespace

int64_t bank_stream_lds_bytes(int groups) { return static_cast<int64_t>(2 * groups + 2) * 32 * 16; }

nnv_status bank_stream_prepare(int groups, int device) {
  static int lds_set[kMaxDevices] = {};
  static std::mutex mu;
  if (device < 0 || device >= kMaxDevices) return NNV_ERR_INVALID_ARGUMENT;
  const int lds_bytes = static_cast<int>(bank_stream_lds_bytes(groups));
  if (lds_bytes > 160 * 1024) return NNV_ERR_UNSUPPORTED_SHAPE;
  const std::lock_guard<std::mutex> lock(mu);
  if (lds_set[device] < lds_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(bank_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds_bytes) != hipSuccess)
      return NNV_ERR_HIP;
    lds_set[device] = lds_bytes;
  }
  return NNV_OK;
}

nnv_status bank_stream_launch(const BankStreamParams &p, int num_cus, int device, hipStream_t stream) {
  if (p.n_rows <= 0) return NNV_OK;
  const nnv_status e = bank_stream_prepare(p.groups, device);   // a no-op after prepack
  if (e != NNV_OK) return e;
  int64_t blocks = p.n_blocks < num_cus ? p.n_blocks : num_cus;   // persistent: one workgroup per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bank_stream_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads),
                     static_cast<unsigned>(bank_stream_lds_bytes(p.groups)), stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
