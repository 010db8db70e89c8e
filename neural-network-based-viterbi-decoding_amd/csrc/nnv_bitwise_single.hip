// K4d: ONE bit-wise decoder position -- the reference's own call shape -- as an HBM-streaming kernel with the
// contraction split over the 16 waves of a workgroup (round 4).
//
//   src/fnn_viterbi_bitwise.py:326-399 build_network   2(6+L) -> n_hidden (30) -> 2, ReLU after both layers
//   src/fnn_viterbi_bitwise.py:503-505 argmax eval     one trained network per bit_position, one position per run
//
// One position is 62 kFLOP against 4,144 B per row: HBM-shaped (0.155 ms of HBM time at 8 TB/s for 300 k rows beside
// 0.127 ms of fp32-MFMA time).  As a tiled GEMM with a 32-column tile (K4b, nnv_dense.hip) it runs at 0.263 ms: both
// operands staged through LDS behind two barriers per k-tile, the loads of a tile not under the MFMAs of the previous.
// With a single position there is nothing to share the received rows WITH -- so they need no LDS at all:
//  * wave w of a 16-wave persistent workgroup owns k-slice w of the 1036 inputs (8-9 fragment groups of 8 inputs)
//    for good.  Its slice of the first layer lives in REGISTERS (36 VGPRs, loaded once per launch): no weight
//    traffic in the loop.
//  * per 32-row tile the wave loads ITS slice of the rows straight from HBM into the MFMA B operand: lane (row r,
//    half h) takes 16 bytes per group -- per row 256-288 contiguous bytes, i.e. whole cache lines, each consumed by the
//    two lanes of a row in back-to-back loads.  The loads of tile t+1 are issued slot by slot as tile t's MFMAs free
//    the registers (one tile = ~10 k cycles ahead, 130 KB in flight per CU): no staging, no barrier on the way in.
//  * the 16 partial 32 x 32 accumulators meet in LDS (64 KB, double-buffered): one barrier, every thread sums the 16
//    partials of ONE (neuron, row) element in slice order, + bias, ReLU; one more barrier, one wave (rotating) runs the
//    2-logit head: sequential fmaf chain over the hidden units, bias, ReLU, first maximum -- as K4b / K4c.
// The same kernel serves the f16x3 mode of a single position (F16 = true: steps of 16 inputs, hi / lo planes split in
// registers, three 16-bit MFMAs per step): 0.24 ms against 0.41 ms on the 128-column f16x3 tile.
// Summation order: per slice k ascending, then the slices in ascending order -- deterministic and a function of the row
// alone, but NOT K4b's single chain: hidden values differ from the tiled path's in the last bits (same 1e-5 contract
// against float64; the tests hold both against the oracle).
#include <stdlib.h>

#include <mutex>

#include "nnv_internal.h"

namespace nnv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kPartFloats = kWaves * 1024;         // one buffer of partial accumulators: [wave][q 4][lane 64][4]
constexpr int kHidFloats = 32 * 32;                // [neuron][row]
#ifndef NNV_SINGLE_BURST
#define NNV_SINGLE_BURST 3
#endif
// Slots refilled back to back (consecutive slots are consecutive 32-byte pieces of the same cache lines).  Same box,
// 300 k rows x 1036 inputs, one position: 1: 0.282 ms, 2: 0.264, 3: 0.254, 5: 0.255, 9 (the whole slice at the end of
// the tile): 0.259; K4b 0.263.  Timing experiments on the same box (profiles/r04/bank_single_probe.txt): without the
// MFMAs (loads + reduction only) 0.209-0.217 ms = 5.7-5.9 TB/s -- and the same with 4 rows x 256 contiguous bytes per
// load instruction (0.205), so it is HBM, not the per-lane rows, that bounds the load side; without the loads 0.184 ms.
constexpr int kBurst = NNV_SINGLE_BURST;
constexpr int kLdsFloats = 2 * kPartFloats + 2 * kHidFloats + 64 + 4;   // + head weights [32][2] + head bias

// The tile loop of a wave that owns NG fragment groups.  No branch inside the group loop: the waits on the rows are
// counts (the loads return in order), never "everything outstanding" -- which would include the next tile's rows.
// F16 (the f16x3 mode): a slot is a STEP of 16 inputs = one v_mfma_f32_32x32x16_f16 triple (w_lo x_hi + w_hi x_lo +
// w_hi x_hi, fp32 accumulation, the order of the f16x3 GEMM); lane (row r, half kb) loads 32 bytes of its row per step
// (two 16-byte loads) and splits them into hi / lo f16 planes in registers; the weights are packed as hi / lo fragments.
// `kill_last` then counts the valid 16-byte chunks (1..4) of the row's LAST step instead of being a flag.
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4 &hi, u32x4 &lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 x = {v[2 * i], v[2 * i + 1]};
    const f16x2 hh = __builtin_convertvector(x, f16x2);
    const f32x2 rr = x - __builtin_convertvector(hh, f32x2);
    hi[i] = __builtin_bit_cast(unsigned, hh);
    lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2));
  }
}

template <int NG, bool F16>
__device__ __forceinline__ void single_run(const BankStreamParams &p, float *lds, const int g0, const int kill_last) {
  constexpr int NA = NG > 0 ? NG : 1;
  constexpr int kSlotBytes = F16 ? 64 : 32;          // bytes of a row per slot
  constexpr int kFragBytes = F16 ? 2048 : 1024;      // bytes of weight fragments per slot
  float *const hw_lds = lds + 2 * kPartFloats + 2 * kHidFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const unsigned lane_off = static_cast<unsigned>(lane) * 16u;

  f32x4 w[NA];                 // fp32: A fragments of 4 MFMAs; f16x3: the hi plane (8 halfs) ...
  f32x4 w2[F16 ? NA : 1];      // ... and the lo plane
  {
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.groups * kFragBytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane_off, (g0 + j) * kFragBytes, 0));
      if constexpr (F16) w2[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane_off, (g0 + j) * kFragBytes + 1024, 0));
    }
  }
  // this thread's element of the reduction: register 4q + e of lane l = hidden neuron 8q + 4 (l >> 5) + e of row l & 31
  const int red_n = 8 * (tid >> 8) + 4 * ((tid >> 7) & 1) + (tid & 3);
  const int red_row = (tid >> 2) & 31;
  const float red_bias = p.bias[red_n];

  // rows of tile t through a descriptor of their own: base = the tile's first row (64-bit), offsets stay small; rows
  // past n_rows (tail tile, or the tile after the last) are out of range: they load as 0 and touch no memory
  auto tile_rsrc = [&](int64_t t) -> __amdgpu_buffer_rsrc_t {
    const int64_t rows_left = p.n_rows - t * 32;
    int64_t bytes = rows_left > 0 ? rows_left * p.row_stride * 4 : 0;
    if (bytes > 0x7FFFF000LL) bytes = 0x7FFFF000LL;
    const float *b = p.x + (rows_left > 0 ? t * 32 * p.row_stride : 0);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(b), 0, static_cast<int>(static_cast<unsigned>(bytes)), 0x00020000);
  };
  // labels the same way (no labels: an empty descriptor, the load returns 0 and touches nothing)
  auto label_rsrc = [&](int64_t t) -> __amdgpu_buffer_rsrc_t {
    const int64_t rows_left = p.labels ? p.n_rows - t * 32 : 0;
    const int64_t bytes = rows_left > 0 ? (rows_left < 32 ? rows_left : 32) : 0;
    const uint8_t *b = rows_left > 0 ? p.labels + t * 32 : nullptr;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(b), 0, static_cast<int>(bytes), 0x00020000);
  };
  const unsigned x_off = static_cast<unsigned>((static_cast<int64_t>(r) * p.row_stride + (F16 ? 8 : 4) * h) * 4);   // lane's row + half
  const int s_off = g0 * kSlotBytes;                                                                    // slice start, bytes
  // The group the row ends in (K % 8 == 4) is the last one of one wave: the upper half of it belongs to the NEXT row
  // and must never be multiplied, not even by 0 (a NaN there would cross rows).  Those lanes load it through an offset
  // beyond every descriptor's range (num_records <= 0x7FFFF000, slice offsets < 2^30): the range check returns 0.
  // (f16x3: the last step holds kill_last = 1..4 valid 16-byte chunks; this lane's two are chunks 2h and 2h + 1.)
  const unsigned x_off_last = F16 ? ((kill_last && 2 * h >= kill_last) ? 0x80000000u : x_off)
                                  : ((kill_last && h) ? 0x80000000u : x_off);
  const unsigned x_off2 = x_off + 16u;
  const unsigned x_off2_last = (kill_last && 2 * h + 1 >= kill_last) ? 0x80000000u : x_off2;

  f32x4 b[NA];
  f32x4 b2[F16 ? NA : 1];      // f16x3: the second 16 bytes of the lane's 32
  auto load_slot = [&](const __amdgpu_buffer_rsrc_t rs, int j) {
    b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, j == NG - 1 ? x_off_last : x_off, s_off + j * kSlotBytes, 0));
    if constexpr (F16)
      b2[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, j == NG - 1 ? x_off2_last : x_off2, s_off + j * kSlotBytes, 0));
  };
  {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(blockIdx.x);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      load_slot(rs, j);
      __builtin_amdgcn_sched_barrier(0);   // in slot order, like the refills: the loop's waits are counts relative to that order
    }
  }

  int buf = 0, it = 0;
  for (int64_t tile = blockIdx.x; tile < p.n_blocks; tile += gridDim.x, ++it) {
    const __amdgpu_buffer_rsrc_t next_rs = tile_rsrc(tile + gridDim.x);
    // this tile's labels, requested BEFORE the refills below: older than every load the head will not wait for
    const unsigned label = __builtin_amdgcn_raw_buffer_load_b8(label_rsrc(tile), static_cast<unsigned>(r), 0, 0);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if constexpr (F16) {
        u32x4 xh, xl;
        split8(b[j], b2[j], xh, xl);
        const f16x8 wh = __builtin_bit_cast(f16x8, w[j]), wl = __builtin_bit_cast(f16x8, w2[j]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, __builtin_bit_cast(f16x8, xh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, __builtin_bit_cast(f16x8, xl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, __builtin_bit_cast(f16x8, xh), acc, 0, 0, 0);
      } else {
        const f32x4 c = b[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], c[e], acc, 0, 0, 0);
      }
      // the next tile's groups into the registers just freed (refilled AFTER their consumers: no copies), kBurst
      // slots back to back: consecutive slots are consecutive 32-byte pieces of the same cache lines
      constexpr int burst = F16 ? 1 : kBurst;
      if ((j + 1) % burst == 0 || j == NG - 1) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = j - j % burst; i <= j; ++i) {
          load_slot(next_rs, i);
          __builtin_amdgcn_sched_barrier(0);   // keep the refills HERE and in slot order: the waits are counts relative to that order
        }
      }
    }
    // ---------------- the 16 partial accumulators meet in LDS --------------------------------------------
    float *const part = lds + buf * kPartFloats;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      *reinterpret_cast<f32x4 *>(part + wave * 1024 + q * 256 + lane * 4) = v;
    }
    __syncthreads();
    float *const hid = lds + 2 * kPartFloats + buf * kHidFloats;
    {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < kWaves; ++wv) s += part[wv * 1024 + tid];
      s += red_bias;                                     // matmul, add, relu: src/fnn_viterbi_bitwise.py:392-393
      hid[red_n * 32 + red_row] = s > 0.f ? s : 0.f;
    }
    __syncthreads();
    // ---------------- head: relu(h . Wout + bout), first maximum (:396-397, :504-505) ----------------------
    if (wave == (it & (kWaves - 1))) {
      float s = 0.f;                                     // lane: row r, output h
#pragma unroll
      for (int n = 0; n < 32; ++n) s = __builtin_fmaf(hid[n * 32 + r], hw_lds[2 * n + h], s);
      float v = s + hw_lds[64 + h];
      v = v > 0.f ? v : 0.f;
      const float o = __shfl_xor(v, 32);
      const int64_t row = tile * 32 + r;
      const bool live = row < p.n_rows;
      const unsigned bit = (h == 0 ? o > v : v > o) ? 1u : 0u;   // v1 > v0
      if (live) {
        if (h == 0) p.bits[row] = static_cast<uint8_t>(bit);
        if (p.head_act) p.head_act[row * 2 + h] = v;
      }
      if (p.errors) {
        const bool wrong = live && h == 0 && label != bit;
        const unsigned long long m = __ballot(wrong);
        if (lane == 0 && m) atomicAdd(p.errors, static_cast<unsigned long long>(__popcll(m)));
      }
    }
    // (Tried: 8-wave workgroups with TWO tiles of rows in flight per wave (k-slices of 16-17 groups: 68 + 136 + 16 of 256
    // VGPRs, 6 spilled): 0.259-0.262 vs 0.258-0.260 ms on the same box -- the depth of the window is not what the
    // overlap of the two sides waits for; not kept.  The same 8-wave kernel with the next tile's whole slice requested
    // at the START of a pass, ahead of every MFMA (tools/ubench_hbm_mfma.hip: a SIMD busy with fp32 MFMA chains hardly
    // issues anything for its other waves, so a refill behind a wave's MFMAs also waits for the other waves'):
    // 0.279-0.281 vs 0.258-0.261 ms.)
    // (Tried: ONE barrier per tile -- tile t's partials reduced at the start of pass t + 1, under the other waves' MFMAs,
    // its head behind pass t + 1's barrier: 0.270-0.276 vs 0.257-0.259 ms on the same box, profiles/r04/
    // bank_single_probe.txt; the serial LDS round trip at the head of every wave's pass costs more than the barrier.)
    buf ^= 1;   // the next tile writes the other buffers: its partials cannot reach this tile's reduction, which ended
                // behind the second barrier; this tile's hidden values are read by the head wave before IT passes the
                // next tile's first barrier, and rewritten only two tiles later
  }
}

// G = ceil(slots / 16): a wave owns G or G - 1 slots (fp32: fragment groups of 8 inputs; f16x3: steps of 16 inputs)
template <int G, bool F16 = false>
__global__ __launch_bounds__(kThreads) void bank_single_kernel(const BankStreamParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // k-slice of this wave: groups [g0, g0 + ng)
  const int base = p.groups / kWaves, rem = p.groups % kWaves;
  const int ng = base + (wave < rem ? 1 : 0);
  const int g0 = wave * base + (wave < rem ? wave : rem);
  // the wave whose LAST group is the one the row ends in (K % 8 == 4): that group's upper half belongs to the NEXT
  // row -- it is never multiplied, not even by 0
  // (f16x3: the number of valid 16-byte chunks, 1..3, of a last step the row does not fill; 0 = nothing to kill)
  const bool owns_last = ng > 0 && g0 + ng == p.groups;
  const int kill_last = F16 ? ((owns_last && p.K % 16 != 0) ? (p.K % 16) / 4 : 0) : ((owns_last && p.K % 8 != 0) ? 1 : 0);
  float *const hw_lds = lds + 2 * kPartFloats + 2 * kHidFloats;
  if (tid < 64) hw_lds[tid] = p.head_w[tid];
  if (tid >= 64 && tid < 66) hw_lds[tid] = p.head_b[tid - 64];
  // Two copies of the tile loop, one per slice length; both run the same barriers the same number of times (the
  // trip count depends on the workgroup only).
  if (ng == G) single_run<G, F16>(p, lds, g0, kill_last);
  else single_run<G - 1, F16>(p, lds, g0, kill_last);
}

using SingleFn = void (*)(const BankStreamParams);
SingleFn single16_fn(int per_wave) {   // f16x3: steps of 16 inputs per wave
  switch (per_wave) {
    case 1: return bank_single_kernel<1, true>;
    case 2: return bank_single_kernel<2, true>;
    case 3: return bank_single_kernel<3, true>;
    case 4: return bank_single_kernel<4, true>;
    default: return bank_single_kernel<5, true>;
  }
}
SingleFn single_fn(int per_wave) {
  switch (per_wave) {
    case 1: return bank_single_kernel<1>;
    case 2: return bank_single_kernel<2>;
    case 3: return bank_single_kernel<3>;
    case 4: return bank_single_kernel<4>;
    case 5: return bank_single_kernel<5>;
    case 6: return bank_single_kernel<6>;
    case 7: return bank_single_kernel<7>;
    case 8: return bank_single_kernel<8>;
    default: return bank_single_kernel<9>;
  }
}

}  // namespace

// Eligible: fp32, n_hidden <= 32, at most 9 fragment groups per wave (n_input <= 1152), 16-byte aligned rows (checked
// per call), 32 rows inside a 30-bit byte offset.
bool bank_single_eligible(int groups, int64_t row_stride) {
  return (groups + kWaves - 1) / kWaves <= 9 && row_stride * 4 * 32 < (1LL << 30);
}

nnv_status bank_single_prepare(int groups, int device) {
  static int set[kMaxDevices][10] = {};
  static std::mutex mu;
  if (device < 0 || device >= kMaxDevices) return NNV_ERR_INVALID_ARGUMENT;
  const int per_wave = (groups + kWaves - 1) / kWaves;
  if (per_wave > 9) return NNV_ERR_UNSUPPORTED_SHAPE;
  const int idx = per_wave;
  const std::lock_guard<std::mutex> lock(mu);
  if (!set[device][idx]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(single_fn(per_wave)), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLdsFloats * 4) != hipSuccess)
      return NNV_ERR_HIP;
    set[device][idx] = 1;
  }
  return NNV_OK;
}

// f16x3 variant: p.groups = steps of 16 inputs, p.w = [step][plane hi | lo][lane 64][8 halfs].  At most 5 steps per wave.
bool bank_single16_eligible(int steps, int64_t row_stride) {
  return (steps + kWaves - 1) / kWaves <= 5 && row_stride * 4 * 32 < (1LL << 30);
}

nnv_status bank_single16_prepare(int steps, int device) {
  static int set[kMaxDevices][6] = {};
  static std::mutex mu;
  if (device < 0 || device >= kMaxDevices) return NNV_ERR_INVALID_ARGUMENT;
  const int per_wave = (steps + kWaves - 1) / kWaves;
  if (per_wave > 5) return NNV_ERR_UNSUPPORTED_SHAPE;
  const std::lock_guard<std::mutex> lock(mu);
  if (!set[device][per_wave]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(single16_fn(per_wave)), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kLdsFloats * 4) != hipSuccess)
      return NNV_ERR_HIP;
    set[device][per_wave] = 1;
  }
  return NNV_OK;
}

nnv_status bank_single16_launch(const BankStreamParams &p, int num_cus, int device, hipStream_t stream) {
  if (p.n_rows <= 0) return NNV_OK;
  const nnv_status e = bank_single16_prepare(p.groups, device);   // a no-op after prepack
  if (e != NNV_OK) return e;
  int64_t blocks = p.n_blocks < num_cus ? p.n_blocks : num_cus;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(single16_fn((p.groups + kWaves - 1) / kWaves), dim3(static_cast<unsigned>(blocks)), dim3(kThreads),
                     kLdsFloats * 4, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

// p describes ONE position: every per-position pointer (w, bias, head_w, head_b, bits, head_act, labels, errors)
// already points at that position's part.
nnv_status bank_single_launch(const BankStreamParams &p, int num_cus, int device, hipStream_t stream) {
  if (p.n_rows <= 0) return NNV_OK;
  const nnv_status e = bank_single_prepare(p.groups, device);   // a no-op after prepack
  if (e != NNV_OK) return e;
  int64_t blocks = p.n_blocks < num_cus ? p.n_blocks : num_cus;   // persistent: one workgroup per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(single_fn((p.groups + kWaves - 1) / kWaves), dim3(static_cast<unsigned>(blocks)), dim3(kThreads),
                     kLdsFloats * 4, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
