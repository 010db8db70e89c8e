// K6: the WHOLE wide network in one persistent launch -- activations never leave the CU.
//
// First user: the reference's sliding-window decoder
//   src/fnn_viterbi_seq_predict.py:293-336  window i = coded[2i : 2i+window)
//   src/fnn_viterbi_seq_predict.py:343-414  200 -> 500 -> 500 -> 500 -> 500 -> 2, tanh everywhere
//   src/fnn_viterbi_seq_predict.py:497,592-611  argmax over the 2 outputs = the decoded bit
// The layer-at-a-time path (nnv_dense.hip, K4) runs one GEMM launch per layer with the [rows, 512]
// activations going out to HBM and back between layers; it sits at 0.73 of the fp32-MFMA peak: operand
// staging (global -> VGPR -> LDS, two barriers per k-tile) costs 9 %, the store epilogue 8 %, padding
// 500 -> 512 / 200 -> 224 another 6 %.  This kernel removes all three for networks whose hidden widths are
// <= 512 and whose class layer is narrow (<= 32 classes):
//  * A persistent workgroup of 16 waves (one per CU) owns 64 rows at a time.  Their activations live in LDS
//    for the whole chain, in the MFMA B-operand order [k-quad][row][4]: 128 quads x 64 rows x 16 B = 128 KB
//    for a 512-wide layer.
//  * Same transposed formulation as K1: H_l^T[neurons x rows] = W_l^T . H_{l-1}^T on v_mfma_f32_32x32x2_f32
//    (exact fp32 fmaf chain, k ascending).  Wave w owns neuron tile w (32 neurons) of the layer for BOTH
//    32-row tiles: per 8 inputs one 16-byte weight fragment per lane straight from global memory (packed
//    on the host in fragment order: one wave-load = one contiguous 1 KB), two conflict-free ds_read_b128 of
//    activations, eight MFMAs.  No operand staging, no barrier inside a layer, no vector-ALU address
//    arithmetic in the loop (scalar base + immediate offsets).
//  * The weights (3.5 MB for the reference's network) are streamed from the XCD's L2 by every workgroup:
//    8 bytes per cycle and CU, each fragment read by exactly ONE wave of the workgroup (no redundancy inside
//    a CU, unlike K1's round-2 per-wave streaming), prefetched kPrefetch fragment groups (~ 4 x 2k cycles)
//    ahead.  All 256 workgroups walk the layers together, so HBM sees each weight once per XCD.
//  * Layer hand-over: barrier (everyone has read the inputs), bias + activation on the accumulators in
//    the reference's op order matmul -> add -> tanh, one ds_write_b128 per 4 neurons (the accumulator's
//    register order IS the next layer's k-quad order), barrier.  K is padded to 8 only (500 -> 504,
//    200 exact), N to 32 (500 -> 512).
//  * The narrow class layer (<= 32 classes) runs on the vector ALU: lane = row, wave = a 1/16 slice of
//    the inputs (weights by scalar loads), partial sums through LDS, first-maximum argmax.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "nnv_internal.h"

namespace nnv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

namespace {

constexpr int kRows = kFusedRows;        // rows per workgroup step (two 32-row MFMA tiles)
// WAVES (template): 16 waves = one workgroup per CU, one neuron tile each up to 512-wide layers; or -- networks of
// at most 256-wide (128-wide) layers whose buffers fit 80 KB (40 KB) -- 8 (4) waves and TWO (FOUR) workgroups per
// CU, whose hand-overs (barriers, first loads, the argmax pass) then run under each other's MFMAs: a 64-row step of
// a narrow network is mostly such latencies (14-200-96-48-128: 24 k MFMA cycles per SIMD against ~20 k of them;
// measured 2.32 -> 1.95 ms per 2 M rows, 16-256-128-64-256 2.94 -> 2.53 ms).
constexpr int kPrefetch = 4;  // weight fragment groups in flight per wave (measured 2 / 4 / 8: 4 is the fastest)

__device__ __forceinline__ float fused_tanh_f32(float z) {  // == dense_tanh_f32 (nnv_dense.hip)
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  return __builtin_fmaf(-2.0f, r, 1.0f);
}

template <int ACT>
__device__ __forceinline__ float fused_act(float z) {
  if constexpr (ACT == 1) return fused_tanh_f32(z);
  else return z > 0.f ? z : 0.f;
}

template <int ACT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void wide_fused_kernel(const WideFusedParams p) {
  constexpr int kWaves = WAVES;
  constexpr int kThreads = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const abuf = lds;                                           // [quad][64 rows][4 floats]
  float *const scratch = lds + static_cast<int64_t>(p.act_quads) * (kRows * 4);  // [2 classes][WAVES][64 rows]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // a scalar: branches / weight addresses are uniform
  const int r = lane & 31, h = lane >> 5;
  const unsigned lane_off = static_cast<unsigned>(lane) * 16u;       // byte offset of the lane inside a 1 KB fragment group

  // Weight fragments by buffer loads: the descriptor and the group offset are scalars (soffset), the lane's 16
  // bytes a constant voffset -- no vector-ALU address arithmetic in the loop, and a prefetch past a layer's end
  // needs no clamp (the range check returns 0; inside the layer it reads the next tile's first groups, unused).
  auto layer_rsrc = [&](int l) -> __amdgpu_buffer_rsrc_t {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w[l]), 0, p.tiles[l] * p.groups[l] * 1024, 0x00020000);
  };
  auto load_w = [&](__amdgpu_buffer_rsrc_t rsrc, int soff) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, soff, 0));
  };
  f32x4 a[kPrefetch];   // fragment groups in flight: wave `wave`'s neuron tile, groups g .. g + kPrefetch - 1
  auto prefetch_layer = [&](int l) {
    if (wave < p.tiles[l]) {
      const __amdgpu_buffer_rsrc_t rsrc = layer_rsrc(l);
      const int tile_off = wave * p.groups[l] * 1024;
#pragma unroll
      for (int i = 0; i < kPrefetch; ++i) a[i] = load_w(rsrc, tile_off + i * 1024);
    }
  };
  prefetch_layer(0);

  // Work is dealt in UNITS of 32 rows (one MFMA row tile), a contiguous range per workgroup whose sizes differ
  // by at most one unit; a workgroup takes its units two at a time (a 64-row step: every weight fragment feeds
  // two row tiles) and a trailing odd unit as a HALF step (the second tile's MFMAs and stores skipped).  Whole
  // 64-row blocks dealt round-robin quantise the run to ceil(blocks / 256) steps: 1 M windows = 61.04 steps per
  // workgroup cost 62; in units it is 61.5.
  const int64_t u_begin = p.n_units * blockIdx.x / gridDim.x;
  const int64_t u_end = p.n_units * (static_cast<int64_t>(blockIdx.x) + 1) / gridDim.x;
  constexpr int kUnitStep = 2;
  for (int64_t u = u_begin; u < u_end; u += kUnitStep) {
    const int64_t row0 = u * 32;
    const bool two = kUnitStep == 2 && u + 1 < u_end;   // workgroup-uniform
    const int64_t row_end = (two ? row0 + 64 : row0 + 32) < p.n_rows ? (two ? row0 + 64 : row0 + 32) : p.n_rows;
    // ---------------- the rows' inputs -> LDS in B-operand order (layer 0's "activations") -------------
    {
      const int nq = 2 * p.groups[0];
      for (int it = tid; it < nq * kRows; it += kThreads) {
        const int q = it >> 6, rr = it & 63;       // consecutive threads: consecutive rows, one contiguous LDS span
        int64_t row = row0 + rr;
        row = row < p.n_rows ? row : p.n_rows - 1;  // tail rows: valid memory, results discarded
        const int k = 4 * q;
        const float *src = p.x + row * p.row_stride + k;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k + 3 < p.K0) {
          if (p.x16) {
            v = *reinterpret_cast<const f32x4 *>(src);
          } else {                                  // windows of a stream: 8-byte aligned
            const f32x2 t0 = *reinterpret_cast<const f32x2 *>(src);
            const f32x2 t1 = *reinterpret_cast<const f32x2 *>(src + 2);
            v[0] = t0[0], v[1] = t0[1], v[2] = t1[0], v[3] = t1[1];
          }
        } else {                                    // the row ends inside this quad: guarded, zero filled
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < p.K0) v[e] = src[e];
        }
        *reinterpret_cast<f32x4 *>(abuf + static_cast<int64_t>(it) * 4) = v;
      }
    }
    __syncthreads();

    // ---------------- hidden layers on the matrix cores ------------------------------------------------
    for (int l = 0; l < p.n_hidden; ++l) {
      const int groups = p.groups[l];
      const bool active = wave < p.tiles[l];         // wave-uniform
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
      f32x4 bias4[4];
      if (active) {
        // a[] already holds this layer's first kPrefetch fragment groups (requested before the previous
        // layer's hand-over, or in the prologue)
        const __amdgpu_buffer_rsrc_t rsrc = layer_rsrc(l);
        const int tile_off = wave * groups * 1024;
        // this lane's biases (added after the chain, the reference's op order): 4 x 4 consecutive neurons
        const float *bl = p.bias[l] + 32 * wave + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4 *>(bl + 8 * q);
        // activations: quad 2g + h, rows r (tile 0) and 32 + r (tile 1); read one group ahead of its MFMAs.
        // (The read ahead of the LAST group lands at most 2 KB behind the buffer's used part: valid LDS, unused.)
        const float *bp = abuf + (h * kRows + r) * 4;
        f32x4 b0 = *reinterpret_cast<const f32x4 *>(bp);
        f32x4 b1 = *reinterpret_cast<const f32x4 *>(bp + 32 * 4);
        auto run = [&](auto two_tag) {
          constexpr bool TWO = decltype(two_tag)::value;
          auto step = [&](const f32x4 av, int g) {
            const f32x4 c0 = b0, c1 = b1;
            b0 = *reinterpret_cast<const f32x4 *>(bp + (g + 1) * (2 * kRows * 4));
            if constexpr (TWO) b1 = *reinterpret_cast<const f32x4 *>(bp + (g + 1) * (2 * kRows * 4) + 32 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c0[e], acc0, 0, 0, 0);
              if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], c1[e], acc1, 0, 0, 0);
            }
          };
          int g = 0;
          for (; g + kPrefetch <= groups; g += kPrefetch) {   // no branch inside: loads in order, waits by count
#pragma unroll
            for (int i = 0; i < kPrefetch; ++i) {
              step(a[i], g + i);
              a[i] = load_w(rsrc, tile_off + (g + i + kPrefetch) * 1024);   // AFTER its consumers: the same four registers are refilled (no copies)
            }
          }
#pragma unroll
          for (int i = 0; i < kPrefetch - 1; ++i)              // the last, partial trip (uniform branches)
            if (g + i < groups) step(a[i], g + i);
        };
        if (two) run(std::true_type{});
        else run(std::false_type{});                           // half step: the first row tile only
      }
      // The NEXT layer's first fragment groups (the next 64-row block's first layer after the last one) are
      // requested now, so that their L2 round trip runs under this layer's hand-over (barrier, bias + tanh,
      // LDS stores, barrier) instead of in front of the next layer's first MFMA -- where all 16 waves of all
      // 256 workgroups would wait for it at the same moment.
      prefetch_layer(l + 1 < p.n_hidden ? l + 1 : 0);
      __syncthreads();   // every wave has read this layer's inputs: the buffer may be overwritten
      if (active) {
        // accumulator register 4q + e <-> neuron 32 wave + 8q + 4h + e: four consecutive registers are one
        // k-quad (8 wave + 2q + h) of the next layer, i.e. one 16-byte LDS store per row and quad
        // (Tried, round 4: dealing a narrow layer's 2 x tiles (tile, row tile) pairs one per wave so that fewer waves
        // idle -- no gain on 14-200-96-48-128 / 16-256-128-64-256 (2.31 vs 2.27 ms, 2.97 vs 2.88 ms per 2 M rows):
        // such networks are bound by the per-step latencies -- barriers, first loads --, not by MFMA issue; reverted.)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v0, v1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = fused_act<ACT>(acc0[4 * q + e] + bias4[q][e]);
            v1[e] = two ? fused_act<ACT>(acc1[4 * q + e] + bias4[q][e]) : 0.f;
          }
          float *dst = abuf + ((8 * wave + 2 * q + h) * kRows + r) * 4;
          *reinterpret_cast<f32x4 *>(dst) = v0;
          if (two) *reinterpret_cast<f32x4 *>(dst + 32 * 4) = v1;
        }
      }
      __syncthreads();
    }

    // ---------------- wide class layer (33..512 classes): it was the last MFMA layer -------------------
    // Its activated outputs sit in the buffer like any hidden layer's; first maximum over them: lane = row,
    // wave = a 1/16 slice of the class quads in ascending order, strict >, then the 16 partial winners in
    // ascending slice order (so the lowest index among equal maxima wins).  Padding classes (act(0) = 0 would
    // beat negative outputs) are masked by their index.
    if (p.wide_head) {
      const int qs = (p.head_quads + kWaves - 1) / kWaves;
      const int q_lo = wave * qs;
      const int q_hi = q_lo + qs < p.head_quads ? q_lo + qs : p.head_quads;
      const int64_t row = row0 + lane;
      const bool row_ok = row < row_end;
      float best = -__builtin_inff();
      int bidx = 0;
      for (int q = q_lo; q < q_hi; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(abuf + (q * kRows + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * q + e;
          if (c < p.n_classes) {
            if (p.act && row_ok) p.act[row * p.n_classes + c] = v[e];   // parity / debug output only
            if (v[e] > best) {
              best = v[e];
              bidx = c;
            }
          }
        }
      }
      int *const scratch_i = reinterpret_cast<int *>(scratch) + kWaves * kRows;
      scratch[wave * kRows + lane] = best;
      scratch_i[wave * kRows + lane] = bidx;
      __syncthreads();
      if (tid < kRows) {
        best = -__builtin_inff();
        bidx = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          const float bv = scratch[w * kRows + tid];
          const int bi = scratch_i[w * kRows + tid];
          if (bv > best) {
            best = bv;
            bidx = bi;
          }
        }
        if (row_ok) {                                           // tid < 64: lane == tid
          if (p.cls32) p.cls32[row] = bidx;
          if (p.cls64) p.cls64[row] = bidx;
        }
      }
      __syncthreads();   // scratch and the buffer are free for the next block
    } else {
    // ---------------- narrow class layer + first-maximum argmax (vector ALU) ---------------------------
      const int qs = (p.head_quads + kWaves - 1) / kWaves;   // k-quads per wave slice
      const int q_lo = wave * qs;
      const int q_hi = q_lo + qs < p.head_quads ? q_lo + qs : p.head_quads;
      float best = -__builtin_inff();
      int bidx = 0;
      const int64_t row = row0 + tid;                          // meaningful for tid < 64 only
      for (int c0 = 0; c0 < p.n_classes; c0 += 2) {
        const bool two = c0 + 1 < p.n_classes;
        const float *w0 = p.head_w + static_cast<int64_t>(c0) * p.head_quads * 4;
        const float *w1 = two ? w0 + p.head_quads * 4 : w0;
        float s0 = 0.f, s1 = 0.f;
        for (int q = q_lo; q < q_hi; ++q) {                    // lane = row; weights wave-uniform (scalar loads)
          const f32x4 v = *reinterpret_cast<const f32x4 *>(abuf + (q * kRows + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = __builtin_fmaf(v[e], w0[4 * q + e], s0);
            s1 = __builtin_fmaf(v[e], w1[4 * q + e], s1);
          }
        }
        scratch[(0 * kWaves + wave) * kRows + lane] = s0;
        scratch[(1 * kWaves + wave) * kRows + lane] = s1;
        __syncthreads();
        if (tid < kRows) {
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            if (cc == 1 && !two) break;
            float z = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) z += scratch[(cc * kWaves + w) * kRows + tid];
            const float v = fused_act<ACT>(z + p.head_b[c0 + cc]);   // matmul, add, activation
            if (p.act && row < row_end) p.act[row * p.n_classes + c0 + cc] = v;
            if (v > best) {
              best = v;
              bidx = c0 + cc;
            }
          }
        }
        __syncthreads();   // scratch is reused by the next class pair; the last one also fences abuf for the next block
      }
      if (tid < kRows && row < row_end) {
        if (p.cls32) p.cls32[row] = bidx;
        if (p.cls64) p.cls64[row] = bidx;
      }
    }
  }
}

}  // namespace

int64_t wide_fused_lds_bytes(const WideFusedParams &p) {
  return static_cast<int64_t>(p.act_quads) * kRows * 16 + 2LL * p.waves * kRows * 4;
}

// W waves x (16 / W) workgroups per CU, W the smallest of 4, 8, 16 with: every layer has at most W neuron tiles and
// one workgroup's LDS is at most 160 KB / (16 / W).  (A/B builds only: NNV_FUSED_WAVES=16 or 8 sets a floor.)
int wide_fused_pick_waves(const WideFusedParams &p) {
  static const int floor_w = [] {
    const char *e = ab_env("NNV_FUSED_WAVES");
    const int v = e ? atoi(e) : 0;
    return (v == 4 || v == 8 || v == 16) ? v : 4;
  }();
  int max_tiles = 0;
  for (int l = 0; l < p.n_hidden; ++l) max_tiles = p.tiles[l] > max_tiles ? p.tiles[l] : max_tiles;
  for (const int w : {4, 8}) {
    if (w < floor_w || max_tiles > w) continue;
    const int64_t lds = static_cast<int64_t>(p.act_quads) * kRows * 16 + 2LL * w * kRows * 4;
    if (lds <= (160 * 1024) / (16 / w)) return w;
  }
  return 16;
}

namespace {
using FusedFn = void (*)(const WideFusedParams);
FusedFn fused_fn(int act, int waves) {
  if (waves == 4) return act == 1 ? wide_fused_kernel<1, 4> : wide_fused_kernel<2, 4>;
  if (waves == 8) return act == 1 ? wide_fused_kernel<1, 8> : wide_fused_kernel<2, 8>;
  return act == 1 ? wide_fused_kernel<1, 16> : wide_fused_kernel<2, 16>;
}
// dynamic LDS above the default cap: opt in once per (device, instantiation)
nnv_status fused_lds_optin(int act, int waves, int device, int lds_bytes) {
  static int lds_set[kMaxDevices][6] = {};
  static std::mutex mu;  // prepack / launches may come from several host threads
  if (device < 0 || device >= kMaxDevices || (act != 1 && act != 2) || (waves != 4 && waves != 8 && waves != 16))
    return NNV_ERR_INVALID_ARGUMENT;
  const int idx = (act - 1) * 3 + (waves == 4 ? 0 : (waves == 8 ? 1 : 2));
  const std::lock_guard<std::mutex> lock(mu);
  if (lds_set[device][idx] < lds_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(fused_fn(act, waves)),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
      return NNV_ERR_HIP;
    lds_set[device][idx] = lds_bytes;
  }
  return NNV_OK;
}
}  // namespace

// Called from nnv_wide_prepack (on the handle's device): the kernel attribute a launch of this network needs is
// set here, so that the launches themselves are pure kernel launches -- capturable into a HIP graph from the
// first call, like K1's (decode_prepare).
nnv_status wide_fused_prepare(const WideFusedParams &p, int act, int device) {
  const int64_t lds_bytes = wide_fused_lds_bytes(p);
  if (lds_bytes > 160 * 1024) return NNV_ERR_UNSUPPORTED_SHAPE;
  return fused_lds_optin(act, p.waves, device, static_cast<int>(lds_bytes));
}

nnv_status wide_fused_launch(const WideFusedParams &p, int act, int num_cus, int device, hipStream_t stream) {
  if (p.n_rows <= 0) return NNV_OK;
  if (act != 1 && act != 2) return NNV_ERR_INVALID_ARGUMENT;
  const int lds_bytes = static_cast<int>(wide_fused_lds_bytes(p));
  if (lds_bytes > 160 * 1024) return NNV_ERR_UNSUPPORTED_SHAPE;
  {
    const nnv_status e = fused_lds_optin(act, p.waves, device, lds_bytes);   // a no-op after prepack
    if (e != NNV_OK) return e;
  }
  // persistent: one 16-wave workgroup per CU, two 8-wave or four 4-wave ones.  Short inputs: one 32-row unit (a half
  // step) per workgroup while there are fewer units than resident workgroups -- a launch is then as long as ONE step
  // of one workgroup, and the half step is the shorter one (200 -> 500 x 4 -> 2, <= 8 k rows: 200 -> 1xx us; the
  // layer-at-a-time path took 138-154 us there and was the faster one before: profiles/r04/wide_rows_sweep.txt)
  const int64_t resident = static_cast<int64_t>(num_cus) * (16 / p.waves);
  int64_t blocks = p.n_units < resident ? p.n_units : resident;
  if (blocks < 1) blocks = 1;
  const dim3 grid(static_cast<unsigned>(blocks)), block(64 * p.waves);
  hipLaunchKernelGGL(fused_fn(act, p.waves), grid, block, lds_bytes, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
