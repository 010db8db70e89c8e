// K7: the reference's TRAINING STEP for the feed-forward decoder as two kernels.
//
//   cost = mean softmax-CE(logits = TANH'd class layer, y) + beta (l2(h1) + l2(h2) + l2(h3))   src/fnn_viterbi.py:549-559
//   tf.train.AdamOptimizer(learning_rate).minimize(cost)                                          src/fnn_viterbi.py:561-563
//   optional tf.nn.dropout after every hidden tanh                                 src/fnn_viterbi_mixed_snr.py:596-627
//
// for the reference's architecture n_in -> 128 -> 64 -> 32 -> n_classes (src/fnn_viterbi.py:35-37) with n_in <= 32 and
// n_classes <= 256 (block_length <= 8: the reference default 7 and BASELINE's 8).  A PyTorch-autograd step of this
// network is ~45 tiny kernels, launch-bound at ~0.29 ms even inside a HIP graph; here a step is two launches, 22 us:
//
//   train_fwd_bwd_kernel  one workgroup of four waves per 32 batch rows, all of them on the dependent chain: forward
//       through registers in K1's transposed formulation (rows on the lanes, the accumulator of layer l IS the B operand
//       of layer l + 1), loss and dZ of the class layer in place, back-propagation dH_l = W_{l+1} dZ_{l+1} as the same
//       kind of MFMA chain on a second, transposed packing of the weights -- every layer's 32-wide tiles dealt over the
//       waves and handed over through LDS in the register layout (see the kernel).  The weight gradients
//       dW_l = H_{l-1}^T dZ_l -- a contraction over the ROWS -- come from second LDS copies [neuron][row] of the
//       activations and of every layer's dZ (stride 36 floats: conflict-free 16-byte operand reads).  Five barriers, no
//       atomics: every workgroup writes its partial gradients to its own slab.
//   train_adam_kernel     one thread per parameter: sums the slabs, adds beta W for h1..h3, TF-1 Adam (lr_t from the
//       host in double, epsilon OUTSIDE the bias correction), and writes the weight in its three forms (canonical
//       [in][out], forward fragment order, backward fragment order).  The cost's shares are added to running sums that
//       nnv_train_export adds up; only a caller that wants per-step costs pays for their assembly by the last block.
//
// exact fp32 (v_mfma_f32_32x32x2_f32); summation order differs from a row-major matmul, the results are held to the
// float64 restatement (oracle/train_oracle.py) at 1e-6 in tests/test_trainer.py.
#include <math.h>

#include <new>
#include <vector>

#include "nnv_internal.h"

namespace nnv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kH1 = 128, kH2 = 64, kH3 = 32;   // the reference's hidden widths (src/fnn_viterbi.py:35-37)
constexpr int kT0 = 1, kT1 = 4, kT2 = 2, kT3 = 1;   // 32-wide tiles of the (padded) input and the hidden layers
constexpr int kS = 36;                          // LDS row stride of a [neuron][32 rows] copy, in floats

struct TrainParams {
  const float *x;            // [B][n_in] this step's rows
  const int32_t *labels;     // [B] class index (the position of the 1 in the reference's one-hot row)
  int32_t n_in, n_classes;
  const float *wf[4];        // forward fragments  [mt][kt][q][lane][4]: W[32kt + 8q + 4(lane>>5) + e][32mt + (lane&31)]
  const float *wb[4];        // backward fragments [kt][mt][q][lane][4]: W[32kt + (lane&31)][32mt + 8q + 4(lane>>5) + e]
  const float *bias[4];      // padded to tiles
  float *gpart;              // [n_wg][grad_floats] partial gradients of this step
  int64_t grad_floats;
  float *cost_part;          // [n_wg] this step's cross-entropy shares
  double *cost_acc;          // nullable [n_wg]: their running sums over the steps (read and reset by nnv_train_export)
  const float *mask[3];      // nullable: [B][128], [B][64], [B][32] dropout masks (0/1) of this step
  float inv_keep, inv_batch;
  int32_t batch;             // rows of the step: the last workgroup's tile may be ragged
};

__device__ __forceinline__ int neuron_of(int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); }

#if defined(NNV_AB_BUILD) && defined(NNV_DIAG_TRAIN)   // diagnostic build only (tools/build_variant.sh): 100 MHz stamps of block 0
__device__ unsigned long long g_train_diag[64];      // [16 wave + k]: k = 0 start, 1..5 behind barrier k, 6 end, 8..12 in front of barrier k - 7
#define NNV_TRAIN_STAMP(slot)                                                                            \
  do {                                                                                                   \
    if (blockIdx.x == 0 && lane == 0) g_train_diag[slot] = __builtin_amdgcn_s_memrealtime();             \
  } while (0)
#else
#define NNV_TRAIN_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ float fast_tanh(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// A single wave has nobody to hide its load latency behind (a fragment group comes from L2 in ~1 us): ALL fragment
// groups of a layer (N = tiles x k-tiles x 4, at most 32 = 128 registers) are requested together, then the layer's MFMA
// chains run on them -- one exposed round trip per layer instead of one per tile.
template <int N>
__device__ __forceinline__ void load_frags(const float *frag, int lane, f32x4 *a) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = *reinterpret_cast<const f32x4 *>(frag + (i * 64 + lane) * 4);
  // keep the requests HERE: left alone, the scheduler sinks every load to just above its four MFMAs (to save
  // registers), which is exactly the exposed round trip per group this function exists to avoid
  __builtin_amdgcn_sched_barrier(0);
}
// acc += W-tile (its KT * 4 fragment groups at a[0 ..)) * in[0 .. KT)
template <int KT>
__device__ __forceinline__ f32x16 chain(const f32x16 (&in)[KT], const f32x4 *a, f32x16 acc) {
#pragma unroll
  for (int i = 0; i < KT * 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], in[i / 4][4 * (i % 4) + e], acc, 0, 0, 0);
  return acc;
}

// one k-tile: acc += W-tile (4 fragment groups) * in
__device__ __forceinline__ f32x16 chain_one(const f32x16 &in, const f32x4 *a, f32x16 acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], in[4 * i + e], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ f32x16 bias_tile(const float *b, int mt, int half) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(b + 32 * mt + 8 * q + 4 * half);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * q + e] = bv[e];
  }
  return acc;
}

// tile (neurons 32 t .. 32 t + 31 of a layer, this wave's 32 rows) -> LDS copy [neuron][row]
__device__ __forceinline__ void put_tile(float *s, int t, const f32x16 &v, int row, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) s[(32 * t + neuron_of(r, half)) * kS + row] = v[r];
}
// dW tile (it, jt) = sum over the 32 rows of h[32 it + i][row] dz[32 jt + j][row]; D has j on the lane, i in the registers
__device__ __forceinline__ void grad_tile(const float *hs, const float *zs, int it, int jt, int lane, float *gw, int n_out_pad) {
  const int li = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(hs + (32 * it + li) * kS + 8 * g + 4 * half);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(zs + (32 * jt + li) * kS + 8 * g + 4 * half);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) gw[static_cast<int64_t>(32 * it + neuron_of(r, half)) * n_out_pad + 32 * jt + li] = acc[r];
}

// db[32 t + n] = sum over the rows of dz: lane n < 32 adds its neuron's 32 values (tiles t = first, first + step, ...)
__device__ __forceinline__ void bias_grad(const float *zs, int tiles, int lane, float *gb, int first = 0, int step = 1) {
  if (lane < 32) {
    for (int t = first; t < tiles; t += step) {
      const float *p = zs + (32 * t + lane) * kS;
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * g);
        s += (v[0] + v[1]) + (v[2] + v[3]);
      }
      gb[32 * t + lane] = s;
    }
  }
}

// hidden activation: tanh, optional dropout (x * mask / keep_prob, src/fnn_viterbi_mixed_snr.py:613-624).  DROP is a
// template parameter: a run-time test per element put 221 branches and their address arithmetic into the no-dropout
// kernel, and with fp32 MFMAs every vector instruction is paid for.
template <bool DROP>
__device__ __forceinline__ void activate(f32x16 &acc, const float *mask_row, int mt, int half, float inv_keep) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float h = fast_tanh(acc[r]);
    if constexpr (DROP) h *= mask_row[32 * mt + neuron_of(r, half)] * inv_keep;
    acc[r] = h;
  }
}

// dz = dh * (dropout scale) * (1 - a^2), a = the tanh BEFORE dropout (= a_dropped * keep where the mask is 1; where it is 0
// the gradient is 0 whatever a was)
template <bool DROP>
__device__ __forceinline__ void through_tanh(f32x16 &dh, const f32x16 &hd, const float *mask_row, int mt, int half,
                                             float inv_keep, float keep) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float d = dh[r], a = hd[r];
    if constexpr (DROP) {
      d *= mask_row[32 * mt + neuron_of(r, half)] * inv_keep;
      a *= keep;
    }
    dh[r] = d * (1.0f - a * a);
  }
}

// Four waves per 32 rows (one per SIMD of a CU), ALL of them on the dependent chain: every layer's 32-wide output tiles
// are dealt over the waves, which hand their tiles to each other through LDS in the REGISTER layout (xch_put / xch_get:
// the lane that wrote a value is the lane that reads it back, 16-byte accesses, conflict-free) -- in the transposed
// formulation a layer's accumulator IS the next layer's B operand, so no shuffle is needed on either side.
//   P0  layer 1: tile w                                                   | barrier 1
//   P1  layer 2: tile w (waves 0, 1)                                      | barrier 2
//   P2  layer 3 (one tile: every wave computes it, cheaper than a hand-over); class tiles w, w + 4: tanh, exp,
//       per-wave partial softmax sums                                     | barrier 3
//   P3  softmax totals, loss, dZ4 of the own class tiles, the own tiles' share of dH3 (a split over K)  | barrier 4
//   P4  dH3 = sum of the four shares (fixed order) -> dZ3; dZ2 tile w (waves 0, 1) while waves 2, 3 start on dW4
//                                                                         | barrier 5
//   P5  dZ1 tile w; then the weight and bias gradients from the [neuron][row] copies, dealt round-robin (dW1 tile w
//       needs only this wave's own dZ1 tile: no sixth barrier)
// Stamps of the one-wave chain this replaces (tools/diag_train_phases.py, round 5): 26.1 us on wave 0, 13.7 of them MFMA
// issue, the other three waves waiting 18 us for dZ4.
constexpr int kXT = 64 * 16;   // floats of one exchange tile

__device__ __forceinline__ void xch_put(float *x, int t, const f32x16 &v, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    *reinterpret_cast<f32x4 *>(x + ((t * 4 + q) * 64 + lane) * 4) = o;
  }
}
__device__ __forceinline__ f32x16 xch_get(const float *x, int t, int lane) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 i = *reinterpret_cast<const f32x4 *>(x + ((t * 4 + q) * 64 + lane) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = i[e];
  }
  return v;
}
__device__ __forceinline__ f32x16 zero_tile() {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
  return v;
}

template <int MTC, bool DROP>
__global__ __launch_bounds__(256) void train_fwd_bwd_kernel(const TrainParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const hs0 = lds;                       // x        [32][36]
  float *const hs1 = hs0 + 32 * kS;             // h1 (after dropout) [128][36]
  float *const hs2 = hs1 + kH1 * kS;            // h2       [64][36]
  float *const hs3 = hs2 + kH2 * kS;            // h3       [32][36]
  float *const zs4 = hs3 + kH3 * kS;            // dZ4 [32 MTC][36]
  float *const zs3 = zs4 + 32 * MTC * kS;       // dZ3 [32][36]
  float *const zs2 = zs3 + kH3 * kS;            // dZ2 [64][36]
  float *const zs1 = zs2 + kH2 * kS;            // dZ1 [128][36]
  float *const xa = zs1 + kH1 * kS;             // exchange: h1 (4 tiles); later the waves' shares of dH3 (4 tiles)
  float *const xb = xa + 4 * kXT;               // exchange: h2 (2 tiles)
  float *const xc = xb + 2 * kXT;               // exchange: dZ2 (2 tiles)
  float *const sm = xc + 2 * kXT;               // [4 waves][sum_e | p_label][32 rows]
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, row = lane & 31, half = lane >> 5;
  // global row of the batch; the rows a ragged last tile lacks shadow the last real row (every load stays in range) and
  // are given no weight in the loss and in dZ4 -- from there on everything they would contribute is zero
  const bool valid = static_cast<int64_t>(blockIdx.x) * 32 + row < p.batch;
  const int64_t grow = valid ? static_cast<int64_t>(blockIdx.x) * 32 + row : p.batch - 1;
  const float row_weight = valid ? p.inv_batch : 0.f;
  float *const g = p.gpart + static_cast<int64_t>(blockIdx.x) * p.grad_floats;
  constexpr int CP = 32 * MTC;
  constexpr int OWN = (MTC + 3) / 4;            // class tiles of a wave: w, w + 4
  constexpr int GF = 256;                       // floats of one fragment group (64 lanes x 4)
  // offsets of the padded parameter arrays inside a gradient slab (the Adam kernel uses the same)
  constexpr int oW1 = 0, oW2 = oW1 + 32 * kH1, oW3 = oW2 + kH1 * kH2, oW4 = oW3 + kH2 * kH3, ob1 = oW4 + kH3 * CP,
                ob2 = ob1 + kH1, ob3 = ob2 + kH2, ob4 = ob3 + kH3;
  const float keep = 1.0f / p.inv_keep;
  // this lane's row of the three dropout masks (DROP only)
  const float *const mk1 = DROP ? p.mask[0] + grow * kH1 : nullptr;
  const float *const mk2 = DROP ? p.mask[1] + grow * kH2 : nullptr;
  const float *const mk3 = DROP ? p.mask[2] + grow * kH3 : nullptr;
  NNV_TRAIN_STAMP(16 * w + 0);

  // ---------------- P0: layer 1, tile w ----------------
  f32x16 xt[kT0];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = neuron_of(r, half);
    xt[0][r] = k < p.n_in ? p.x[grow * p.n_in + k] : 0.f;
  }
  f32x4 f1[kT0 * 4];
  load_frags<kT0 * 4>(p.wf[0] + w * (kT0 * 4 * GF), lane, f1);
  f32x4 f2[kT1 * 4];                      // the NEXT phase's fragments are requested before this phase's MFMAs
  if (w < kT2) load_frags<kT1 * 4>(p.wf[1] + w * (kT1 * 4 * GF), lane, f2);
  if (w == 0) put_tile(hs0, 0, xt[0], row, half);
  f32x16 h1w = chain<kT0>(xt, f1, bias_tile(p.bias[0], w, half));
  activate<DROP>(h1w, mk1, w, half, p.inv_keep);
  put_tile(hs1, w, h1w, row, half);
  xch_put(xa, w, h1w, lane);
  f32x4 f3[kT2 * 4];
  load_frags<kT2 * 4>(p.wf[2], lane, f3);
  NNV_TRAIN_STAMP(16 * w + 8);
  __syncthreads();   // 1: h1
  NNV_TRAIN_STAMP(16 * w + 1);

  // ---------------- P1: layer 2, tile w (waves 0, 1) ----------------
  f32x4 f4[OWN * 4], b4[OWN * 4];         // class layer, own tiles: forward and backward fragments
#pragma unroll
  for (int j = 0; j < OWN; ++j) {
    const int mt = w + 4 * j;
    if (mt < MTC) {
      load_frags<4>(p.wf[3] + mt * (kT3 * 4 * GF), lane, f4 + 4 * j);
      load_frags<4>(p.wb[3] + mt * (4 * GF), lane, b4 + 4 * j);
    }
  }
  if (w < kT2) {
    f32x16 h1[kT1];
#pragma unroll
    for (int t = 0; t < kT1; ++t) h1[t] = xch_get(xa, t, lane);
    f32x16 a = chain<kT1>(h1, f2, bias_tile(p.bias[1], w, half));
    activate<DROP>(a, mk2, w, half, p.inv_keep);
    put_tile(hs2, w, a, row, half);
    xch_put(xb, w, a, lane);
  }
  NNV_TRAIN_STAMP(16 * w + 9);
  __syncthreads();   // 2: h2
  NNV_TRAIN_STAMP(16 * w + 2);

  // ---------------- P2: layer 3 (every wave), class tiles w, w + 4 ----------------
  f32x16 h2[kT2], h3[kT3];
#pragma unroll
  for (int t = 0; t < kT2; ++t) h2[t] = xch_get(xb, t, lane);
  f32x4 b3[kT3 * 4], b2[kT2 * 4];         // backward fragments of dZ2 tile (w & 1) and of dZ1 tile w
  load_frags<kT3 * 4>(p.wb[2] + (w & 1) * (kT3 * 4 * GF), lane, b3);
  load_frags<kT2 * 4>(p.wb[1] + w * (kT2 * 4 * GF), lane, b2);
  h3[0] = chain<kT2>(h2, f3, bias_tile(p.bias[2], 0, half));
  activate<DROP>(h3[0], mk3, 0, half, p.inv_keep);
  if (w == 0) put_tile(hs3, 0, h3[0], row, half);
  // class layer: p = tanh(z4); softmax cross-entropy ON p (the reference passes the tanh'd tensor as logits)
  f32x16 pz[OWN];
  const int label = p.labels[grow];
  float sum_e = 0.f, p_label = 0.f;
#pragma unroll
  for (int j = 0; j < OWN; ++j) {
    const int mt = w + 4 * j;
    if (mt < MTC) {
      pz[j] = chain<kT3>(h3, f4 + 4 * j, bias_tile(p.bias[3], mt, half));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * mt + neuron_of(r, half);
        const float t = fast_tanh(pz[j][r]);
        pz[j][r] = t;
        if (c < p.n_classes) sum_e += __builtin_amdgcn_exp2f(t * 1.4426950408889634f);   // |t| <= 1: no max subtraction needed
        p_label += c == label ? t : 0.f;
      }
    } else {
      pz[j] = zero_tile();
    }
  }
  sum_e += __shfl_xor(sum_e, 32);
  p_label += __shfl_xor(p_label, 32);
  if (lane < 32) {
    sm[(2 * w + 0) * 32 + row] = sum_e;
    sm[(2 * w + 1) * 32 + row] = p_label;
  }
  NNV_TRAIN_STAMP(16 * w + 10);
  __syncthreads();   // 3: softmax shares (and h3 in hs3)
  NNV_TRAIN_STAMP(16 * w + 3);

  // ---------------- P3: loss, dZ4 of the own tiles, their share of dH3 ----------------
  sum_e = (sm[0 * 32 + row] + sm[2 * 32 + row]) + (sm[4 * 32 + row] + sm[6 * 32 + row]);
  p_label = (sm[1 * 32 + row] + sm[3 * 32 + row]) + (sm[5 * 32 + row] + sm[7 * 32 + row]);
  if (w == 0) {
    float loss = __builtin_amdgcn_logf(sum_e) * 0.6931471805599453f - p_label;   // -log softmax(p)[label]
    loss = (half == 0 && valid) ? loss : 0.f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
    if (lane == 0) {
      p.cost_part[blockIdx.x] = loss * p.inv_batch;
      if (p.cost_acc) p.cost_acc[blockIdx.x] += static_cast<double>(loss * p.inv_batch);   // own slot, stream-ordered steps
    }
  }
  // dZ4 = (softmax(p) - y) / B * (1 - p^2), in place
  const float inv_sum = 1.0f / sum_e;
  f32x16 share = zero_tile();
#pragma unroll
  for (int j = 0; j < OWN; ++j) {
    const int mt = w + 4 * j;
    if (mt < MTC) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * mt + neuron_of(r, half);
        const float t = pz[j][r];
        const float sme = c < p.n_classes ? __builtin_amdgcn_exp2f(t * 1.4426950408889634f) * inv_sum : 0.f;
        pz[j][r] = (sme - (c == label ? 1.0f : 0.f)) * row_weight * (1.0f - t * t);
      }
      put_tile(zs4, mt, pz[j], row, half);
      share = chain_one(pz[j], b4 + 4 * j, share);
    }
  }
  xch_put(xa, w, share, lane);            // xa's h1 tiles were last read before barrier 2
  NNV_TRAIN_STAMP(16 * w + 11);
  __syncthreads();   // 4: dZ4, the shares of dH3
  NNV_TRAIN_STAMP(16 * w + 4);

  // ---------------- P4: dZ3 (every wave); dZ2 tile w (waves 0, 1); dW4's first tiles (waves 2, 3) ----------------
  f32x16 d3[kT3];
  {
    const f32x16 s0 = xch_get(xa, 0, lane), s1 = xch_get(xa, 1, lane), s2 = xch_get(xa, 2, lane), s3 = xch_get(xa, 3, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) d3[0][r] = (s0[r] + s1[r]) + (s2[r] + s3[r]);
  }
  through_tanh<DROP>(d3[0], h3[0], mk3, 0, half, p.inv_keep, keep);
  if (w < kT2) {
    if (w == 0) put_tile(zs3, 0, d3[0], row, half);
    f32x16 d2w = chain<kT3>(d3, b3, zero_tile());
    f32x16 hsel;
#pragma unroll
    for (int r = 0; r < 16; ++r) hsel[r] = w == 0 ? h2[0][r] : h2[1][r];
    through_tanh<DROP>(d2w, hsel, mk2, w, half, p.inv_keep, keep);
    put_tile(zs2, w, d2w, row, half);
    xch_put(xc, w, d2w, lane);
  } else {
    const int jt = w - kT2;                // class tiles 0, 1 (MTC permitting); the rest joins the pool of P5
    if (jt < MTC) grad_tile(hs3, zs4, 0, jt, lane, g + oW4, CP);
    bias_grad(zs4, MTC, lane, g + ob4, jt, 2);
  }
  NNV_TRAIN_STAMP(16 * w + 12);
  __syncthreads();   // 5: dZ3, dZ2
  NNV_TRAIN_STAMP(16 * w + 5);

  // ---------------- P5: dZ1 tile w, then the gradients ----------------
  {
    f32x16 d2[kT2];
#pragma unroll
    for (int t = 0; t < kT2; ++t) d2[t] = xch_get(xc, t, lane);
    f32x16 d1w = chain<kT2>(d2, b2, zero_tile());
    through_tanh<DROP>(d1w, h1w, mk1, w, half, p.inv_keep, keep);
    put_tile(zs1, w, d1w, row, half);
  }
  // the pool: dW2 (8 tiles), dW3 (2), dW4 from class tile 2 on -- job k to wave k mod 4
  constexpr int J2 = kT1 * kT2, J3 = J2 + kT2, JN = J3 + (MTC > 2 ? MTC - 2 : 0);
  for (int k = w; k < JN; k += 4) {
    if (k < J2) grad_tile(hs1, zs2, k / kT2, k % kT2, lane, g + oW2, kH2);
    else if (k < J3) grad_tile(hs2, zs3, k - J2, 0, lane, g + oW3, kH3);
    else grad_tile(hs3, zs4, 0, k - J3 + 2, lane, g + oW4, CP);
  }
  if (w >= 2) bias_grad(zs2, kT2, lane, g + ob2, w - 2, 2);
  if (w == 1) bias_grad(zs3, kT3, lane, g + ob3);
  // dW1 / db1 tile w: x (behind barrier 1) and this wave's OWN dZ1 tile -- a wave's LDS accesses complete in order
  __builtin_amdgcn_wave_barrier();
  grad_tile(hs0, zs1, 0, w, lane, g + oW1, kH1);
  bias_grad(zs1, w + 1, lane, g + ob1, w, 4);
  NNV_TRAIN_STAMP(16 * w + 6);
}

struct AdamParams {
  float *w;          // canonical padded parameters: W1 [32][128] | W2 [128][64] | W3 [64][32] | W4 [32][CP] | b1 | b2 | b3 | b4
  float *m, *v;      // Adam moments, same layout
  float *wf[4], *wb[4];
  const float *gpart;
  int32_t n_wg;
  int64_t grad_floats;
  int32_t cp;        // padded class count
  float lr_t, beta1, beta2, eps, beta_reg;
  const float *cost_part;
  float *cost_out;   // nullable: cost of this step = sum of the workgroups' cross-entropy parts + beta * regulariser
  double *reg_acc;   // nullable [gridDim.x]: running sums of the blocks' regulariser shares (an epoch's mean cost is what
                     // the reference prints: nnv_train_export adds the slots up, no per-step assembly)
  float *reg_part;   // [gridDim.x] the blocks' shares of the regulariser
  unsigned *done;    // blocks finished (the last one assembles the cost and resets it)
  int32_t apply;     // 0: gradients only (debug entry): w, m, v untouched, the summed gradient is written to gsum
  float *gsum;       // [grad_floats] (apply == 0)
};

__global__ __launch_bounds__(256) void train_adam_kernel(const AdamParams p) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int cp = p.cp;
  const int64_t oW[5] = {0, 32 * kH1, 32 * kH1 + kH1 * kH2, 32 * kH1 + kH1 * kH2 + kH2 * kH3,
                         32 * kH1 + kH1 * kH2 + kH2 * kH3 + static_cast<int64_t>(kH3) * cp};
  float reg = 0.f;
  if (i < p.grad_floats) {
    float gsum = 0.f;
    for (int w = 0; w < p.n_wg; ++w) gsum += p.gpart[static_cast<int64_t>(w) * p.grad_floats + i];
    const float wv = p.w[i];
    int layer = -1;            // 0..3 for a weight
    for (int l = 0; l < 4; ++l)
      if (i >= oW[l] && i < oW[l + 1]) layer = l;
    if (layer >= 0 && layer < 3) {                   // tf.nn.l2_loss on h1, h2, h3 only
      gsum += p.beta_reg * wv;
      reg = 0.5f * wv * wv;
    }
    if (!p.apply) {
      p.gsum[i] = gsum;
    } else {
      float m = p.m[i], v = p.v[i];
      m += (gsum - m) * (1.0f - p.beta1);
      v += (gsum * gsum - v) * (1.0f - p.beta2);
      const float nw = wv - p.lr_t * m / (sqrtf(v) + p.eps);
      p.m[i] = m;
      p.v[i] = v;
      p.w[i] = nw;
      if (layer >= 0) {
        const int mpad = layer == 0 ? kH1 : (layer == 1 ? kH2 : (layer == 2 ? kH3 : cp));
        const int ktn = layer == 0 ? kT0 : (layer == 1 ? kT1 : (layer == 2 ? kT2 : kT3));   // k-tiles (inputs)
        const int mtn = mpad / 32;
        const int64_t rel = i - oW[layer];
        const int k = static_cast<int>(rel / mpad), mm = static_cast<int>(rel % mpad);
        {  // forward fragments [mt][kt][q][lane][e]
          const int mt = mm >> 5, ml = mm & 31, kt = k >> 5, kk = k & 31;
          const int q = kk >> 3, h = (kk >> 2) & 1, e = kk & 3;
          p.wf[layer][((((mt * ktn + kt) * 4 + q) * 64) + ml + 32 * h) * 4 + e] = nw;
        }
        {  // backward fragments [kt][mt][q][lane][e]
          const int kt = k >> 5, kl = k & 31, mt = mm >> 5, ml = mm & 31;
          const int q = ml >> 3, h = (ml >> 2) & 1, e = ml & 3;
          p.wb[layer][((((kt * mtn + mt) * 4 + q) * 64) + kl + 32 * h) * 4 + e] = nw;
        }
      }
    }
  }
  // regulariser of the cost: the block's share to its own slot (no atomics on values: the sum order is fixed).  A
  // training step only ADDS it to the block's running sum.  When the caller asks for this step's cost, the LAST block to
  // finish (a ticket counter behind a fence) assembles it from the shares and resets the counter -- an atomic round trip
  // and a second pass that the default path does not pay (8.7 -> 5.x us per launch).
  __shared__ float wave_reg[4];
  __shared__ unsigned ticket;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) reg += __shfl_xor(reg, off);
  if ((threadIdx.x & 63) == 0) wave_reg[threadIdx.x >> 6] = reg;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float block_reg = (wave_reg[0] + wave_reg[1]) + (wave_reg[2] + wave_reg[3]);
    if (p.reg_acc) p.reg_acc[blockIdx.x] += static_cast<double>(block_reg);
    if (p.cost_out) {
      p.reg_part[blockIdx.x] = block_reg;
      __threadfence();
      ticket = atomicAdd(p.done, 1u);
    }
  }
  if (!p.cost_out) return;   // kernel-uniform
  __syncthreads();
  if (ticket == gridDim.x - 1) {   // block-uniform: the last block sums the shares in parallel (fixed order per lane)
    __threadfence();
    float r = 0.f;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) r += __builtin_nontemporal_load(p.reg_part + b);
    float c = 0.f;
    for (int w = threadIdx.x; w < p.n_wg; w += 256) c += p.cost_part[w];
    float t = c + p.beta_reg * r;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    __syncthreads();                       // wave_reg is reused
    if ((threadIdx.x & 63) == 0) wave_reg[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      *p.cost_out = (wave_reg[0] + wave_reg[1]) + (wave_reg[2] + wave_reg[3]);
      *p.done = 0u;
    }
  }
}

}  // namespace
}  // namespace nnv

struct nnv_train_handle {
  int32_t n_in, n_classes, cp, mtc, batch, n_wg, device;
  int64_t grad_floats;
  float lr, beta_reg, keep_prob;
  int64_t step;              // Adam's t
  float *dev;                // one allocation: w | m | v | wf[4] | wb[4] | gpart | gsum | cost_part | reg_part | done | scratch | cost_acc | reg_acc
  float *w, *m, *v, *wf[4], *wb[4], *gpart, *gsum, *cost_part, *reg_part, *cost_scratch;
  double *cost_acc, *reg_acc;   // [n_wg], [adam_blocks]: running sums since the last export
  unsigned *done;
  int32_t adam_blocks;
  int64_t off_w[4], off_b[4];   // inside a parameter array
  int32_t kdim[4], mdim[4];     // padded [in][out] of the four layers
  int32_t kreal[4], mreal[4];
};

namespace {

using nnv::kH1;
using nnv::kH2;
using nnv::kH3;

template <int MTC>
constexpr int fwd_bwd_lds_bytes() {   // hs0..hs3 + dZ4..dZ1, 8 exchange tiles, the softmax shares
  return ((32 + kH1 + kH2 + kH3 + MTC * 32 + kH3 + kH2 + kH1) * nnv::kS + 8 * nnv::kXT + 8 * 32) * 4;
}

template <int MTC, bool DROP>
void launch_one(const nnv::TrainParams &p, int n_wg, hipStream_t s) {
  hipLaunchKernelGGL((nnv::train_fwd_bwd_kernel<MTC, DROP>), dim3(n_wg), dim3(256), fwd_bwd_lds_bytes<MTC>(), s, p);
}
template <int MTC>
void launch_fwd_bwd(const nnv::TrainParams &p, int n_wg, hipStream_t s) {
  if (p.mask[0]) launch_one<MTC, true>(p, n_wg, s);
  else launch_one<MTC, false>(p, n_wg, s);
}

// > 64 KB of dynamic LDS must be opted into per function AND per device: done by nnv_train_create on the handle's
// device for both instantiations of its class width (idempotent, off the per-step path).
template <int MTC>
bool optin_lds() {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(nnv::train_fwd_bwd_kernel<MTC, false>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, fwd_bwd_lds_bytes<MTC>()) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void *>(nnv::train_fwd_bwd_kernel<MTC, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, fwd_bwd_lds_bytes<MTC>()) == hipSuccess;
}
bool optin_lds_for(int mtc) {
  switch (mtc) {
    case 1: return optin_lds<1>();
    case 2: return optin_lds<2>();
    case 3: return optin_lds<3>();
    case 4: return optin_lds<4>();
    case 5: return optin_lds<5>();
    case 6: return optin_lds<6>();
    case 7: return optin_lds<7>();
    case 8: return optin_lds<8>();
    default: return false;
  }
}

nnv_status run_fwd_bwd(const nnv_train_handle *h, const float *x, const int32_t *labels, const float *const masks[3],
                       bool training_step, hipStream_t s) {
  nnv::TrainParams p{};
  p.x = x;
  p.labels = labels;
  p.n_in = h->n_in;
  p.n_classes = h->n_classes;
  for (int l = 0; l < 4; ++l) {
    p.wf[l] = h->wf[l];
    p.wb[l] = h->wb[l];
    p.bias[l] = h->w + h->off_b[l];
  }
  p.gpart = h->gpart;
  p.grad_floats = h->grad_floats;
  p.cost_part = h->cost_part;
  p.cost_acc = training_step ? h->cost_acc : nullptr;
  for (int l = 0; l < 3; ++l) p.mask[l] = masks ? masks[l] : nullptr;
  p.inv_keep = 1.0f / h->keep_prob;
  p.inv_batch = 1.0f / static_cast<float>(h->batch);
  p.batch = h->batch;
  switch (h->mtc) {
    case 1: launch_fwd_bwd<1>(p, h->n_wg, s); break;
    case 2: launch_fwd_bwd<2>(p, h->n_wg, s); break;
    case 3: launch_fwd_bwd<3>(p, h->n_wg, s); break;
    case 4: launch_fwd_bwd<4>(p, h->n_wg, s); break;
    case 5: launch_fwd_bwd<5>(p, h->n_wg, s); break;
    case 6: launch_fwd_bwd<6>(p, h->n_wg, s); break;
    case 7: launch_fwd_bwd<7>(p, h->n_wg, s); break;
    case 8: launch_fwd_bwd<8>(p, h->n_wg, s); break;
    default: return NNV_ERR_UNSUPPORTED_SHAPE;
  }
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

nnv_status run_adam(nnv_train_handle *h, bool apply, float *cost_out, hipStream_t s) {
  nnv::AdamParams a{};
  a.w = h->w;
  a.m = h->m;
  a.v = h->v;
  for (int l = 0; l < 4; ++l) {
    a.wf[l] = h->wf[l];
    a.wb[l] = h->wb[l];
  }
  a.gpart = h->gpart;
  a.n_wg = h->n_wg;
  a.grad_floats = h->grad_floats;
  a.cp = h->cp;
  a.beta1 = 0.9f;
  a.beta2 = 0.999f;
  a.eps = 1e-8f;
  a.beta_reg = h->beta_reg;
  a.cost_part = h->cost_part;
  a.cost_out = cost_out;
  a.reg_acc = apply ? h->reg_acc : nullptr;
  a.reg_part = h->reg_part;
  a.done = h->done;
  a.apply = apply ? 1 : 0;
  a.gsum = h->gsum;
  if (apply) {
    h->step += 1;   // TF-1: lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), in double on the host
    const double t = static_cast<double>(h->step);
    a.lr_t = static_cast<float>(static_cast<double>(h->lr) * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
  }
  hipLaunchKernelGGL(nnv::train_adam_kernel, dim3(h->adam_blocks), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace

extern "C" {

int32_t nnv_train_create(const nnv_train_desc *d, nnv_train_handle **out) {
  if (!d || !out) return NNV_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (d->n_input < 1 || d->n_classes < 2 || d->batch_size < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (!(d->keep_prob > 0.f && d->keep_prob <= 1.f) || !(d->learning_rate > 0.f)) return NNV_ERR_INVALID_ARGUMENT;
  for (int l = 0; l < 4; ++l)
    if (!d->W_host[l] || !d->b_host[l]) return NNV_ERR_INVALID_ARGUMENT;
  // the reference's architecture only: hidden widths 128, 64, 32; rows of at most 32 soft bits; at most 256 classes;
  // any batch up to 32 Ki rows (its default is 256; a ragged last 32-row tile is masked)
  if (d->n_hidden[0] != kH1 || d->n_hidden[1] != kH2 || d->n_hidden[2] != kH3) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (d->n_input > 32 || d->n_classes > 256 || d->batch_size > 32 * 1024) return NNV_ERR_UNSUPPORTED_SHAPE;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NNV_ERR_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return NNV_ERR_NO_DEVICE;
  nnv_train_handle *h = new (std::nothrow) nnv_train_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  memset(h, 0, sizeof(*h));
  h->n_in = d->n_input;
  h->n_classes = d->n_classes;
  h->mtc = (d->n_classes + 31) / 32;
  h->cp = 32 * h->mtc;
  h->batch = d->batch_size;
  h->n_wg = (d->batch_size + 31) / 32;
  h->device = dev;
  h->lr = d->learning_rate;
  h->beta_reg = d->beta_regularization;
  h->keep_prob = d->keep_prob;
  const int kd[4] = {32, kH1, kH2, kH3}, md[4] = {kH1, kH2, kH3, h->cp};
  const int kr[4] = {d->n_input, kH1, kH2, kH3}, mr[4] = {kH1, kH2, kH3, d->n_classes};
  int64_t off = 0;
  for (int l = 0; l < 4; ++l) {
    h->kdim[l] = kd[l];
    h->mdim[l] = md[l];
    h->kreal[l] = kr[l];
    h->mreal[l] = mr[l];
    h->off_w[l] = off;
    off += static_cast<int64_t>(kd[l]) * md[l];
  }
  for (int l = 0; l < 4; ++l) {
    h->off_b[l] = off;
    off += md[l];
  }
  h->grad_floats = off;
  const int64_t frag = h->off_b[0];                       // all four weight matrices, padded
  h->adam_blocks = static_cast<int32_t>((off + 255) / 256);
  const int64_t total = 3 * off + 2 * frag + static_cast<int64_t>(h->n_wg) * off + off + 3 * (h->n_wg + h->adam_blocks) + 4 + 64;
  if (hipMalloc(reinterpret_cast<void **>(&h->dev), total * sizeof(float)) != hipSuccess) {
    delete h;
    return NNV_ERR_OUT_OF_MEMORY;
  }
  float *q = h->dev;
  h->w = q; q += off;
  h->m = q; q += off;
  h->v = q; q += off;
  for (int l = 0; l < 4; ++l) { h->wf[l] = q; q += static_cast<int64_t>(kd[l]) * md[l]; }
  for (int l = 0; l < 4; ++l) { h->wb[l] = q; q += static_cast<int64_t>(kd[l]) * md[l]; }
  h->gpart = q; q += static_cast<int64_t>(h->n_wg) * off;
  h->gsum = q; q += off;
  h->cost_part = q; q += h->n_wg;
  h->reg_part = q; q += h->adam_blocks;
  h->done = reinterpret_cast<unsigned *>(q); q += 1;   // zero in the host image below: the kernel resets it itself
  h->cost_scratch = q; q += 1;
  q += (q - h->dev) & 1;                               // doubles: 8-byte aligned (hipMalloc's base is)
  h->cost_acc = reinterpret_cast<double *>(q); q += 2 * h->n_wg;
  h->reg_acc = reinterpret_cast<double *>(q); q += 2 * h->adam_blocks;
  // host image: canonical (zero padded) + both fragment orders
  std::vector<float> host(static_cast<size_t>(total), 0.f);
  float *hw = host.data();
  float *hf = host.data() + 3 * off, *hb = hf + frag;
  int64_t fo = 0;
  for (int l = 0; l < 4; ++l) {
    const int K = kd[l], M = md[l], KT = K / 32, MT = M / 32;
    for (int k = 0; k < kr[l]; ++k)
      for (int m = 0; m < mr[l]; ++m) {
        const float v = d->W_host[l][static_cast<size_t>(k) * mr[l] + m];
        hw[h->off_w[l] + static_cast<int64_t>(k) * M + m] = v;
        {
          const int mt = m >> 5, ml = m & 31, kt = k >> 5, kk = k & 31, qq = kk >> 3, hh = (kk >> 2) & 1, e = kk & 3;
          hf[fo + ((((static_cast<int64_t>(mt) * KT + kt) * 4 + qq) * 64) + ml + 32 * hh) * 4 + e] = v;
        }
        {
          const int kt = k >> 5, kl = k & 31, mt = m >> 5, ml = m & 31, qq = ml >> 3, hh = (ml >> 2) & 1, e = ml & 3;
          hb[fo + ((((static_cast<int64_t>(kt) * MT + mt) * 4 + qq) * 64) + kl + 32 * hh) * 4 + e] = v;
        }
      }
    for (int m = 0; m < mr[l]; ++m) hw[h->off_b[l] + m] = d->b_host[l][m];
    fo += static_cast<int64_t>(K) * M;
  }
  if (!optin_lds_for(h->mtc) ||
      hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->dev);
    delete h;
    return NNV_ERR_HIP;
  }
  *out = h;
  return NNV_OK;
}

int32_t nnv_train_free(nnv_train_handle *h) {
  if (!h) return NNV_OK;
  const hipError_t e = hipFree(h->dev);
  delete h;
  return e == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_train_steps(nnv_train_handle *h, const float *x_dev, const int32_t *labels_dev, int32_t n_steps,
                        const float *const masks_dev[3], float *costs_dev, void *stream) {
  if (!h || n_steps < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (n_steps == 0) return NNV_OK;
  if (!x_dev || !labels_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (h->keep_prob < 1.f && !masks_dev) return NNV_ERR_INVALID_ARGUMENT;   // dropout needs its masks (the caller draws them)
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int widths[3] = {kH1, kH2, kH3};
  for (int t = 0; t < n_steps; ++t) {
    const float *mk[3] = {nullptr, nullptr, nullptr};
    if (masks_dev)
      for (int l = 0; l < 3; ++l)
        mk[l] = masks_dev[l] ? masks_dev[l] + static_cast<int64_t>(t) * h->batch * widths[l] : nullptr;
    nnv_status st = run_fwd_bwd(h, x_dev + static_cast<int64_t>(t) * h->batch * h->n_in,
                                labels_dev + static_cast<int64_t>(t) * h->batch, masks_dev ? mk : nullptr, true, s);
    if (st != NNV_OK) return st;
    st = run_adam(h, true, costs_dev ? costs_dev + t : nullptr, s);
    if (st != NNV_OK) return st;
  }
  return NNV_OK;
}

int32_t nnv_train_gradients(nnv_train_handle *h, const float *x_dev, const int32_t *labels_dev,
                            const float *const masks_dev[3], float *cost_host, float *const gW_host[4],
                            float *const gb_host[4]) {
  if (!h || !x_dev || !labels_dev) return NNV_ERR_INVALID_ARGUMENT;
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  nnv_status st = run_fwd_bwd(h, x_dev, labels_dev, masks_dev, false, nullptr);
  if (st != NNV_OK) return st;
  st = run_adam(h, false, h->cost_scratch, nullptr);
  if (st != NNV_OK) return st;
  if (hipDeviceSynchronize() != hipSuccess) return NNV_ERR_HIP;
  std::vector<float> g(static_cast<size_t>(h->grad_floats));
  float c = 0.f;
  if (hipMemcpy(g.data(), h->gsum, g.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
  if (hipMemcpy(&c, h->cost_scratch, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
  if (cost_host) *cost_host = c;
  for (int l = 0; l < 4; ++l) {
    if (gW_host && gW_host[l])
      for (int k = 0; k < h->kreal[l]; ++k)
        for (int m = 0; m < h->mreal[l]; ++m)
          gW_host[l][static_cast<size_t>(k) * h->mreal[l] + m] = g[h->off_w[l] + static_cast<int64_t>(k) * h->mdim[l] + m];
    if (gb_host && gb_host[l])
      for (int m = 0; m < h->mreal[l]; ++m) gb_host[l][m] = g[h->off_b[l] + m];
  }
  return NNV_OK;
}

int32_t nnv_train_export(const nnv_train_handle *h, float *const W_host[4], float *const b_host[4], double *cost_sum_host) {
  if (!h) return NNV_ERR_INVALID_ARGUMENT;
  if (hipDeviceSynchronize() != hipSuccess) return NNV_ERR_HIP;
  std::vector<float> w(static_cast<size_t>(h->grad_floats));
  if (hipMemcpy(w.data(), h->w, w.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
  for (int l = 0; l < 4; ++l) {
    if (W_host && W_host[l])
      for (int k = 0; k < h->kreal[l]; ++k)
        for (int m = 0; m < h->mreal[l]; ++m)
          W_host[l][static_cast<size_t>(k) * h->mreal[l] + m] = w[h->off_w[l] + static_cast<int64_t>(k) * h->mdim[l] + m];
    if (b_host && b_host[l])
      for (int m = 0; m < h->mreal[l]; ++m) b_host[l][m] = w[h->off_b[l] + m];
  }
  if (cost_sum_host) {   // sum of the step costs since the last export (the reference prints their mean per epoch), then reset
    const size_t n = static_cast<size_t>(h->n_wg) + static_cast<size_t>(h->adam_blocks);   // the two arrays are adjacent
    std::vector<double> acc(n);
    if (hipMemcpy(acc.data(), h->cost_acc, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
    double ce = 0.0, reg = 0.0;
    for (int i = 0; i < h->n_wg; ++i) ce += acc[static_cast<size_t>(i)];
    for (int i = 0; i < h->adam_blocks; ++i) reg += acc[static_cast<size_t>(h->n_wg + i)];
    *cost_sum_host = ce + static_cast<double>(h->beta_reg) * reg;
    if (hipMemset(h->cost_acc, 0, n * sizeof(double)) != hipSuccess) return NNV_ERR_HIP;
  }
  return NNV_OK;
}

#if defined(NNV_AB_BUILD) && defined(NNV_DIAG_TRAIN)
int32_t nnv_train_diag_read(unsigned long long *out64) {   // 64 stamps of the last step's block 0 (tools/diag_train_phases.py)
  if (hipDeviceSynchronize() != hipSuccess) return NNV_ERR_HIP;
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(nnv::g_train_diag), 64 * sizeof(unsigned long long)) == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}
#endif

}  // extern "C"
