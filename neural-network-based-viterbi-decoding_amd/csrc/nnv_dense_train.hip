// K8: the TRAINING STEP of the reference's bit-wise and sliding-window decoders for
// a dense network of 1..5 layers of any width, one hand-written launch per layer and direction.
//
//   pred = act(... act(x W1 + b1) ...), the activation after EVERY layer      src/fnn_viterbi_bitwise.py:388-397 (ReLU)
//                                                                             src/fnn_viterbi_seq_predict.py:404-412 (tanh)
//   cost = reduce_mean(sigmoid_cross_entropy_with_logits(logits=pred, labels=y)) + beta (l2_loss(h1) + l2_loss(h2))
//                                                  src/fnn_viterbi_bitwise.py:438-448, src/fnn_viterbi_seq_predict.py:475-485
//   tf.train.AdamOptimizer(learning_rate).minimize(cost)                                            :449-452 / :486-489
//
// Batches are small (the reference: 1000 rows x 1036 -> 30 -> 2, and 100 rows x 200 -> 500 x 4 -> 2), so a step is
// launch- and latency-shaped: PyTorch autograd spends ~60 tiny kernels on it (0.24 / 0.40 ms per step replayed from a HIP
// graph, the sliding-window loop bound by the HOST's graph launches).  Here a step is 2 L + 1 launches:
//
//   stage      the batch's rows (row stride and step stride are arguments: the sliding-window rows are overlapping
//              windows of ONE stream, window i = stream[2 i .. 2 i + 200)) into a zero-padded [Bp][Kp] tile buffer whose
//              column K holds 1: every bias is row K of its weight matrix, so no kernel below has a bias path and db is
//              a row of dW.
//              The NEXT step's batch is staged by extra blocks of this step's Adam launch (it depends on nothing the step
//              computes): a launch of its own only for the first step of a call.
//   forward    per layer: H_{l+1} = act(H_l W_l), one 32 x 32 output tile per workgroup, the contraction split over its
//              eight waves and summed through LDS in a fixed order.  The last layer's epilogue is the loss: sigmoid
//              cross-entropy on the activated outputs and dZ of the last layer, one share of the cost per 32 rows.
//   backward   per layer, ONE launch: dW_l = H_l^T dZ_l (tiles over [Kp][Np], contraction over the rows) and, for l > 0,
//              dZ_{l-1} = (dZ_l W_l^T) * act'(H_l) (tiles over [Bp][Kp]).
//   adam       one thread per parameter of all layers: + beta W where the regulariser names the matrix, TF-1 Adam.
//
// All operands are padded to 32 in both dimensions and row-major, so the four operand patterns of the three products
// (contraction index contiguous or not, for A and for B) are plain 16-byte or coalesced 4-byte loads; exact fp32
// (v_mfma_f32_32x32x2_f32).  Checked against oracle/dense_train_oracle.py (float64) in tests/test_dense_trainer.py.
#include <math.h>

#include <new>
#include <vector>

#include "nnv_internal.h"

namespace nnv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMaxL = 5;

__device__ __forceinline__ int row_of(int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); }

constexpr int kWaves = 8;            // waves of a tile's workgroup: the contraction is dealt over them (k-tile w, w + 8, ...)
constexpr int kThreads = 64 * kWaves;
constexpr int kChunk = 4;            // k-tiles whose operands a wave requests together before the first MFMA

// One 32 x 32 tile C[m0 .., n0 ..] = sum over k < 32 KT of A(m, k) B(k, n) by a workgroup of kWaves waves.
//   AK: A is stored with k contiguous (A[m * lda + k]), else with m contiguous (A[k * lda + m]);
//   BK: B is stored with k contiguous (B[n * ldb + k]), else with n contiguous (B[k * ldb + n]).
// The k order inside a 32-wide k-tile is the lane's choice as long as A and B agree: lane (i, half) takes
// k = 8 q + 4 half + e for the four MFMAs e of group q, i.e. four consecutive k -- one 16-byte load where k is contiguous.
// These products are a few k-tiles deep and their operands come from L2: what a wave has to hide is load LATENCY, so
// it requests the operands of up to kChunk of its k-tiles in one batch (128 registers) and only then starts the MFMA
// chain -- one exposed round trip per chunk (first version, one k-tile at a time on four waves: 2 us per k-tile).
// Returns true in the waves w < 4 that hold a finished quarter: registers r = 4 w .. 4 w + 3 of the tile in out[0..3],
// C[m0 + row_of(r, half)][n0 + i]; the partial tiles are added in a fixed order.
template <bool AK, bool BK>
__device__ __forceinline__ bool tile_product(const float *A, int lda, const float *B, int ldb, int m0, int n0, int KT,
                                             float *red, f32x4 &out) {
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kt0 = w; kt0 < KT; kt0 += kWaves * kChunk) {
    f32x4 a[kChunk][4], b[kChunk][4];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int kt = kt0 + u * kWaves;
      if (kt < KT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 32 * kt + 8 * q + 4 * half;
          if constexpr (AK) {
            a[u][q] = *reinterpret_cast<const f32x4 *>(A + static_cast<int64_t>(m0 + i) * lda + k);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][q][e] = A[static_cast<int64_t>(k + e) * lda + m0 + i];
          }
          if constexpr (BK) {
            b[u][q] = *reinterpret_cast<const f32x4 *>(B + static_cast<int64_t>(n0 + i) * ldb + k);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b[u][q][e] = B[static_cast<int64_t>(k + e) * ldb + n0 + i];
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the requests in front of the chain (the scheduler would sink them)
#pragma unroll
    for (int u = 0; u < kChunk; ++u)
      if (kt0 + u * kWaves < KT) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q][e], b[u][q][e], acc, 0, 0, 0);
      }
  }
  // the waves' partial tiles through LDS, [wave][group of 4 registers][lane] as 16-byte slots; wave w < 4 finishes group w
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    *reinterpret_cast<f32x4 *>(red + ((w * 4 + g) * 64 + lane) * 4) = v;
  }
  __syncthreads();
  if (w >= 4) return false;
  f32x4 p[kWaves];
#pragma unroll
  for (int v = 0; v < kWaves; ++v) p[v] = *reinterpret_cast<const f32x4 *>(red + ((v * 4 + w) * 64 + lane) * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) out[e] = ((p[0][e] + p[1][e]) + (p[2][e] + p[3][e])) + ((p[4][e] + p[5][e]) + (p[6][e] + p[7][e]));
  return true;
}

__device__ __forceinline__ float act_fn(float z, int act) {
  if (act == NNV_ACT_RELU) return fmaxf(z, 0.f);
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);   // tanh z = 1 - 2 / (1 + e^{2z})
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}
__device__ __forceinline__ float dact_from_output(float h, int act) {
  return act == NNV_ACT_RELU ? (h > 0.f ? 1.0f : 0.f) : 1.0f - h * h;
}

struct StageParams {
  const float *x;        // row m of this step: x[m * row_stride + k], k < K
  int64_t row_stride;
  float *h0;             // [Bp][Kp]
  int32_t B, K, Bp, Kp;
};
__device__ __forceinline__ void stage_block(const StageParams &p, int block) {
  const int64_t idx = static_cast<int64_t>(block) * 256 + threadIdx.x;
  if (idx >= static_cast<int64_t>(p.Bp) * p.Kp) return;
  const int m = static_cast<int>(idx / p.Kp), k = static_cast<int>(idx % p.Kp);
  float v = 0.f;
  if (m < p.B) v = k < p.K ? p.x[m * p.row_stride + k] : (k == p.K ? 1.0f : 0.f);
  p.h0[idx] = v;
}
__global__ __launch_bounds__(256) void dense_stage_kernel(const StageParams p) { stage_block(p, blockIdx.x); }

struct FwdParams {
  const float *h_in;     // [Bp][Kp]
  const float *w;        // [Kp][Np], row K = the bias
  float *h_out;          // [Bp][Np]
  int32_t Kp, Np, B, ones_col, act;   // ones_col: the column of h_out that carries the next layer's 1 (-1: last layer)
  // last layer only (dz != null; Np == 32, one workgroup per 32 rows): the loss and its gradient in the epilogue
  const int32_t *labels; // [B] class index: y = one_hot(label, n_out)
  float *dz;             // [Bp][32] dZ of the last layer, zeros in the padding
  int32_t n_out;
  float inv_count;       // 1 / (B n_out)
  float *ce_part;        // [Bp / 32] this step's cross-entropy shares (already * inv_count)
  double *ce_acc;        // nullable [Bp / 32]: their running sums over the steps
};
__global__ __launch_bounds__(kThreads) void dense_fwd_kernel(const FwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[kWaves * 4 * 64 * 4];
  __shared__ float ce_wave[4];
  const int nt = p.Np / 32;
  const int m0 = 32 * (blockIdx.x / nt), n0 = 32 * (blockIdx.x % nt);
  f32x4 c;
  const bool fin = tile_product<true, false>(p.h_in, p.Kp, p.w, p.Np, m0, n0, p.Kp / 32, red, c);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, n = n0 + (lane & 31), half = lane >> 5;
  float ce = 0.f;
  if (fin) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + row_of(4 * w + e, half);
      float h = act_fn(c[e], p.act);
      if (n == p.ones_col) h = m < p.B ? 1.0f : 0.f;
      p.h_out[static_cast<int64_t>(m) * p.Np + n] = h;
      if (p.dz) {   // kernel-uniform: sigmoid cross-entropy on the ACTIVATED output h, and dZ = dL/dh * act'(h)
        float d = 0.f;
        if (m < p.B && n < p.n_out) {
          const float y = n == p.labels[m] ? 1.0f : 0.f;
          const float ez = __expf(-fabsf(h));
          ce += fmaxf(h, 0.f) - h * y + log1pf(ez);
          const float sig = h >= 0.f ? 1.0f / (1.0f + ez) : ez / (1.0f + ez);
          d = (sig - y) * p.inv_count * dact_from_output(h, p.act);
        }
        p.dz[static_cast<int64_t>(m) * p.Np + n] = d;
      }
    }
  }
  if (!p.dz) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ce += __shfl_xor(ce, off);
  if (fin && lane == 0) ce_wave[w] = ce;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float part = ((ce_wave[0] + ce_wave[1]) + (ce_wave[2] + ce_wave[3])) * p.inv_count;
    p.ce_part[blockIdx.x] = part;
    if (p.ce_acc) p.ce_acc[blockIdx.x] += static_cast<double>(part);   // own slot, stream-ordered steps
  }
}

// softmax cross-entropy taken on the ACTIVATED outputs (the feed-forward decoder's loss, src/fnn_viterbi.py:549-553:
// the reference passes the tanh'd class layer as logits), class counts beyond one 32-wide tile: a launch of its own
// behind the last forward launch.  One wave per row, classes on the lanes (coalesced), three passes: max, sum, dZ.
struct SoftmaxLossParams {
  const float *pred;     // [Bp][Np] activated outputs
  const int32_t *labels; // [B]
  float *dz;             // [Bp][Np]: rows >= B are never written (they stay 0), columns >= n_out are written 0
  int32_t B, Np, n_out, act;
  float inv_b;           // 1 / B
  float *ce_part;        // [gridDim.x]
  double *ce_acc;        // nullable [gridDim.x]
};
__global__ __launch_bounds__(256) void dense_softmax_loss_kernel(const SoftmaxLossParams p) {
  __shared__ float ce_wave[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + w;
  float loss = 0.f;
  if (m < p.B) {
    const float *row = p.pred + static_cast<int64_t>(m) * p.Np;
    float *drow = p.dz + static_cast<int64_t>(m) * p.Np;
    const int label = p.labels[m];
    float mx = -__builtin_inff();
    for (int c = lane; c < p.n_out; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f, t_label = 0.f;
    for (int c = lane; c < p.n_out; c += 64) {
      const float t = row[c];
      sum += __expf(t - mx);
      t_label += c == label ? t : 0.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sum += __shfl_xor(sum, off);
      t_label += __shfl_xor(t_label, off);
    }
    loss = __logf(sum) + mx - t_label;                     // -log softmax(pred)[label], the same in every lane
    const float inv_sum = 1.0f / sum;
    for (int c = lane; c < p.Np; c += 64) {
      float d = 0.f;
      if (c < p.n_out) {
        const float t = row[c];
        d = (__expf(t - mx) * inv_sum - (c == label ? 1.0f : 0.f)) * p.inv_b * dact_from_output(t, p.act);
      }
      drow[c] = d;
    }
  }
  if (lane == 0) ce_wave[w] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float part = ((ce_wave[0] + ce_wave[1]) + (ce_wave[2] + ce_wave[3])) * p.inv_b;
    p.ce_part[blockIdx.x] = part;
    if (p.ce_acc) p.ce_acc[blockIdx.x] += static_cast<double>(part);
  }
}

struct BwdParams {
  const float *h_in;     // [Bp][Kp] this layer's input activations (column K = 1)
  const float *dz;       // [Bp][Np]
  const float *w;        // [Kp][Np]
  float *dw;             // [Kp][Np]
  float *dz_prev;        // nullable (layer 0): [Bp][Kp]
  int32_t Bp, Kp, Np, K, act, n_dw_tiles;
};
__global__ __launch_bounds__(kThreads) void dense_bwd_kernel(const BwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[kWaves * 4 * 64 * 4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  f32x4 c;
  if (static_cast<int>(blockIdx.x) < p.n_dw_tiles) {   // dW[i][j] = sum over the rows b of H[b][i] dZ[b][j]
    const int nt = p.Np / 32;
    const int i0 = 32 * (blockIdx.x / nt), j0 = 32 * (blockIdx.x % nt);
    if (!tile_product<false, false>(p.h_in, p.Kp, p.dz, p.Np, i0, j0, p.Bp / 32, red, c)) return;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      p.dw[static_cast<int64_t>(i0 + row_of(4 * w + e, half)) * p.Np + j0 + (lane & 31)] = c[e];
  } else {                                             // dZprev[b][i] = (sum over n of dZ[b][n] W[i][n]) act'(H[b][i]), i < K
    const int t = blockIdx.x - p.n_dw_tiles, it = p.Kp / 32;
    const int b0 = 32 * (t / it), i0 = 32 * (t % it);
    // the activations the epilogue multiplies by act'(.) are requested BEFORE the product (an L2-cold load otherwise
    // sits between the last MFMA and the store); only the finishing waves w < 4 own output registers
    const int i = i0 + (lane & 31);
    float hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (w < 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[e] = p.h_in[static_cast<int64_t>(b0 + row_of(4 * w + e, half)) * p.Kp + i];
    }
    if (!tile_product<true, true>(p.dz, p.Np, p.w, p.Np, b0, i0, p.Np / 32, red, c)) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t at = static_cast<int64_t>(b0 + row_of(4 * w + e, half)) * p.Kp + i;
      p.dz_prev[at] = i < p.K ? c[e] * dact_from_output(hv[e], p.act) : 0.f;
    }
  }
}

struct DenseAdamParams {
  float *w, *m, *v;      // all layers' padded matrices back to back, then the extra (decay-only) variables
  const float *dw;       // same layout (the extra part is not read)
  int64_t n_params, n_matrix;          // n_matrix: floats of the layer matrices (the extras follow)
  int64_t layer_end[kMaxL];            // exclusive end offsets of the layers
  int32_t n_layers, Np[kMaxL], K[kMaxL];
  uint32_t reg_mask;
  float lr_t, beta1, beta2, eps, beta_reg;
  int32_t apply;
  float *gsum;           // apply == 0: the gradient incl. beta W
  double *reg_acc;       // nullable [adam_blocks]
  float *reg_part;       // nullable [adam_blocks]: this step's shares (the caller wants this step's cost)
  // the NEXT step's batch is staged by extra blocks of this launch (it depends on nothing this step computes)
  StageParams next;
  int32_t n_stage_blocks;
};
__global__ __launch_bounds__(256) void dense_adam_kernel(const DenseAdamParams p) {
  if (static_cast<int>(blockIdx.x) < p.n_stage_blocks) {
    stage_block(p.next, blockIdx.x);
    return;
  }
  const int bid = blockIdx.x - p.n_stage_blocks;
  const int64_t i = static_cast<int64_t>(bid) * 256 + threadIdx.x;
  float reg = 0.f;
  if (i < p.n_params) {
    const float wv = p.w[i];
    float g;
    if (i < p.n_matrix) {
      int l = 0;
      while (l < p.n_layers - 1 && i >= p.layer_end[l]) ++l;
      const int64_t rel = i - (l ? p.layer_end[l - 1] : 0);
      g = p.dw[i];
      if (((p.reg_mask >> l) & 1u) && rel / p.Np[l] < p.K[l]) {   // tf.nn.l2_loss(W_l): the matrix, not its bias row
        g += p.beta_reg * wv;
        reg = 0.5f * wv * wv;
      }
    } else {                                                      // a variable only the regulariser touches
      g = p.beta_reg * wv;
      reg = 0.5f * wv * wv;
    }
    if (!p.apply) {
      p.gsum[i] = g;
    } else {
      float m = p.m[i], v = p.v[i];
      m += (g - m) * (1.0f - p.beta1);
      v += (g * g - v) * (1.0f - p.beta2);
      p.m[i] = m;
      p.v[i] = v;
      p.w[i] = wv - p.lr_t * m / (sqrtf(v) + p.eps);
    }
  }
  __shared__ float wave_reg[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) reg += __shfl_xor(reg, off);
  if ((threadIdx.x & 63) == 0) wave_reg[threadIdx.x >> 6] = reg;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float block_reg = (wave_reg[0] + wave_reg[1]) + (wave_reg[2] + wave_reg[3]);
    if (p.reg_acc) p.reg_acc[bid] += static_cast<double>(block_reg);
    if (p.reg_part) p.reg_part[bid] = block_reg;
  }
}

// A caller that wants THIS step's cost (tests, nnv_dense_train_gradients) pays one more small launch that adds the
// shares up in a fixed order; a training loop does not (an atomic ticket with a fence in every one of the sliding-window
// network's 3,584 Adam blocks doubled the step).
struct CostParams {
  const float *reg_part, *ce_part;
  int32_t n_reg, n_ce;
  float beta_reg;
  float *cost_out;
};
__global__ __launch_bounds__(256) void dense_cost_kernel(const CostParams p) {
  float r = 0.f, ce = 0.f;
  for (int b = threadIdx.x; b < p.n_reg; b += 256) r += p.reg_part[b];
  for (int b = threadIdx.x; b < p.n_ce; b += 256) ce += p.ce_part[b];
  float t = ce + p.beta_reg * r;
  __shared__ float wave_sum[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) *p.cost_out = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

}  // namespace
}  // namespace nnv

struct nnv_dense_train_handle {
  int32_t L, act, B, Bp, n_out, device, loss;
  int32_t K[nnv::kMaxL], N[nnv::kMaxL], Kp[nnv::kMaxL], Np[nnv::kMaxL];
  uint32_t reg_mask;
  float lr, beta_reg;
  int64_t step, n_matrix, n_extra, n_params, off_w[nnv::kMaxL];
  int32_t adam_blocks;
  float *dev;                       // one allocation
  float *w, *m, *v, *dw, *gsum;     // [n_params]
  float *h[nnv::kMaxL + 1];         // h[0] = the staged batch, h[l + 1] = layer l's output
  float *dz[nnv::kMaxL];
  float *reg_part, *ce_part, *cost_scratch;
  int32_t n_row_tiles;
  double *ce_acc, *reg_acc;
};

namespace {

int round_up32(int v) { return (v + 31) / 32 * 32; }

nnv::StageParams stage_params(const nnv_dense_train_handle *h, const float *x, int64_t row_stride) {
  return nnv::StageParams{x, row_stride, h->h[0], h->B, h->K[0], h->Bp, h->Kp[0]};
}
unsigned stage_blocks(const nnv_dense_train_handle *h) {
  return static_cast<unsigned>((static_cast<int64_t>(h->Bp) * h->Kp[0] + 255) / 256);
}

// forward (the loss in the last layer's epilogue) and backward of the batch that is ALREADY staged in h->h[0]
nnv_status dense_fwd_bwd(nnv_dense_train_handle *h, const int32_t *labels, bool training_step, hipStream_t s) {
  for (int l = 0; l < h->L; ++l) {
    const bool last = l + 1 == h->L;
    nnv::FwdParams p{};
    p.h_in = h->h[l];
    p.w = h->w + h->off_w[l];
    p.h_out = h->h[l + 1];
    p.Kp = h->Kp[l];
    p.Np = h->Np[l];
    p.B = h->B;
    p.ones_col = last ? -1 : h->N[l];
    p.act = h->act;
    if (last && h->loss == NNV_LOSS_SIGMOID_CE) {
      p.labels = labels;
      p.dz = h->dz[l];
      p.n_out = h->n_out;
      p.inv_count = 1.0f / (static_cast<float>(h->B) * static_cast<float>(h->n_out));
      p.ce_part = h->ce_part;
      p.ce_acc = training_step ? h->ce_acc : nullptr;
    }
    hipLaunchKernelGGL(nnv::dense_fwd_kernel, dim3((h->Bp / 32) * (h->Np[l] / 32)), dim3(nnv::kThreads), 0, s, p);
  }
  if (h->loss == NNV_LOSS_SOFTMAX_CE) {
    const int last = h->L - 1;
    nnv::SoftmaxLossParams p{h->h[h->L], labels, h->dz[last], h->B, h->Np[last], h->n_out, h->act,
                             1.0f / static_cast<float>(h->B), h->ce_part, training_step ? h->ce_acc : nullptr};
    hipLaunchKernelGGL(nnv::dense_softmax_loss_kernel, dim3((h->B + 3) / 4), dim3(256), 0, s, p);
  }
  for (int l = h->L - 1; l >= 0; --l) {
    nnv::BwdParams p{h->h[l], h->dz[l], h->w + h->off_w[l], h->dw + h->off_w[l], l > 0 ? h->dz[l - 1] : nullptr,
                     h->Bp, h->Kp[l], h->Np[l], h->K[l], h->act, (h->Kp[l] / 32) * (h->Np[l] / 32)};
    const int n_dh = l > 0 ? (h->Bp / 32) * (h->Kp[l] / 32) : 0;
    hipLaunchKernelGGL(nnv::dense_bwd_kernel, dim3(p.n_dw_tiles + n_dh), dim3(nnv::kThreads), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

// next_x: the NEXT step's rows, staged by extra blocks of the Adam launch (nullptr: none)
nnv_status dense_adam(nnv_dense_train_handle *h, bool apply, float *cost_out, const float *next_x, int64_t row_stride, hipStream_t s) {
  nnv::DenseAdamParams a{};
  a.w = h->w;
  a.m = h->m;
  a.v = h->v;
  a.dw = h->dw;
  a.n_params = h->n_params;
  a.n_matrix = h->n_matrix;
  a.n_layers = h->L;
  for (int l = 0; l < h->L; ++l) {
    a.layer_end[l] = h->off_w[l] + static_cast<int64_t>(h->Kp[l]) * h->Np[l];
    a.Np[l] = h->Np[l];
    a.K[l] = h->K[l];
  }
  a.reg_mask = h->reg_mask;
  a.beta1 = 0.9f;
  a.beta2 = 0.999f;
  a.eps = 1e-8f;
  a.beta_reg = h->beta_reg;
  a.apply = apply ? 1 : 0;
  a.gsum = h->gsum;
  a.reg_acc = apply ? h->reg_acc : nullptr;
  a.reg_part = cost_out ? h->reg_part : nullptr;
  a.n_stage_blocks = 0;
  if (next_x) {
    a.next = stage_params(h, next_x, row_stride);
    a.n_stage_blocks = static_cast<int32_t>(stage_blocks(h));
  }
  if (apply) {
    h->step += 1;   // TF-1: lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t), in double on the host
    const double t = static_cast<double>(h->step);
    a.lr_t = static_cast<float>(static_cast<double>(h->lr) * sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
  }
  hipLaunchKernelGGL(nnv::dense_adam_kernel, dim3(a.n_stage_blocks + h->adam_blocks), dim3(256), 0, s, a);
  if (cost_out) {
    nnv::CostParams c{h->reg_part, h->ce_part, h->adam_blocks, h->n_row_tiles, h->beta_reg, cost_out};
    hipLaunchKernelGGL(nnv::dense_cost_kernel, dim3(1), dim3(256), 0, s, c);
  }
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace

extern "C" {

int32_t nnv_dense_train_create(const nnv_dense_train_desc *d, nnv_dense_train_handle **out) {
  if (!d || !out) return NNV_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (d->n_layers < 1 || d->n_layers > nnv::kMaxL || d->batch_size < 1 || d->batch_size > (1 << 20)) return NNV_ERR_INVALID_ARGUMENT;
  if (d->activation != NNV_ACT_TANH && d->activation != NNV_ACT_RELU) return NNV_ERR_INVALID_ARGUMENT;
  if (!(d->learning_rate > 0.f) || d->n_extra < 0 || (d->n_extra > 0 && !d->extra_host)) return NNV_ERR_INVALID_ARGUMENT;
  for (int l = 0; l <= d->n_layers; ++l)
    if (d->dims[l] < 1 || d->dims[l] > NNV_WIDE_MAX_WIDTH) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (d->loss != NNV_LOSS_SIGMOID_CE && d->loss != NNV_LOSS_SOFTMAX_CE) return NNV_ERR_INVALID_ARGUMENT;
  // sigmoid cross-entropy lives in the last forward launch's epilogue (one 32-wide tile); softmax has its own launch
  if (d->loss == NNV_LOSS_SIGMOID_CE && d->dims[d->n_layers] > 32) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (d->dims[d->n_layers] > NNV_WIDE_MAX_CLASSES) return NNV_ERR_UNSUPPORTED_SHAPE;
  for (int l = 0; l < d->n_layers; ++l)
    if (!d->W_host[l] || !d->b_host[l]) return NNV_ERR_INVALID_ARGUMENT;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return NNV_ERR_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return NNV_ERR_NO_DEVICE;
  nnv_dense_train_handle *h = new (std::nothrow) nnv_dense_train_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  memset(h, 0, sizeof(*h));
  h->L = d->n_layers;
  h->act = d->activation;
  h->B = d->batch_size;
  h->Bp = round_up32(d->batch_size);
  h->n_out = d->dims[d->n_layers];
  h->loss = d->loss;
  h->device = dev;
  h->reg_mask = d->reg_layer_mask;
  h->lr = d->learning_rate;
  h->beta_reg = d->beta_regularization;
  int64_t off = 0;
  for (int l = 0; l < h->L; ++l) {
    h->K[l] = d->dims[l];
    h->N[l] = d->dims[l + 1];
    h->Kp[l] = round_up32(h->K[l] + 1);                               // + the column of ones that carries the bias
    h->Np[l] = l + 1 < h->L ? round_up32(h->N[l] + 1) : round_up32(h->N[l]);
    h->off_w[l] = off;
    off += static_cast<int64_t>(h->Kp[l]) * h->Np[l];
  }
  h->n_matrix = off;
  h->n_extra = d->n_extra;
  h->n_params = off + d->n_extra;
  h->adam_blocks = static_cast<int32_t>((h->n_params + 255) / 256);
  int64_t act_floats = static_cast<int64_t>(h->Bp) * h->Kp[0];
  for (int l = 0; l < h->L; ++l) act_floats += 2 * static_cast<int64_t>(h->Bp) * h->Np[l];   // h[l + 1] and dz[l]
  h->n_row_tiles = d->loss == NNV_LOSS_SOFTMAX_CE ? (h->B + 3) / 4 : h->Bp / 32;   // workgroups that write a share of the cost
  const int64_t total = 5 * h->n_params + act_floats + h->adam_blocks + h->n_row_tiles + 8 +
                        2 * (static_cast<int64_t>(h->n_row_tiles) + h->adam_blocks);
  if (hipMalloc(reinterpret_cast<void **>(&h->dev), total * sizeof(float)) != hipSuccess) {
    delete h;
    return NNV_ERR_OUT_OF_MEMORY;
  }
  float *q = h->dev;
  h->w = q; q += h->n_params;
  h->m = q; q += h->n_params;
  h->v = q; q += h->n_params;
  h->dw = q; q += h->n_params;
  h->gsum = q; q += h->n_params;
  h->h[0] = q; q += static_cast<int64_t>(h->Bp) * h->Kp[0];
  for (int l = 0; l < h->L; ++l) {
    h->h[l + 1] = q; q += static_cast<int64_t>(h->Bp) * h->Np[l];
    h->dz[l] = q; q += static_cast<int64_t>(h->Bp) * h->Np[l];
  }
  h->reg_part = q; q += h->adam_blocks;
  h->ce_part = q; q += h->n_row_tiles;
  h->cost_scratch = q; q += 1;
  q += (q - h->dev) & 1;                                             // doubles: 8-byte aligned
  h->ce_acc = reinterpret_cast<double *>(q); q += 2 * static_cast<int64_t>(h->n_row_tiles);
  h->reg_acc = reinterpret_cast<double *>(q); q += 2 * static_cast<int64_t>(h->adam_blocks);
  std::vector<float> host(static_cast<size_t>(h->n_params), 0.f);    // everything else starts as zeros
  for (int l = 0; l < h->L; ++l) {
    float *W = host.data() + h->off_w[l];
    for (int k = 0; k < h->K[l]; ++k)
      for (int n = 0; n < h->N[l]; ++n) W[static_cast<int64_t>(k) * h->Np[l] + n] = d->W_host[l][static_cast<size_t>(k) * h->N[l] + n];
    for (int n = 0; n < h->N[l]; ++n) W[static_cast<int64_t>(h->K[l]) * h->Np[l] + n] = d->b_host[l][n];
  }
  for (int64_t i = 0; i < h->n_extra; ++i) host[static_cast<size_t>(h->n_matrix + i)] = d->extra_host[i];
  if (hipMemset(h->dev, 0, static_cast<size_t>(total) * sizeof(float)) != hipSuccess ||
      hipMemcpy(h->w, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->dev);
    delete h;
    return NNV_ERR_HIP;
  }
  *out = h;
  return NNV_OK;
}

int32_t nnv_dense_train_free(nnv_dense_train_handle *h) {
  if (!h) return NNV_OK;
  const hipError_t e = hipFree(h->dev);
  delete h;
  return e == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_dense_train_steps(nnv_dense_train_handle *h, const float *x_dev, int64_t row_stride, int64_t step_stride,
                              const int32_t *labels_dev, int32_t n_steps, float *costs_dev, void *stream) {
  if (!h || n_steps < 0 || row_stride < 1 || step_stride < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (n_steps == 0) return NNV_OK;
  if (!x_dev || !labels_dev) return NNV_ERR_INVALID_ARGUMENT;
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(nnv::dense_stage_kernel, dim3(stage_blocks(h)), dim3(256), 0, s, stage_params(h, x_dev, row_stride));
  for (int t = 0; t < n_steps; ++t) {
    nnv_status st = dense_fwd_bwd(h, labels_dev + static_cast<int64_t>(t) * h->B, true, s);
    if (st != NNV_OK) return st;
    st = dense_adam(h, true, costs_dev ? costs_dev + t : nullptr,
                    t + 1 < n_steps ? x_dev + static_cast<int64_t>(t + 1) * step_stride : nullptr, row_stride, s);
    if (st != NNV_OK) return st;
  }
  return NNV_OK;
}

int32_t nnv_dense_train_gradients(nnv_dense_train_handle *h, const float *x_dev, int64_t row_stride, const int32_t *labels_dev,
                                  float *cost_host, float *const gW_host[5], float *const gb_host[5], float *gextra_host) {
  if (!h || !x_dev || !labels_dev || row_stride < 1) return NNV_ERR_INVALID_ARGUMENT;
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  hipLaunchKernelGGL(nnv::dense_stage_kernel, dim3(stage_blocks(h)), dim3(256), 0, nullptr, stage_params(h, x_dev, row_stride));
  nnv_status st = dense_fwd_bwd(h, labels_dev, false, nullptr);
  if (st != NNV_OK) return st;
  st = dense_adam(h, false, h->cost_scratch, nullptr, 0, nullptr);
  if (st != NNV_OK) return st;
  if (hipDeviceSynchronize() != hipSuccess) return NNV_ERR_HIP;
  std::vector<float> g(static_cast<size_t>(h->n_params));
  float c = 0.f;
  if (hipMemcpy(g.data(), h->gsum, g.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
  if (hipMemcpy(&c, h->cost_scratch, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
  if (cost_host) *cost_host = c;
  for (int l = 0; l < h->L; ++l) {
    const float *G = g.data() + h->off_w[l];
    if (gW_host && gW_host[l])
      for (int k = 0; k < h->K[l]; ++k)
        for (int n = 0; n < h->N[l]; ++n) gW_host[l][static_cast<size_t>(k) * h->N[l] + n] = G[static_cast<int64_t>(k) * h->Np[l] + n];
    if (gb_host && gb_host[l])
      for (int n = 0; n < h->N[l]; ++n) gb_host[l][n] = G[static_cast<int64_t>(h->K[l]) * h->Np[l] + n];
  }
  if (gextra_host)
    for (int64_t i = 0; i < h->n_extra; ++i) gextra_host[i] = g[static_cast<size_t>(h->n_matrix + i)];
  return NNV_OK;
}

int32_t nnv_dense_train_export(const nnv_dense_train_handle *h, float *const W_host[5], float *const b_host[5], float *extra_host,
                               double *cost_sum_host) {
  if (!h) return NNV_ERR_INVALID_ARGUMENT;
  if (hipDeviceSynchronize() != hipSuccess) return NNV_ERR_HIP;
  std::vector<float> w(static_cast<size_t>(h->n_params));
  if (hipMemcpy(w.data(), h->w, w.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
  for (int l = 0; l < h->L; ++l) {
    const float *W = w.data() + h->off_w[l];
    if (W_host && W_host[l])
      for (int k = 0; k < h->K[l]; ++k)
        for (int n = 0; n < h->N[l]; ++n) W_host[l][static_cast<size_t>(k) * h->N[l] + n] = W[static_cast<int64_t>(k) * h->Np[l] + n];
    if (b_host && b_host[l])
      for (int n = 0; n < h->N[l]; ++n) b_host[l][n] = W[static_cast<int64_t>(h->K[l]) * h->Np[l] + n];
  }
  if (extra_host)
    for (int64_t i = 0; i < h->n_extra; ++i) extra_host[i] = w[static_cast<size_t>(h->n_matrix + i)];
  if (cost_sum_host) {   // sum of the step costs since the last export, then reset
    const size_t n = static_cast<size_t>(h->n_row_tiles) + static_cast<size_t>(h->adam_blocks);   // ce_acc | reg_acc, adjacent
    std::vector<double> acc(n);
    if (hipMemcpy(acc.data(), h->ce_acc, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return NNV_ERR_HIP;
    double ce = 0.0, reg = 0.0;
    for (size_t i = 0; i < static_cast<size_t>(h->n_row_tiles); ++i) ce += acc[i];
    for (size_t i = static_cast<size_t>(h->n_row_tiles); i < n; ++i) reg += acc[i];
    *cost_sum_host = ce + static_cast<double>(h->beta_reg) * reg;
    if (hipMemset(h->ce_acc, 0, n * sizeof(double)) != hipSuccess) return NNV_ERR_HIP;
  }
  return NNV_OK;
}

}  // extern "C"
