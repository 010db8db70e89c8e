// K9: the two halves of back-propagation through time for the reference's recurrent decoder (SURVEY.md section 8f rank 4,
// the training side) -- tf.contrib.rnn.LSTMCell unrolled by tf.nn.dynamic_rnn, encoder and decoder
// (src/rnn_viterbi.py:277-299), trained by AdamOptimizer on the mean squared error (:310-316).
//
//   z_t = [x_t, m_{t-1}] kernel + bias -> (i, j, f, o)           BasicLSTMCell's gate order, forget_bias on f
//   c_t = sigmoid(f + forget_bias) c_{t-1} + sigmoid(i) tanh(j);  m_t = sigmoid(o) tanh(c_t)
//
// PyTorch autograd unrolls 4 L = 400 cells into ~16,000 tiny kernels per training step (45 ms replayed from a HIP graph:
// the reference's default run takes 4.5 minutes; K9: 4.7 ms).  Here the RECURRENCE is one launch per LSTM and direction, every
// time step inside it; what is not recurrent stays a handful of PyTorch kernels around it:
//   before:  zx[t] = x_t kernel[0] + bias for all t at once (one broadcast; its gradient is autograd's)
//   K9 fwd:  one lane per batch row walks t = 0 .. T-1: z = zx[t] + m_{t-1} W_h (W_h = kernel[1:], 20 x 80, read through
//            the scalar cache: every lane needs the same weight, so it is an SGPR operand of the FMA), gates, cell;
//            the activated gates, c_t and m_t are kept for the way back.  The next step's zx is requested before this
//            step's 1,600 FMAs.
//   K9 bwd:  t = T-1 .. 0: dz_t from (dm_t, dc_t) and the kept gates, dm_{t-1} = dz_t W_h^T, dc_{t-1} = dc_t sigmoid(f).
//   after:   dW_h = sum_t m_{t-1}^T dz_t as ONE matrix product over all (t, row) pairs (PyTorch / rocBLAS).
// Everything is laid out [T][feature][B] (the batch on the lanes: coalesced).  Exact fp32 FMAs, no matrix cores: at 100
// rows x 20 units a time step is 1,600 FMAs per lane and the walk is latency-, not throughput-shaped; one lane per row
// keeps the cell update free of cross-lane traffic.  Checked against float64 autograd of the unrolled graph
// (tests/test_lstm_trainer.py).  The reference's recurrent decoder cannot learn by construction: this
// makes its flow finish in seconds, nothing more.
#include <math.h>

#include "nnv_internal.h"

namespace nnv {
namespace {

typedef float __attribute__((address_space(4))) cfloat;
using f32x2 = __attribute__((ext_vector_type(2))) float;   // constant address space: uniform loads become s_load

__device__ __forceinline__ float sigmoid_f(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)); }
__device__ __forceinline__ float tanh_f(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// Pointer arguments are separate and __restrict__ (not a parameter struct): the kept tensors are written while the
// inputs are read, and the compiler may then order the two freely.
struct LstmDims {
  int32_t T, B;
  float forget_bias;
};

// zx [T][4H][B] x_t kernel[0] + bias (gate order i, j, f, o); wh [H][4H] kernel[1:]; c0, m0 [H][B];
// gates [T][4H][B] sigmoid(i), tanh(j), sigmoid(f + forget_bias), sigmoid(o); c_all, m_all [T][H][B]
template <int H>
__global__ __launch_bounds__(64) void lstm_seq_fwd_kernel(const float *__restrict__ zx, const float *__restrict__ wh,
                                                          const float *__restrict__ c0, const float *__restrict__ m0,
                                                          float *__restrict__ gates, float *__restrict__ c_all,
                                                          float *__restrict__ m_all, const LstmDims p) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
  const int64_t B = p.B;
  float c[H], m[H], z[4 * H], zn[4 * H];
#pragma unroll
  for (int k = 0; k < H; ++k) {
    c[k] = c0[k * B + b];
    m[k] = m0[k * B + b];
  }
#pragma unroll
  for (int g = 0; g < 4 * H; ++g) z[g] = zx[g * B + b];
  for (int t = 0; t < p.T; ++t) {
    if (t + 1 < p.T) {
      const float *nx = zx + static_cast<int64_t>(t + 1) * 4 * H * B + b;
#pragma unroll
      for (int g = 0; g < 4 * H; ++g) zn[g] = nx[g * B];
    }
    // the weights: every lane needs the same one, so they go through the SCALAR cache straight into SGPR operands of the
    // FMAs (no vector register, no LDS port).  They never change: left alone, LLVM hoists all 1,600 loads out of the
    // time loop and spills them into vector lanes; a pointer that is opaque once per step keeps them streaming next to
    // their FMAs (constant address space: a uniform load from it is an s_load whatever else the kernel stores).
    unsigned long long wp = reinterpret_cast<unsigned long long>(wh);
    asm volatile("" : "+s"(wp));
    const cfloat *w = reinterpret_cast<const cfloat *>(wp);
    // two gate columns per instruction (v_pk_fma_f32: the weight pair is an aligned SGPR pair, m_k is broadcast): a lone
    // wave is bound by instruction issue, and a packed FMA issues in the time of a scalar one
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const f32x2 mk = {m[k], m[k]};
#pragma unroll
      for (int g = 0; g < 4 * H; g += 2) {
        const f32x2 wv = {w[k * 4 * H + g], w[k * 4 * H + g + 1]};
        const f32x2 zv = {z[g], z[g + 1]};
        const f32x2 r = __builtin_elementwise_fma(mk, wv, zv);
        z[g] = r[0];
        z[g + 1] = r[1];
      }
    }
    float *go = gates + static_cast<int64_t>(t) * 4 * H * B + b;
    float *co = c_all + static_cast<int64_t>(t) * H * B + b;
    float *mo = m_all + static_cast<int64_t>(t) * H * B + b;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const float si = sigmoid_f(z[k]), tj = tanh_f(z[H + k]), sf = sigmoid_f(z[2 * H + k] + p.forget_bias), so = sigmoid_f(z[3 * H + k]);
      c[k] = __builtin_fmaf(sf, c[k], si * tj);
      m[k] = so * tanh_f(c[k]);
      go[k * B] = si;
      go[(H + k) * B] = tj;
      go[(2 * H + k) * B] = sf;
      go[(3 * H + k) * B] = so;
      co[k * B] = c[k];
      mo[k * B] = m[k];
    }
#pragma unroll
    for (int g = 0; g < 4 * H; ++g) z[g] = zn[g];
  }
}

// dm_all [T][H][B] gradient w.r.t. every m_t as an OUTPUT (zeros where unused); dcT, dmT [H][B] w.r.t. the final state;
// dz [T][4H][B] out: w.r.t. the gate pre-activations; dc0, dm0 [H][B] out
template <int H>
__global__ __launch_bounds__(64) void lstm_seq_bwd_kernel(const float *__restrict__ dm_all, const float *__restrict__ dcT,
                                                          const float *__restrict__ dmT, const float *__restrict__ gates,
                                                          const float *__restrict__ c_all, const float *__restrict__ c0,
                                                          const float *__restrict__ wh, float *__restrict__ dz,
                                                          float *__restrict__ dc0, float *__restrict__ dm0, const LstmDims p) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= p.B) return;
  const int64_t B = p.B;
  float dc[H], dmc[H], dzr[4 * H];
#pragma unroll
  for (int k = 0; k < H; ++k) {
    dc[k] = dcT[k * B + b];
    dmc[k] = dmT[k * B + b];
  }
  for (int t = p.T - 1; t >= 0; --t) {
    const float *go = gates + static_cast<int64_t>(t) * 4 * H * B + b;
    const float *ct = c_all + static_cast<int64_t>(t) * H * B + b;
    const float *cp = t > 0 ? c_all + static_cast<int64_t>(t - 1) * H * B + b : c0 + b;
    const float *dmo = dm_all + static_cast<int64_t>(t) * H * B + b;
    float *dzo = dz + static_cast<int64_t>(t) * 4 * H * B + b;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const float si = go[k * B], tj = go[(H + k) * B], sf = go[(2 * H + k) * B], so = go[(3 * H + k) * B];
      const float tc = tanh_f(ct[k * B]);
      const float dm = dmo[k * B] + dmc[k];
      const float dck = __builtin_fmaf(dm * so, 1.0f - tc * tc, dc[k]);
      dzr[k] = dck * tj * si * (1.0f - si);
      dzr[H + k] = dck * si * (1.0f - tj * tj);
      dzr[2 * H + k] = dck * cp[k * B] * sf * (1.0f - sf);
      dzr[3 * H + k] = dm * tc * so * (1.0f - so);
      dc[k] = dck * sf;
    }
#pragma unroll
    for (int g = 0; g < 4 * H; ++g) dzo[g * B] = dzr[g];
    unsigned long long wp = reinterpret_cast<unsigned long long>(wh);   // opaque per step: see the forward kernel
    asm volatile("" : "+s"(wp));
    const cfloat *w = reinterpret_cast<const cfloat *>(wp);
#pragma unroll
    for (int k = 0; k < H; ++k) {
      f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};           // packed FMAs, two chains of pairs per output
#pragma unroll
      for (int g = 0; g < 4 * H; g += 4) {
        const f32x2 d0 = {dzr[g], dzr[g + 1]}, d1 = {dzr[g + 2], dzr[g + 3]};
        const f32x2 w0 = {w[k * 4 * H + g], w[k * 4 * H + g + 1]}, w1 = {w[k * 4 * H + g + 2], w[k * 4 * H + g + 3]};
        sa = __builtin_elementwise_fma(d0, w0, sa);
        sb = __builtin_elementwise_fma(d1, w1, sb);
      }
      dmc[k] = (sa[0] + sa[1]) + (sb[0] + sb[1]);
    }
  }
#pragma unroll
  for (int k = 0; k < H; ++k) {
    dc0[k * B + b] = dc[k];
    dm0[k * B + b] = dmc[k];
  }
}

}  // namespace
}  // namespace nnv

extern "C" {

int32_t nnv_lstm_seq_forward_f32(const float *zx_dev, const float *wh_dev, const float *c0_dev, const float *m0_dev,
                                 int32_t n_steps, int32_t batch, int32_t hidden, float forget_bias, float *gates_dev,
                                 float *c_all_dev, float *m_all_dev, void *stream) {
  if (!zx_dev || !wh_dev || !c0_dev || !m0_dev || !gates_dev || !c_all_dev || !m_all_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (n_steps < 1 || batch < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (hidden != 20) return NNV_ERR_UNSUPPORTED_SHAPE;   // the reference's encoder_hidden_units (src/rnn_viterbi.py:36)
  const nnv::LstmDims p{n_steps, batch, forget_bias};
  hipLaunchKernelGGL(nnv::lstm_seq_fwd_kernel<20>, dim3((batch + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), zx_dev, wh_dev,
                     c0_dev, m0_dev, gates_dev, c_all_dev, m_all_dev, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_lstm_seq_backward_f32(const float *dm_all_dev, const float *dcT_dev, const float *dmT_dev, const float *gates_dev,
                                  const float *c_all_dev, const float *c0_dev, const float *wh_dev, int32_t n_steps,
                                  int32_t batch, int32_t hidden, float *dz_dev, float *dc0_dev, float *dm0_dev, void *stream) {
  if (!dm_all_dev || !dcT_dev || !dmT_dev || !gates_dev || !c_all_dev || !c0_dev || !wh_dev || !dz_dev || !dc0_dev || !dm0_dev)
    return NNV_ERR_INVALID_ARGUMENT;
  if (n_steps < 1 || batch < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (hidden != 20) return NNV_ERR_UNSUPPORTED_SHAPE;
  const nnv::LstmDims p{n_steps, batch, 0.f};
  hipLaunchKernelGGL(nnv::lstm_seq_bwd_kernel<20>, dim3((batch + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), dm_all_dev,
                     dcT_dev, dmT_dev, gates_dev, c_all_dev, c0_dev, wh_dev, dz_dev, dc0_dev, dm0_dev, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // extern "C"
