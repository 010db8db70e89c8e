// Recurrent decoder forward (LSTM encoder -> LSTM decoder -> linear) for gfx950.
//
// Replaces the inference tensors of the reference's build_seq2seq_model
//   tf.contrib.rnn.LSTMCell + tf.nn.dynamic_rnn (encoder)   src/rnn_viterbi.py:289-297
//   the same for the decoder, initial_state = encoder state  src/rnn_viterbi.py:302-312
//   tf.contrib.layers.linear(decoder_outputs, 1)             src/rnn_viterbi.py:315-318
// with the TF-1.x LSTMCell arithmetic (gate order i, j, f, o; forget_bias 1.0 inside the
// sigmoid; c = sig(f) c + sig(i) tanh(j); m = sig(o) tanh(c)).  SURVEY.md section 8f row 4 /
// BASELINE configs[4].  The reference script itself cannot run: parity unpinned,
// the checker is oracle/rnn_oracle.py.
//
// Design (MI355X-first):
//  * One wave owns 32 sequences (batch rows) for the whole 2 x T recurrence; the row is on the
//    lane, the hidden units are in registers, so the recurrent state never leaves the register
//    file and there is no LDS, no barrier and no inter-wave traffic at all.
//  * Per time step the gate pre-activations are one transposed product on the fp32 matrix cores
//    (v_mfma_f32_32x32x2_f32, exact fp32): G^T[gates x 32 rows] = W^T [h_{t-1}; x_t; 1].
//    The two lane halves of a 32x32x2 step carry two different k values, so the H hidden units
//    are split over the halves: half h holds units h*S .. h*S+S-1 (S = ceil(H/2) "slots").
//    A lane half sees 16 of a tile's 32 output neurons; the four gates of slot k are placed at
//    the half's position 16*T + r = g*S + k, so i, j, f, o of one unit sit in the SAME lane in
//    fixed registers: the cell update is register-wise VALU work, and h_t (one register per
//    slot, both halves) is the next step's B operand as it stands.  4S <= 47 positions are
//    used of the 48 a half owns in 3 tiles: 3*(S+1) MFMAs per step, 80 of 96 outputs live
//    for H = 20 (a gate-per-tile layout would need 4 tiles and 52 MFMAs).
//  * x_t and the bias ride in ONE extra k-step: half 0 supplies x_t, half 1 the constant 1.
//  * The A operands (weights) of a whole LSTM cell are 3*(S+1) <= 36 registers per lane,
//    loaded once per cell: the time loop issues no weight loads.
//  * The output projection w.h_{t-1} + b is position 4S of half 0 (zero cost in the next step's
//    MFMAs); one extra chain after the last step emits the last logit.
//  * sigmoid / tanh scales (-log2 e, 2 log2 e) are folded into the packed weights.
//  * Inputs, logits, hard bits and labels are time-major [T][B]: a step's 32 rows are one
//    128-byte line.
#include <vector>
#include <new>
#include <string.h>

#include "nnv_internal.h"

struct nnv_seq2seq_handle {
  int32_t hidden;   // H
  int32_t slots;    // S (instantiated value >= ceil(H / 2))
  float *wpack_dev; // [2 cells][3 tiles][S + 1 k-steps][64 lanes]
  int device;
};

namespace nnv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Seq2SeqParams {
  const float *enc_in;  // [T_enc][B]
  const float *dec_in;  // [T_dec][B]
  int32_t t_enc, t_dec;
  int64_t batch;
  int64_t n_tiles;
  const float *wpack;
  float *logits;            // [T_dec][B], nullable
  uint8_t *bits;            // [n_bits][B] hard decisions logit > 0, nullable
  const uint8_t *info;      // [n_bits][B] 0/1 information bits, nullable
  int32_t n_bits;
  unsigned long long *counters;  // {bit_errors, frame_errors, total_bits, total_frames}
};

constexpr float kTwoLog2e = 2.8853900817779268f;

template <int S>
__global__ __launch_bounds__(256) void seq2seq_kernel(const Seq2SeqParams p) {
  constexpr int KS = S + 1;
  constexpr int MP = 4 * S;  // projection position (half 0)
  static_assert(MP < 48, "gates + projection must fit 3 tiles");
  const int lane = threadIdx.x & 63;
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (tile >= p.n_tiles) return;  // whole waves only; the kernel has no barrier
  const int half = lane >> 5;
  const int64_t row = tile * 32 + (lane & 31);
  const bool live = row < p.batch;
  const int64_t row_c = live ? row : p.batch - 1;

  float w[3][KS];
  auto load_cell = [&](int cell) {
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
      for (int s = 0; s < KS; ++s) w[T][s] = p.wpack[((cell * 3 + T) * KS + s) * 64 + lane];
  };
  float c[S], h[S];
#pragma unroll
  for (int k = 0; k < S; ++k) c[k] = h[k] = 0.f;

  // one LSTM step; returns the projection slot (w.h_prev + b, meaningful in half 0 of the decoder)
  auto step = [&](float xin) -> float {
    f32x16 acc[3];
#pragma unroll
    for (int T = 0; T < 3; ++T) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[T][0], h[0], zero, 0, 0, 0);
    }
#pragma unroll
    for (int s = 1; s < S; ++s)
#pragma unroll
      for (int T = 0; T < 3; ++T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[T][s], h[s], acc[T], 0, 0, 0);
#pragma unroll
    for (int T = 0; T < 3; ++T) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[T][S], xin, acc[T], 0, 0, 0);
    const float proj = acc[MP / 16][MP % 16];
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const float zi = acc[(0 * S + k) / 16][(0 * S + k) % 16];
      const float zj = acc[(1 * S + k) / 16][(1 * S + k) % 16];
      const float zf = acc[(2 * S + k) / 16][(2 * S + k) % 16];
      const float zo = acc[(3 * S + k) / 16][(3 * S + k) % 16];
      // pre-activations arrive scaled: sigmoid gates by -log2 e, the candidate by 2 log2 e
      const float si = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zi));
      const float sf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zf));
      const float so = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zo));
      const float rj = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zj));
      const float tj = __builtin_fmaf(-2.0f, rj, 1.0f);                      // tanh(j)
      const float cn = __builtin_fmaf(si, tj, sf * c[k]);
      const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn * kTwoLog2e));
      c[k] = cn;
      h[k] = so * __builtin_fmaf(-2.0f, rc, 1.0f);                           // sig(o) tanh(c)
    }
    return proj;
  };

  // ------------------------------ encoder ------------------------------------------------
  load_cell(0);
  {
    float xin = (half == 0 && p.t_enc > 0) ? p.enc_in[row_c] : 1.0f;
    for (int t = 0; t < p.t_enc; ++t) {
      float xnext = 1.0f;
      if (half == 0 && t + 1 < p.t_enc) xnext = p.enc_in[static_cast<int64_t>(t + 1) * p.batch + row_c];
      (void)step(xin);
      xin = xnext;
    }
  }
  // ------------------------------ decoder ------------------------------------------------
  load_cell(1);
  unsigned bit_err = 0;
  auto emit = [&](int t, float logit) {  // logit of decoder step t, valid in half 0
    if (half != 0 || !live) return;
    const int64_t o = static_cast<int64_t>(t) * p.batch + row;
    if (p.logits) p.logits[o] = logit;
    if (t < p.n_bits) {
      const unsigned bit = logit > 0.f ? 1u : 0u;
      if (p.bits) p.bits[o] = static_cast<uint8_t>(bit);
      if (p.info) bit_err += bit ^ static_cast<unsigned>(p.info[o] & 1);
    }
  };
  {
    float xin = (half == 0 && p.t_dec > 0) ? p.dec_in[row_c] : 1.0f;
    for (int t = 0; t < p.t_dec; ++t) {
      float xnext = 1.0f;
      if (half == 0 && t + 1 < p.t_dec) xnext = p.dec_in[static_cast<int64_t>(t + 1) * p.batch + row_c];
      const float proj = step(xin);   // = logit of step t - 1 (the step-0 value is the encoder's h: unused)
      if (t > 0) emit(t - 1, proj);
      xin = xnext;
    }
    if (p.t_dec > 0) {                // last logit: the projection chain alone
      constexpr int TP = MP / 16;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[TP][0], h[0], zero, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[TP][s], h[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[TP][S], half == 0 ? 0.f : 1.0f, acc, 0, 0, 0);
      emit(p.t_dec - 1, acc[MP % 16]);
    }
  }
  if (p.counters && p.info) {
    unsigned frame_err = bit_err != 0u ? 1u : 0u;
    unsigned rows = (half == 0 && live) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      bit_err += __shfl_xor(bit_err, off);
      frame_err += __shfl_xor(frame_err, off);
      rows += __shfl_xor(rows, off);
    }
    if (lane == 0) {
      const int nb = p.n_bits < p.t_dec ? p.n_bits : p.t_dec;
      if (bit_err) atomicAdd(p.counters + 0, static_cast<unsigned long long>(bit_err));
      if (frame_err) atomicAdd(p.counters + 1, static_cast<unsigned long long>(frame_err));
      atomicAdd(p.counters + 2, static_cast<unsigned long long>(rows) * nb);
      atomicAdd(p.counters + 3, static_cast<unsigned long long>(rows));
    }
  }
}

using Seq2SeqFn = void (*)(const Seq2SeqParams);
struct SeqEntry { int slots; Seq2SeqFn fn; };
const SeqEntry kSeqKernels[] = {
    {2, seq2seq_kernel<2>}, {4, seq2seq_kernel<4>}, {6, seq2seq_kernel<6>},
    {8, seq2seq_kernel<8>}, {10, seq2seq_kernel<10>}, {11, seq2seq_kernel<11>},
};

const SeqEntry *find_seq_kernel(int hidden) {
  const int need = (hidden + 1) / 2;
  for (const SeqEntry &e : kSeqKernels)
    if (e.slots >= need) return &e;
  return nullptr;
}

// [cell][tile][k-step][lane]; see the header comment for the placement rule.
void pack_seq2seq(const nnv_seq2seq_desc *d, int S, std::vector<float> *out) {
  const int H = d->hidden, KS = S + 1;
  out->assign(static_cast<size_t>(2) * 3 * KS * 64, 0.f);
  const double kLog2e = 1.4426950408889634;
  for (int cell = 0; cell < 2; ++cell) {
    const float *kernel = cell == 0 ? d->enc_kernel : d->dec_kernel;  // [1 + H][4H]
    const float *bias = cell == 0 ? d->enc_bias : d->dec_bias;        // [4H]
    for (int T = 0; T < 3; ++T)
      for (int s = 0; s < KS; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int pos = lane & 31, kh = lane >> 5;
          const int hh = (pos >> 2) & 1;                    // lane half that sees this output neuron
          const int reg = 4 * (pos >> 3) + (pos & 3);
          const int m = 16 * T + reg;                       // the half's position 0..47
          double v = 0.0;
          if (m < 4 * S) {
            const int g = m / S, k = m % S, u = hh * S + k;
            if (u < H) {
              const double scale = g == 1 ? 2.0 * kLog2e : -kLog2e;
              if (s < S) {
                const int uin = kh * S + s;
                if (uin < H) v = scale * kernel[static_cast<size_t>(1 + uin) * 4 * H + g * H + u];
              } else if (kh == 0) {
                v = scale * kernel[g * H + u];
              } else {
                v = scale * (static_cast<double>(bias[g * H + u]) + (g == 2 ? d->forget_bias : 0.0));
              }
            }
          } else if (m == 4 * S && hh == 0 && cell == 1) {
            if (s < S) {
              const int uin = kh * S + s;
              if (uin < H) v = d->proj_w[uin];
            } else if (kh == 1) {
              v = d->proj_b[0];
            }
          }
          (*out)[((static_cast<size_t>(cell) * 3 + T) * KS + s) * 64 + lane] = static_cast<float>(v);
        }
  }
}

}  // namespace
}  // namespace nnv

extern "C" {

int32_t nnv_seq2seq_prepack(const nnv_seq2seq_desc *d, nnv_seq2seq_handle **out) {
  if (!d || !out) return NNV_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (d->hidden < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (!d->enc_kernel || !d->enc_bias || !d->dec_kernel || !d->dec_bias || !d->proj_w || !d->proj_b)
    return NNV_ERR_INVALID_ARGUMENT;
  const nnv::SeqEntry *k = nnv::find_seq_kernel(d->hidden);
  if (!k) return NNV_ERR_UNSUPPORTED_SHAPE;  // hidden > 22
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return NNV_ERR_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return NNV_ERR_NO_DEVICE;
  std::vector<float> packed;
  nnv::pack_seq2seq(d, k->slots, &packed);
  nnv_seq2seq_handle *h = new (std::nothrow) nnv_seq2seq_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  h->hidden = d->hidden;
  h->slots = k->slots;
  h->device = dev;
  h->wpack_dev = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&h->wpack_dev), packed.size() * sizeof(float)) != hipSuccess) {
    delete h;
    return NNV_ERR_OUT_OF_MEMORY;
  }
  if (hipMemcpy(h->wpack_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->wpack_dev);
    delete h;
    return NNV_ERR_HIP;
  }
  *out = h;
  return NNV_OK;
}

int32_t nnv_seq2seq_free(nnv_seq2seq_handle *h) {
  if (!h) return NNV_OK;
  const hipError_t e = hipFree(h->wpack_dev);
  delete h;
  return e == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_seq2seq_forward_f32(const nnv_seq2seq_handle *h, const float *enc_in_dev,
                                const float *dec_in_dev, int32_t t_enc, int32_t t_dec, int64_t batch,
                                float *logits_dev, uint8_t *bits_dev, const uint8_t *info_dev,
                                int32_t n_bits, uint64_t *counters_dev, void *stream) {
  if (!h || batch < 0 || t_enc < 0 || t_dec < 0 || n_bits < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (batch == 0) return NNV_OK;
  if ((t_enc > 0 && !enc_in_dev) || (t_dec > 0 && !dec_in_dev)) return NNV_ERR_INVALID_ARGUMENT;
  if (counters_dev && !info_dev) return NNV_ERR_INVALID_ARGUMENT;
  if ((bits_dev || info_dev) && n_bits > t_dec) return NNV_ERR_INVALID_ARGUMENT;
  const nnv::SeqEntry *k = nnv::find_seq_kernel(h->hidden);
  if (!k || k->slots != h->slots) return NNV_ERR_UNSUPPORTED_SHAPE;
  nnv::Seq2SeqParams p;
  p.enc_in = enc_in_dev;
  p.dec_in = dec_in_dev;
  p.t_enc = t_enc;
  p.t_dec = t_dec;
  p.batch = batch;
  p.n_tiles = (batch + 31) / 32;
  p.wpack = h->wpack_dev;
  p.logits = logits_dev;
  p.bits = bits_dev;
  p.info = info_dev;
  p.n_bits = n_bits;
  p.counters = reinterpret_cast<unsigned long long *>(counters_dev);
  const int64_t blocks = (p.n_tiles + 3) / 4;
  if (blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k->fn, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // extern "C"
