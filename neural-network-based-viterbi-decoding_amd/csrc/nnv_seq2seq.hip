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
//  * PREC = kPrecF16x3 (opt-in): every operand split hi/lo into two f16 and each product block
//    computed as three v_mfma_f32_32x32x16_f16 (fp32 accumulate), as in the feed-forward kernel.
//    One f16 MFMA spans 16 k-slots (8 per lane half), so the [h; x; 1] vector of H <= 22 units
//    is 2 k-steps: 3 tiles x 2 steps x 3 products = 18 MFMAs per time step on the 16-bit matrix
//    pipe, which co-executes with the gate VALU work (the fp32 MFMAs do not).
#include <vector>
#include <new>
#include <string.h>

#include "nnv_internal.h"

struct nnv_seq2seq_handle {
  int32_t hidden;     // H
  int32_t slots;      // S (instantiated value >= ceil(H / 2))
  int32_t precision;  // kPrecF32 | kPrecF16x3
  float *wpack_dev;   // fp32: [2 cells][3 tiles][S + 1 k-steps][64 lanes]; f16x3: see pack_seq2seq_f16x3
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

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ f32x16 zero16() {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}

__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// (a, b) -> packed f16 hi pair and lo pair, lo = f16(v - hi) by one v_fma_mix_f32 per value
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 x = {a, b};
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(b));
  const f32x2 r = {r0, r1};
  hi = h;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// The A operands (weights) of one LSTM cell, register resident.
template <int S, int PREC>
struct CellWeights;
template <int S>
struct CellWeights<S, kPrecF32> {
  float w[3][S + 1];
  __device__ __forceinline__ void load(const float *wpack, int cell, int lane) {
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
      for (int s = 0; s < S + 1; ++s) w[T][s] = wpack[((cell * 3 + T) * (S + 1) + s) * 64 + lane];
  }
};
template <int S>
struct CellWeights<S, kPrecF16x3> {
  u32x4 hi[3][2], lo[3][2];
  __device__ __forceinline__ void load(const float *wpack, int cell, int lane) {
    const u32x4 *q = reinterpret_cast<const u32x4 *>(wpack);
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        hi[T][s] = q[(((cell * 3 + T) * 2 + s) * 2 + 0) * 64 + lane];
        lo[T][s] = q[(((cell * 3 + T) * 2 + s) * 2 + 1) * 64 + lane];
      }
  }
};

template <int S, int PREC>
__global__ __launch_bounds__(256) void seq2seq_kernel(const Seq2SeqParams p) {
  constexpr int MP = 4 * S;  // projection position (half 0)
  static_assert(MP < 48, "gates + projection must fit 3 tiles");
  static_assert(S + 1 <= 16, "[h; x] must fit the 16 k-slots a lane half has in two f16 MFMAs");
  const int lane = threadIdx.x & 63;
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (tile >= p.n_tiles) return;  // whole waves only; the kernel has no barrier
  const int half = lane >> 5;
  const int64_t row = tile * 32 + (lane & 31);
  const bool live = row < p.batch;
  const int64_t row_c = live ? row : p.batch - 1;

  CellWeights<S, PREC> W;
  float c[S], h[S];   // c: the cell state times 2 log2(e) (see step), h: the output m
#pragma unroll
  for (int k = 0; k < S; ++k) c[k] = h[k] = 0.f;

  // gate pre-activations of one step: acc[T] = W_T^T [h; xin] (xin: x_t in half 0, 1 in half 1);
  // n_tiles_needed < 3 computes only the projection's tile (the last logit)
  auto gates = [&](float xin, f32x16 (&acc)[3], bool proj_only) {
    if constexpr (PREC == kPrecF32) {
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        if (proj_only && T != MP / 16) continue;
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.w[T][0], h[0], zero16(), 0, 0, 0);
#pragma unroll
        for (int s = 1; s < S; ++s) acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.w[T][s], h[s], acc[T], 0, 0, 0);
        acc[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(W.w[T][S], xin, acc[T], 0, 0, 0);
      }
    } else {
      // this lane half's 16 k-slots: h[0..S-1], xin, zeros; two k-steps of 8, split hi/lo
      u32x4 bhi[2], blo[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 8 * s + 2 * q;
          if (n > S) {
            bhi[s][q] = 0u;
            blo[s][q] = 0u;
          } else {
            const float v0 = n < S ? h[n < S ? n : 0] : xin;
            const float v1 = n + 1 < S ? h[n + 1 < S ? n + 1 : 0] : (n + 1 == S ? xin : 0.f);
            unsigned ph, pl;
            split_pair(v0, v1, ph, pl);
            bhi[s][q] = ph;
            blo[s][q] = pl;
          }
        }
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        if (proj_only && T != MP / 16) continue;
        acc[T] = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (8 * s > S) continue;               // a k-step holding only zero slots
          acc[T] = mfma_h(W.lo[T][s], bhi[s], acc[T]);  // small terms first
          acc[T] = mfma_h(W.hi[T][s], blo[s], acc[T]);
          acc[T] = mfma_h(W.hi[T][s], bhi[s], acc[T]);
        }
      }
    }
  };

  // one LSTM step; returns the projection slot (w.h_prev + b, meaningful in half 0 of the decoder)
  auto step = [&](float xin) -> float {
    f32x16 acc[3];
    gates(xin, acc, false);
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
      // the cell state is carried as ct = 2 log2(e) c, the argument tanh(c)'s exp2 wants:
      // ct' = sig(f) ct + sig(i) (2 log2(e) tanh(j)), tanh(j) = 1 - 2 rj
      const float tj = __builtin_fmaf(-2.0f * kTwoLog2e, rj, kTwoLog2e);
      const float cn = __builtin_fmaf(si, tj, sf * c[k]);
      const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn));
      c[k] = cn;
      h[k] = so * __builtin_fmaf(-2.0f, rc, 1.0f);                           // sig(o) tanh(c)
    }
    return proj;
  };

  // ------------------------------ encoder ------------------------------------------------
  W.load(p.wpack, 0, lane);
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
  W.load(p.wpack, 1, lane);
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
    if (p.t_dec > 0) {                // last logit: the projection's tile alone
      f32x16 acc[3];
      gates(1.0f, acc, true);         // the x slot of the projection row is 0, its bias slot sees 1
      emit(p.t_dec - 1, acc[MP / 16][MP % 16]);
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
struct SeqEntry { int slots, prec; Seq2SeqFn fn; };
#define NNV_SEQ(S) {S, kPrecF32, seq2seq_kernel<S, kPrecF32>}, {S, kPrecF16x3, seq2seq_kernel<S, kPrecF16x3>}
const SeqEntry kSeqKernels[] = {NNV_SEQ(2), NNV_SEQ(4), NNV_SEQ(6), NNV_SEQ(8), NNV_SEQ(10), NNV_SEQ(11)};

const SeqEntry *find_seq_kernel(int hidden, int prec) {
  const int need = (hidden + 1) / 2;
  for (const SeqEntry &e : kSeqKernels)
    if (e.slots >= need && e.prec == prec) return &e;
  return nullptr;
}

// The packed value of (cell, tile T, output position `pos`, input k-slot n of lane half kh):
// n < S: hidden unit kh*S + n; n == S: x_t (kh = 0) or the bias (kh = 1); see the header comment.
double packed_weight(const nnv_seq2seq_desc *d, int S, int cell, int T, int pos, int kh, int n) {
  const int H = d->hidden;
  const double kLog2e = 1.4426950408889634;
  const float *kernel = cell == 0 ? d->enc_kernel : d->dec_kernel;  // [1 + H][4H]
  const float *bias = cell == 0 ? d->enc_bias : d->dec_bias;        // [4H]
  const int hh = (pos >> 2) & 1;                    // lane half that sees this output neuron
  const int reg = 4 * (pos >> 3) + (pos & 3);
  const int m = 16 * T + reg;                       // the half's position 0..47
  if (n > S) return 0.0;
  if (m < 4 * S) {
    const int g = m / S, k = m % S, u = hh * S + k;
    if (u >= H) return 0.0;
    const double scale = g == 1 ? 2.0 * kLog2e : -kLog2e;
    if (n < S) {
      const int uin = kh * S + n;
      return uin < H ? scale * kernel[static_cast<size_t>(1 + uin) * 4 * H + g * H + u] : 0.0;
    }
    if (kh == 0) return scale * kernel[g * H + u];
    return scale * (static_cast<double>(bias[g * H + u]) + (g == 2 ? d->forget_bias : 0.0));
  }
  if (m == 4 * S && hh == 0 && cell == 1) {
    if (n < S) {
      const int uin = kh * S + n;
      return uin < H ? d->proj_w[uin] : 0.0;
    }
    return kh == 1 ? d->proj_b[0] : 0.0;
  }
  return 0.0;
}

// fp32: [cell][tile][k-step s = slot n][lane]
void pack_seq2seq(const nnv_seq2seq_desc *d, int S, std::vector<float> *out) {
  const int KS = S + 1;
  out->assign(static_cast<size_t>(2) * 3 * KS * 64, 0.f);
  for (int cell = 0; cell < 2; ++cell)
    for (int T = 0; T < 3; ++T)
      for (int s = 0; s < KS; ++s)
        for (int lane = 0; lane < 64; ++lane)
          (*out)[((static_cast<size_t>(cell) * 3 + T) * KS + s) * 64 + lane] =
              static_cast<float>(packed_weight(d, S, cell, T, lane & 31, lane >> 5, s));
}

// f16x3: [cell][tile][k-step s < 2][hi | lo][lane][8 halves], slot n = 8 s + e.  false: a weight
// is outside the f16 range.
bool pack_seq2seq_f16x3(const nnv_seq2seq_desc *d, int S, std::vector<float> *out) {
  out->assign(static_cast<size_t>(2) * 3 * 2 * 2 * 64 * 4, 0.f);
  uint16_t *w = reinterpret_cast<uint16_t *>(out->data());
  bool in_range = true;
  for (int cell = 0; cell < 2; ++cell)
    for (int T = 0; T < 3; ++T)
      for (int s = 0; s < 2; ++s) {
        uint16_t *hi = w + ((((static_cast<size_t>(cell) * 3 + T) * 2 + s) * 2 + 0) * 64) * 8;
        uint16_t *lo = hi + 64 * 8;
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const double v = packed_weight(d, S, cell, T, lane & 31, lane >> 5, 8 * s + e);
            if (!(v > -65504.0 && v < 65504.0)) in_range = false;
            const uint16_t h = f32_to_f16(static_cast<float>(v));
            hi[lane * 8 + e] = h;
            lo[lane * 8 + e] = f32_to_f16(static_cast<float>(v - static_cast<double>(f16_to_f32(h))));
          }
      }
  return in_range;
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
  if (d->precision != NNV_PRECISION_F32 && d->precision != NNV_PRECISION_F16X3) return NNV_ERR_INVALID_ARGUMENT;
  const nnv::SeqEntry *k = nnv::find_seq_kernel(d->hidden, d->precision);
  if (!k) return NNV_ERR_UNSUPPORTED_SHAPE;  // hidden > 22
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return NNV_ERR_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return NNV_ERR_NO_DEVICE;
  std::vector<float> packed;
  if (d->precision == NNV_PRECISION_F16X3) {
    if (!nnv::pack_seq2seq_f16x3(d, k->slots, &packed)) return NNV_ERR_UNSUPPORTED_SHAPE;  // weight outside f16
  } else {
    nnv::pack_seq2seq(d, k->slots, &packed);
  }
  nnv_seq2seq_handle *h = new (std::nothrow) nnv_seq2seq_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  h->hidden = d->hidden;
  h->slots = k->slots;
  h->precision = d->precision;
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
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  if ((t_enc > 0 && !enc_in_dev) || (t_dec > 0 && !dec_in_dev)) return NNV_ERR_INVALID_ARGUMENT;
  if (counters_dev && !info_dev) return NNV_ERR_INVALID_ARGUMENT;
  if ((bits_dev || info_dev) && n_bits > t_dec) return NNV_ERR_INVALID_ARGUMENT;
  const nnv::SeqEntry *k = nnv::find_seq_kernel(h->hidden, h->precision);
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
