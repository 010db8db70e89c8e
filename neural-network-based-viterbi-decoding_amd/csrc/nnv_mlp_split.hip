// K1w: the fused decode of K1 (nnv_mlp_decode.hip) for SHORT inputs -- the same packed network, the same arithmetic,
// the same decision rule, laid out for LATENCY instead of throughput.
//
// K1 is a persistent kernel: a workgroup stages the whole packed network into LDS (up to 149 KB) and each of its waves
// walks 32-row tiles through all four layers on its own -- the right shape for 80 M rows, the wrong one for 2,000
// (BASELINE configs[0]): there a tile is ONE wave's serial chain of 320-712 MFMAs behind a 100 KB staging copy, 14-19 us
// however few rows there are.  K1w gives every 32-row tile a workgroup of four waves (one per SIMD) and deals each
// layer's 32-wide output tiles over them, as the training kernel K7 does (nnv_train.hip): layer 1 tile w; layer 2 on
// waves 0-1; layer 3's single tile on every wave (cheaper than a hand-over); the class tiles w, w + 4, ... with a
// per-wave running first maximum; tiles are handed to the other waves through LDS in the REGISTER layout (the lane that
// wrote a value reads it back: 16-byte accesses, conflict-free).  Weights are read straight from the packed image in
// global memory (L2), a phase's fragments requested before the previous phase's MFMAs; nothing is staged.
//
// The decision is K1's, value for value: first maximum over the class pre-activations clamped to +-kTanhSat (lower
// class index wins ties), and for a row whose winner reaches the flat top of fp32 tanh (>= kFlatBand) the first maximum
// over the tanh'd values instead (src/fnn_viterbi.py:606-610 is an argmax over the TANH outputs).  The per-wave winners
// are merged by (value, then lower index), which is the first maximum in class order whatever the dealing.  Same
// outputs as K1 on the same rows, bit for bit (tests/test_gpu_split.py).
//
// fp32, the first one to three of the reference's hidden widths (128, 64, 32: src/fnn_viterbi.py:35-37) exactly as K1
// takes them, class layers of any K1 size; classes,
// int64 classes, bit matrix and the four counters out; no per-SNR bins, no debug activations (those calls stay on K1).
#include "nnv_internal.h"

namespace nnv {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// kept identical to nnv_mlp_decode.hip (the decision rule is shared; tests compare the two kernels bit for bit)
constexpr float kTanhSatW = 9.011f;
constexpr float kFlatBandW = 7.9f;
constexpr int kXT = 64 * 16;   // floats of one exchange tile

__device__ __forceinline__ float tanh_w(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}
__device__ __forceinline__ f32x16 bias_acc(const float *bias_tile, int half) {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_tile + 8 * q + 4 * half);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * q + e] = bv[e];
  }
  return acc;
}
// hidden activation in the form the packed network expects: r = 1 / (1 + 2^zeta) (tanh's affine map is folded into the
// next layer's weights at prepack, nnv_capi.hip pack_network)
__device__ __forceinline__ void act_r(f32x16 &acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[r]));
}
template <int N>
__device__ __forceinline__ void load_groups(const float *frag, int lane, f32x4 *a) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = *reinterpret_cast<const f32x4 *>(frag + (i * 64 + lane) * 4);
  __builtin_amdgcn_sched_barrier(0);   // keep the requests here (the scheduler would sink each to its four MFMAs)
}
// acc += W-tile (4 KT fragment groups) * in[0 .. KT)
template <int KT>
__device__ __forceinline__ f32x16 chain(const f32x16 (&in)[KT], const f32x4 *a, f32x16 acc) {
#pragma unroll
  for (int i = 0; i < KT * 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], in[i / 4][4 * (i % 4) + e], acc, 0, 0, 0);
  return acc;
}
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

constexpr int kA = 4, kB = 2, kC = 1;   // 32-wide tiles of the hidden layers 128, 64, 32

// NHID hidden layers: the first NHID of the reference's widths (128, 64, 32), as in K1
template <int G1, int NHID>
__global__ __launch_bounds__(256) void mlp_decode_split_kernel(const DecodeParams p) {
  __shared__ __attribute__((aligned(16))) float xa[kA * kXT];   // h1, four tiles
  __shared__ __attribute__((aligned(16))) float xb[kB * kXT];   // h2, two tiles
  __shared__ float win_v[4][32];                                 // the waves' winners per row: value, class
  __shared__ int win_i[4][32];
  constexpr int G = kFragGroupFloats;
  constexpr int W1F = kA * G1 * G, W2F = NHID >= 2 ? kB * kA * 4 * G : 0, W3F = NHID >= 3 ? kC * kB * 4 * G : 0, HIDF = W1F + W2F + W3F;
  constexpr int KTO = NHID == 1 ? kA : (NHID == 2 ? kB : kC);   // 32-wide k-tiles of the class layer's input
  constexpr int GPT = 4 * KTO;                                   // fragment groups per class tile
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kRowsPerTile + j;
  const float *const g_out = p.wpack + HIDF;
  const float *const g_bias = g_out + static_cast<int64_t>(p.mt_out) * GPT * G;   // b1 (128) | b2 (64) | b3 (32) | class (32 mt_out)
  const float *const g_bo = g_bias + 32 * (kA + (NHID >= 2 ? kB : 0) + (NHID >= 3 ? kC : 0));

  // ---------------- P0: layer 1, tile w ----------------
  float xr[4 * G1];
  {
    const int valid = p.bp < p.n_in - half * p.bp ? p.bp : p.n_in - half * p.bp;   // the lane's own elements of its half row
    const float *px = p.x + row * p.n_in + half * p.bp;
#pragma unroll
    for (int s = 0; s < 4 * G1; ++s) xr[s] = (row < p.n_rows && s < valid) ? px[s] : 0.f;
  }
  f32x4 f1[G1];
  load_groups<G1>(p.wpack + static_cast<int64_t>(w) * G1 * G, lane, f1);
  f32x4 f2[kA * 4] = {};
  if (NHID >= 2 && w < kB) load_groups<kA * 4>(p.wpack + W1F + static_cast<int64_t>(w) * kA * 4 * G, lane, f2);
  {
    f32x16 acc = bias_acc(g_bias + 32 * w, half);
#pragma unroll
    for (int g = 0; g < G1; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f1[g][e], xr[4 * g + e], acc, 0, 0, 0);
    act_r(acc);
    xch_put(xa, w, acc, lane);
  }
  f32x4 f3[kB * 4] = {};
  if (NHID >= 3) load_groups<kB * 4>(p.wpack + W1F + W2F, lane, f3);
  __syncthreads();   // 1: h1

  // ---------------- P1: layer 2, tile w (waves 0, 1) ----------------
  if constexpr (NHID >= 2) {
    if (w < kB) {
      f32x16 h1[kA];
#pragma unroll
      for (int t = 0; t < kA; ++t) h1[t] = xch_get(xa, t, lane);
      f32x16 acc = chain<kA>(h1, f2, bias_acc(g_bias + 32 * kA + 32 * w, half));
      act_r(acc);
      xch_put(xb, w, acc, lane);
    }
    __syncthreads();   // 2: h2
  }

  // ---------------- P2: layer 3 (every wave), then this wave's class tiles ----------------
  f32x16 hl[KTO];      // the class layer's input: h1 (one hidden layer), h2 (two) or h3 (three)
  if constexpr (NHID == 1) {
#pragma unroll
    for (int t = 0; t < kA; ++t) hl[t] = xch_get(xa, t, lane);
  } else if constexpr (NHID == 2) {
#pragma unroll
    for (int t = 0; t < kB; ++t) hl[t] = xch_get(xb, t, lane);
  } else {
    f32x16 h2[kB];
#pragma unroll
    for (int t = 0; t < kB; ++t) h2[t] = xch_get(xb, t, lane);
    hl[0] = chain<kB>(h2, f3, bias_acc(g_bias + 32 * (kA + kB), half));
    act_r(hl[0]);
  }
  // running first maximum over the wave's tiles (ascending), on the clamped pre-activations: K1's rule
  float best = -__builtin_inff();
  int btile = 0;
  f32x16 keep;
#pragma unroll
  for (int r = 0; r < 16; ++r) keep[r] = 0.f;
  f32x4 fc[GPT] = {}, fn[GPT] = {};
  f32x16 bc = keep, bn = keep;
  if (w < p.mt_out) {
    load_groups<GPT>(g_out + static_cast<int64_t>(w) * GPT * G, lane, fc);
    bc = bias_acc(g_bo + 32 * w, half);
  }
  for (int mt = w; mt < p.mt_out; mt += 4) {
    if (mt + 4 < p.mt_out) {   // the next tile's fragments and bias, requested before this tile's MFMAs
      load_groups<GPT>(g_out + static_cast<int64_t>(mt + 4) * GPT * G, lane, fn);
      bn = bias_acc(g_bo + 32 * (mt + 4), half);
    }
    f32x16 acc = chain<KTO>(hl, fc, bc);
    if ((mt + 1) * 32 > p.n_classes) {   // last tile of a class count that is not a multiple of 32
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = 32 * mt + 4 * half + 8 * (r >> 2) + (r & 3);
        acc[r] = ci < p.n_classes ? acc[r] : -__builtin_inff();
      }
    }
    float m = __builtin_fmaxf(acc[0], acc[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, acc[r]), acc[r + 1]);
    m = __builtin_amdgcn_fmed3f(m, -kTanhSatW, kTanhSatW);
    const bool c = !(m <= best);       // as K1: a NaN maximum always takes the slot
    best = c ? m : best;
    btile = c ? mt : btile;
#pragma unroll
    for (int r = 0; r < 16; ++r) keep[r] = c ? acc[r] : keep[r];
#pragma unroll
    for (int q = 0; q < GPT; ++q) fc[q] = fn[q];
    bc = bn;
  }
  int bidx;
  {
    unsigned t = 16u;
#pragma unroll
    for (int r = 15; r >= 0; --r) t = keep[r] >= best ? static_cast<unsigned>(r) : t;
    t = t == 16u ? 0u : t;
    bidx = 32 * btile + 4 * half + 8 * static_cast<int>(t >> 2) + static_cast<int>(t & 3u);
  }
  auto merge = [](float &v, int &i, float ov, int oi) {   // the better of two (value, class): higher value, then lower class
    const bool take = (ov > v) || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
  };
  merge(best, bidx, __shfl_xor(best, 32), __shfl_xor(bidx, 32));
  if (w >= p.mt_out) best = -__builtin_inff();          // a wave without class tiles takes no part
  if (half == 0) {
    win_v[w][j] = best;
    win_i[w][j] = bidx;
  }
  __syncthreads();   // 3: the waves' winners
  float gbest = win_v[0][j];
  int gidx = win_i[0][j];
#pragma unroll
  for (int v = 1; v < 4; ++v) merge(gbest, gidx, win_v[v][j], win_i[v][j]);

  // ---------------- flat top of fp32 tanh: decide as the reference does (K1's cold path) ----------------
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(gbest >= kFlatBandW) != 0, 0)) {   // the same in every wave of the tile
    float tb = -__builtin_inff();
    int ti = 0;
    for (int mt = w; mt < p.mt_out; mt += 4) {
      load_groups<GPT>(g_out + static_cast<int64_t>(mt) * GPT * G, lane, fc);
      const f32x16 acc = chain<KTO>(hl, fc, bias_acc(g_bo + 32 * mt, half));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = 32 * mt + 4 * half + 8 * (r >> 2) + (r & 3);
        const float t = tanh_w(acc[r]);
        const bool c = ci < p.n_classes && t > tb;
        tb = c ? t : tb;
        ti = c ? ci : ti;
      }
    }
    merge(tb, ti, __shfl_xor(tb, 32), __shfl_xor(ti, 32));
    __syncthreads();   // everyone has read the first round's winners
    if (half == 0) {
      win_v[w][j] = tb;
      win_i[w][j] = ti;
    }
    __syncthreads();
    float fb = win_v[0][j];
    int fi = win_i[0][j];
#pragma unroll
    for (int v = 1; v < 4; ++v) merge(fb, fi, win_v[v][j], win_i[v][j]);
    const bool flat = gbest >= kFlatBandW;
    gidx = flat ? fi : gidx;
  }

  // ---------------- emit (wave 0, one lane per row) ----------------
  if (w != 0) return;
  unsigned bit_err = 0, frame_err = 0;
  if (half == 0 && row < p.n_rows) {
    if (p.cls32) p.cls32[row] = gidx;
    if (p.cls64) p.cls64[row] = gidx;
    if (p.bits) {
      uint8_t *pb = p.bits + row * p.tile_bits;
      for (int t = 0; t < p.tile_bits; ++t) pb[t] = static_cast<uint8_t>((gidx >> (p.tile_bits - 1 - t)) & 1);
    }
    if (p.labels) {
      const unsigned diff = static_cast<unsigned>(gidx) ^ static_cast<unsigned>(p.labels[row]);
      bit_err = __builtin_popcount(diff);
      frame_err = diff != 0u;
    }
  }
  if (p.counters) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      bit_err += __shfl_xor(bit_err, off);
      frame_err += __shfl_xor(frame_err, off);
    }
    if (lane == 0 && (bit_err | frame_err)) {
      atomicAdd(p.counters + 0, static_cast<unsigned long long>(bit_err));
      atomicAdd(p.counters + 1, static_cast<unsigned long long>(frame_err));
    }
    if (blockIdx.x == 0 && lane == 0) {
      atomicAdd(p.counters + 2, static_cast<unsigned long long>(p.n_rows) * p.tile_bits);
      atomicAdd(p.counters + 3, static_cast<unsigned long long>(p.n_rows));
    }
  }
}

using SplitFn = void (*)(const DecodeParams);
template <int NHID>
SplitFn split_fn_g(int g1) {
  switch (g1) {
    case 1: return mlp_decode_split_kernel<1, NHID>;
    case 2: return mlp_decode_split_kernel<2, NHID>;
    case 3: return mlp_decode_split_kernel<3, NHID>;
    case 4: return mlp_decode_split_kernel<4, NHID>;
    default: return nullptr;   // rows longer than 32 soft bits do not reach K1 at all
  }
}
SplitFn split_fn(int g1, int nhid) {
  return nhid == 1 ? split_fn_g<1>(g1) : (nhid == 2 ? split_fn_g<2>(g1) : (nhid == 3 ? split_fn_g<3>(g1) : nullptr));
}

}  // namespace

bool split_supported(const PackedShape &s) {
  if (s.precision != kPrecF32 || s.nhid < 1 || s.nhid > 3 || s.mt[0] != kA) return false;
  if ((s.nhid >= 2 && s.mt[1] != kB) || (s.nhid >= 3 && s.mt[2] != kC)) return false;
  return split_fn(s.g1, s.nhid) != nullptr;
}

nnv_status split_launch(const PackedShape &s, const DecodeParams &p, hipStream_t stream) {
  const SplitFn fn = split_supported(s) ? split_fn(s.g1, s.nhid) : nullptr;
  if (!fn || p.act || p.n_bins > 1) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (p.n_tiles > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(p.n_tiles)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
