// f16x3 variant of the layer-at-a-time path (nnv_dense.hip): every operand is a pair of f16
// planes, hi = f16(v) and lo = f16(v - hi) (22 significant bits together), and each product block
// is three v_mfma_f32_32x32x16_f16 (lo*hi + hi*lo + hi*hi) with fp32 accumulation: the same
// "3 x FP16" emulation of an fp32 GEMM as the PREC = f16x3 mode of nnv_mlp_decode.hip, ~2^-21
// relative error per product.  The 16-bit matrix pipe is 16x the fp32 MFMA rate, so the layer is
// no longer MFMA-bound at fp32 speed.
//
// Activations travel between layers ALREADY SPLIT: C_hi / C_lo planes, [rows][Mpad] f16 each (the
// same bytes per element as one fp32), written by the producer's epilogue where the values sit
// in registers anyway; the consumer stages them with plain 16-byte copies (no conversion in its
// main loop).  Only layer 1, which gathers fp32 rows from the received stream, splits while
// staging.  LDS image per row and k-tile: [hi 32 x f16][lo 32 x f16][16 B pad] = the 144-byte row
// stride of the fp32 kernel, so every ds_read_b128 / ds_write is conflict free.
// Operand maps (v_mfma_f32_32x32x16_f16): lane l holds A[row l&31][k = 8(l>>5) + e] and
// B[k = 8(l>>5) + e][col l&31], e = 0..7: one b128 read per operand per k-step.
#include "nnv_internal.h"

namespace nnv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

// LDS row per k-tile of BK: [hi BK x f16][lo BK x f16][16 B pad] (BK 32: 144 B, BK 64: 272 B)

__device__ __forceinline__ float tanh_f32x(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  return __builtin_fmaf(-2.0f, r, 1.0f);
}

__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 x = {a, b};
  const f16x2 h = __builtin_convertvector(x, f16x2);
  const f32x2 r = x - __builtin_convertvector(h, f32x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

__device__ __forceinline__ f32x16 mfma16x(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                0, 0, 0);
}

// HEAD (the bit-wise decoder BANK in f16x3, round 3): the output columns are P networks' hidden layers, 32 per
// position, and the epilogue finishes each network in place -- the same epilogue as dense_tanh_kernel<.., HEAD> in
// nnv_dense.hip, fed from the fp32 accumulators (the hidden values are NOT rounded to f16 on the way to the head).
template <bool GATHER, int BK, bool HEAD = false>
__global__ __launch_bounds__(256) void dense_f16x3_kernel(const DenseF16Params p) {
  constexpr int kPlane = BK * 2;            // bytes of one plane segment per row
  constexpr int kRowBytes = 2 * kPlane + 16;
  constexpr int kChunks = kPlane / 16;      // 16-byte chunks per plane segment (4 or 8)
  constexpr int kRowsPerPass = 256 / kChunks;
  constexpr int kPasses = 128 / kRowsPerPass;
  constexpr int kGPasses = BK / 16;         // gather: 8-float row pieces per thread
  constexpr int kGRows = 2048 / BK;         // rows covered per pass (BK / 8 threads per row)
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[(kDenseBM + kDenseBN) * kRowBytes];
  unsigned char *const lds_a = lds_all;
  unsigned char *const lds_w = lds_all + kDenseBM * kRowBytes;
  unsigned char *const stage = lds_all;   // epilogue: output rows staged over the (dead) operand tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t blk = blockIdx.x;
  const int xcd = static_cast<int>(blk & 7);  // same XCD-aware tile map as the fp32 kernel
  const int64_t slot = blk >> 3;
  int col_blk;
  int64_t row_blk;
  if (HEAD && p.col_group > 0) {  // column blocks in L2-sized groups, row blocks fastest inside a group (nnv_dense.hip)
    const int64_t n_r8 = (p.nblk_rows + 7) / 8;
    const int64_t per_group = n_r8 * p.col_group;
    const int groups = (p.nblk_m + p.col_group - 1) / p.col_group;
    int64_t g = slot / per_group;
    if (g > groups - 1) g = groups - 1;
    const int gsz = g == groups - 1 ? p.nblk_m - static_cast<int>(g) * p.col_group : p.col_group;
    const int64_t rem = slot - g * per_group;
    col_blk = static_cast<int>(g) * p.col_group + static_cast<int>(rem % gsz);
    row_blk = (rem / gsz) * 8 + xcd;
  } else {
    col_blk = static_cast<int>(slot % p.nblk_m);
    row_blk = (slot / p.nblk_m) * 8 + xcd;
  }
  if (row_blk >= p.nblk_rows) return;
  const int64_t row0 = row_blk * kDenseBM;
  const int col0 = col_blk * kDenseBN;

  // staging: per operand 128 rows x (64 B hi + 64 B lo) = 1024 x 16 B; 4 per thread
  const int st_row = tid / kChunks;     // + kRowsPerPass per pass
  const int st_chunk = tid % kChunks;   // 16-byte chunk (8 halves) of the plane segment
  u32x4 ra[2 * kPasses], rw[2 * kPasses];  // [0, kPasses): hi plane ; [kPasses, 2 kPasses): lo plane
  // gather mode (layer 1): fp32 rows, split while staging
  // each thread owns 8 consecutive k of one row: its hi and lo halves leave as ONE 16-byte LDS
  // write each (8-byte writes at the 144-byte row stride collided: 10 % of the layer's cycles)
  const int g_row = tid / (BK / 8), g_kq = (tid % (BK / 8)) * 8;
  f32x4 ga[2 * kGPasses];

  auto load_tiles = [&](int kt) {
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      const int64_t wrow = col0 + st_row + kRowsPerPass * ps;
      const int64_t woff = wrow * p.Kpad + kt * BK + st_chunk * 8;  // in halves
      rw[ps] = *reinterpret_cast<const u32x4 *>(p.W_hi + woff);
      rw[kPasses + ps] = *reinterpret_cast<const u32x4 *>(p.W_lo + woff);
      if (!GATHER) {
        int64_t r = row0 + st_row + kRowsPerPass * ps;
        r = r < p.n_rows ? r : p.n_rows - 1;
        const int64_t aoff = r * p.lda + kt * BK + st_chunk * 8;
        ra[ps] = *reinterpret_cast<const u32x4 *>(p.A_hi + aoff);
        ra[kPasses + ps] = *reinterpret_cast<const u32x4 *>(p.A_lo + aoff);
      }
    }
    if (GATHER) {
      const int k = kt * BK + g_kq;
#pragma unroll
      for (int ps = 0; ps < kGPasses; ++ps) {
        int64_t r = row0 + g_row + kGRows * ps;
        r = r < p.n_rows ? r : p.n_rows - 1;
        const float *src = p.A_f32 + r * p.a_row_stride + k;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // rows are only 8-byte aligned and may end inside the tile
          const int kk = k + 4 * h;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (kk + 1 < p.K) {
            const f32x2 t = *reinterpret_cast<const f32x2 *>(src + 4 * h);
            v[0] = t[0];
            v[1] = t[1];
          } else if (kk < p.K) {
            v[0] = src[4 * h];
          }
          if (kk + 3 < p.K) {
            const f32x2 t = *reinterpret_cast<const f32x2 *>(src + 4 * h + 2);
            v[2] = t[0];
            v[3] = t[1];
          } else if (kk + 2 < p.K) {
            v[2] = src[4 * h + 2];
          }
          ga[2 * ps + h] = v;
        }
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      unsigned char *w = lds_w + (st_row + kRowsPerPass * ps) * kRowBytes + st_chunk * 16;
      *reinterpret_cast<u32x4 *>(w) = rw[ps];
      *reinterpret_cast<u32x4 *>(w + kPlane) = rw[kPasses + ps];
      if (!GATHER) {
        unsigned char *a = lds_a + (st_row + kRowsPerPass * ps) * kRowBytes + st_chunk * 16;
        *reinterpret_cast<u32x4 *>(a) = ra[ps];
        *reinterpret_cast<u32x4 *>(a + kPlane) = ra[kPasses + ps];
      }
    }
    if (GATHER) {
#pragma unroll
      for (int ps = 0; ps < kGPasses; ++ps) {
        u32x4 hi, lo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned h0, l0, h1, l1;
          split2(ga[2 * ps + h][0], ga[2 * ps + h][1], h0, l0);
          split2(ga[2 * ps + h][2], ga[2 * ps + h][3], h1, l1);
          hi[2 * h] = h0;
          hi[2 * h + 1] = h1;
          lo[2 * h] = l0;
          lo[2 * h + 1] = l1;
        }
        unsigned char *a = lds_a + (g_row + kGRows * ps) * kRowBytes + g_kq * 2;
        *reinterpret_cast<u32x4 *>(a) = hi;
        *reinterpret_cast<u32x4 *>(a + kPlane) = lo;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
  const int nk = p.Kpad / BK;
  load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned char *pa = lds_a + (wr + 32 * t + li) * kRowBytes + 32 * s + 16 * lh;
        const unsigned char *pw = lds_w + (wc + 32 * t + li) * kRowBytes + 32 * s + 16 * lh;
        ah[t] = *reinterpret_cast<const u32x4 *>(pa);
        al[t] = *reinterpret_cast<const u32x4 *>(pa + kPlane);
        bh[t] = *reinterpret_cast<const u32x4 *>(pw);
        bl[t] = *reinterpret_cast<const u32x4 *>(pw + kPlane);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = mfma16x(al[a], bh[b], acc[a][b]);
          acc[a][b] = mfma16x(ah[a], bl[b], acc[a][b]);
          acc[a][b] = mfma16x(ah[a], bh[b], acc[a][b]);
        }
    }
  }

  // epilogue: + bias, activation, split; the block's 128 x 128 tile goes through LDS one plane at
  // a time so that it leaves as whole 256-byte row segments (16 B per lane) instead of 2-byte
  // scattered stores (the accumulator has the neuron on the lane and the rows in registers).
  if constexpr (HEAD) {
    // fp32 staging of 64 rows x 128 hidden values at a time (register tile a of both wave row groups), row stride
    // 136 floats (conflict-free 4-byte writes / 16-byte reads), over the dead operand tiles; then one (row,
    // position) per thread: relu(h . Wout + bout), first maximum of the two outputs, one byte out.
    constexpr int kHStride = kDenseBN + 8;
    static_assert(64 * kHStride * 4 <= (kDenseBM + kDenseBN) * kRowBytes, "head staging must fit the operand tiles");
    float *const hstage = reinterpret_cast<float *>(lds_all);
    const int wrow = wave >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int lcol = wc + 32 * b + li;
        const float bias = p.bias[col0 + lcol];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = wrow * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = acc[a][b][r] + bias;
          v = p.act == 2 ? (v > 0.f ? v : 0.f) : tanh_f32x(v);
          hstage[lrow * kHStride + lcol] = v;
        }
      }
      __syncthreads();
      const int lrow = tid & 63;
      const int lpos = __builtin_amdgcn_readfirstlane(tid >> 6);   // a wave holds ONE position
      const int64_t pos = static_cast<int64_t>(col_blk) * 4 + lpos;
      const int64_t row = row0 + (lrow >> 5) * 64 + 32 * a + (lrow & 31);
      if (pos < p.n_pos) {
        const float *hw = p.head_w + pos * 64;
        const float *hv = hstage + lrow * kHStride + 32 * lpos;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 h4 = *reinterpret_cast<const f32x4 *>(hv + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = __builtin_fmaf(h4[e], hw[2 * (4 * q + e)], s0);
            s1 = __builtin_fmaf(h4[e], hw[2 * (4 * q + e) + 1], s1);
          }
        }
        float v0 = s0 + p.head_b[2 * pos], v1 = s1 + p.head_b[2 * pos + 1];
        v0 = v0 > 0.f ? v0 : 0.f;
        v1 = v1 > 0.f ? v1 : 0.f;
        const unsigned bit = v1 > v0 ? 1u : 0u;
        const bool live = row < p.n_rows;
        if (live) {
          p.bits[pos * p.n_rows + row] = static_cast<uint8_t>(bit);
          if (p.head_act) {
            const f32x2 o = {v0, v1};
            *reinterpret_cast<f32x2 *>(p.head_act + (pos * p.n_rows + row) * 2) = o;
          }
        }
        if (p.errors) {
          const bool wrong = live && p.labels[pos * p.n_rows + row] != bit;
          const unsigned long long m = __ballot(wrong);
          if (lane == 0 && m) atomicAdd(p.errors + pos, static_cast<unsigned long long>(__popcll(m)));
        }
      }
    }
    return;
  }
  constexpr int kOutStride = kDenseBN * 2 + 16;  // bytes per staged row
  static_assert(kDenseBM * kOutStride <= 2 * kDenseBM * kRowBytes, "output staging must fit the operand tiles");
  _Float16 lo_keep[2][2][16];
  __syncthreads();  // every wave has finished reading the operand tiles
#pragma unroll
  for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int lcol = wc + 32 * b + li;
      const float bias = p.bias[col0 + lcol];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = wr + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
          _Float16 out;
          if (plane == 0) {
            float v = acc[a][b][r] + bias;
            v = p.act == 2 ? (v > 0.f ? v : 0.f) : tanh_f32x(v);
            out = static_cast<_Float16>(v);
            lo_keep[a][b][r] = static_cast<_Float16>(v - static_cast<float>(out));
          } else {
            out = lo_keep[a][b][r];
          }
          *reinterpret_cast<_Float16 *>(stage + lrow * kOutStride + lcol * 2) = out;
        }
      }
    }
    __syncthreads();
    _Float16 *dst = plane == 0 ? p.C_hi : p.C_lo;
#pragma unroll
    for (int it = 0; it < (kDenseBM * kDenseBN * 2 / 16) / 256; ++it) {
      const int chunk = tid + 256 * it;        // 16 chunks of 16 B per row
      const int lrow = chunk >> 4, c16 = chunk & 15;
      const int64_t row = row0 + lrow;
      if (row < p.n_rows) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + lrow * kOutStride + c16 * 16);
        *reinterpret_cast<u32x4 *>(dst + row * p.ldc + col0 + c16 * 8) = v;
      }
    }
    __syncthreads();
  }
}

// head for split activations: x = hi + lo reconstructed in fp32, fp32 weights
__global__ __launch_bounds__(256) void head_argmax_f16x3_kernel(const HeadF16Params p) {
  const int lane = threadIdx.x & 31;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  const bool live = row < p.n_rows;
  const _Float16 *ah = p.A_hi + (live ? row : 0) * p.lda;
  const _Float16 *al = p.A_lo + (live ? row : 0) * p.lda;
  float best = -__builtin_inff();
  int bidx = 0;
  for (int c = 0; c < p.n_classes; ++c) {
    const float *w = p.Wt + static_cast<int64_t>(c) * p.Kpad;
    float s = 0.f;
    for (int k = lane * 8; k < p.Kpad; k += 256) {
      const f16x8 h = *reinterpret_cast<const f16x8 *>(ah + k);
      const f16x8 l = *reinterpret_cast<const f16x8 *>(al + k);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        s = __builtin_fmaf(static_cast<float>(h[e]) + static_cast<float>(l[e]), w[k + e], s);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 32);
    const float z = s + p.bias[c];
    const float v = p.act_fn == 2 ? (z > 0.f ? z : 0.f) : tanh_f32x(z);
    if (p.act && live && lane == 0) p.act[row * p.n_classes + c] = v;
    if (v > best) {
      best = v;
      bidx = c;
    }
  }
  if (live && lane == 0) {
    if (p.cls32) p.cls32[row] = bidx;
    if (p.cls64) p.cls64[row] = bidx;
  }
}

nnv_status dense_f16x3_launch(const DenseF16Params &p, bool gather, hipStream_t stream) {
  const int64_t blocks = (p.nblk_rows + 7) / 8 * 8 * p.nblk_m;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  static const bool use32 = ab_env("NNV_DENSE16_BK32") != nullptr;  // A/B builds only
  if (gather)
    hipLaunchKernelGGL((dense_f16x3_kernel<true, 32>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  else if (p.Kpad % 64 == 0 && !use32)
    hipLaunchKernelGGL((dense_f16x3_kernel<false, 64>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((dense_f16x3_kernel<false, 32>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

nnv_status bitwise_f16x3_launch(const DenseF16Params &p, hipStream_t stream) {
  const int64_t blocks = (p.nblk_rows + 7) / 8 * 8 * p.nblk_m;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  if (!p.bits || !p.head_w || !p.head_b || (p.errors && !p.labels)) return NNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL((dense_f16x3_kernel<true, 32, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

nnv_status head_f16x3_launch(const HeadF16Params &p, hipStream_t stream) {
  const int64_t blocks = (p.n_rows * 32 + 255) / 256;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(head_argmax_f16x3_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
