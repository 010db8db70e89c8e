// Layer-at-a-time path for WIDE feed-forward decoders (hidden widths that do
// not fit the register-chained kernel of nnv_mlp_decode.hip): one LDS-tiled
// fp32-MFMA GEMM launch per dense layer with the bias add and tanh fused into
// the epilogue, activations in HBM between layers, and a small head kernel for
// a narrow class layer.
//
// First user: the reference's sliding-window decoder
//   src/fnn_viterbi_seq_predict.py:293-336  window i = coded[2i : 2i+window)
//   src/fnn_viterbi_seq_predict.py:343-414  200 -> 500 -> 500 -> 500 -> 500 -> 2, tanh everywhere
//   src/fnn_viterbi_seq_predict.py:497,592-611  argmax over the 2 outputs = the decoded bit
// Layer 1 gathers its rows straight from the single received stream
// (row i starts at stream + i*row_stride), so the [N, window] matrix the
// reference materialises in Python is never built.
//
// GEMM: C[rows, M] = tanh(A[rows, K] * W[K, M] + b), exact fp32
// (v_mfma_f32_32x32x2_f32 = k-ordered fmaf chain), op order matmul, add, tanh
// as in the reference.  128x128 block tile, BK = 32, 4 waves x (2x2) 32x32 MFMA
// tiles, operands staged through LDS as [row][k] / [neuron][k] with a 36-float
// row stride so that every ds_read_b128 / ds_write_b128 is conflict free; the
// k index inside an 8-wide group is permuted (lane half h takes k0+4h..k0+4h+3)
// so one b128 read feeds 4 MFMAs for both operands; next k-tile's global loads
// are issued before the MFMAs of the current one, and the fragment reads of k-group
// g + 1 are interleaved with the 16 MFMAs of group g (one ds_read per four MFMAs).
#include "nnv_internal.h"

namespace nnv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int kBM = kDenseBM;

__device__ __forceinline__ float dense_tanh_f32(float z) {
  const float e = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  return __builtin_fmaf(-2.0f, r, 1.0f);
}

// BK = 32 (row stride 36 floats) or 64 (68): both strides make every b128 access conflict free.
// ACT: 1 tanh, 2 relu (compile time: a run-time switch became three scalar branches per value)
// BN: output columns per block.  128 (2 x 2 waves of 64 x 64) for wide layers; 32 (4 x 1 waves of
// 32 x 32, BK = 32 only) for NARROW layers such as the bit-wise decoder's 1036 -> 30: those are
// HBM-shaped (4 KB of row in for 62 kFLOP), and padding 30 outputs to a 128-wide tile made them
// MFMA-bound at 0.21 of the HBM roofline.
// HEAD (bit-wise decoder bank, src/fnn_viterbi_bitwise.py:388-397,504-505): the output columns are
// P independent networks' hidden layers, 32 columns per bit position (30 used), and the epilogue
// finishes each network in place -- relu(h . Wout + bout) -> argmax over its 2 outputs -- so that the
// [rows, 32 P] hidden matrix never goes to HBM; one decided bit per (position, row) leaves instead.
// CLS (a WIDE class layer, 33..65536 classes -- e.g. the reference's DecoderConfig at block_length 13..15, 32 -> 2^b):
// the epilogue keeps, per row, the first maximum of the activated outputs over this block's 128 columns and writes one
// (value, class) pair per (row, column block) instead of the tile: the [rows, 2^b] output matrix (128 KB per row at
// b = 15) never goes to HBM; partial_argmax_kernel merges the pairs.  Same values, same first-maximum rule as
// materialising the row and scanning it (row_argmax_kernel, which stays for the debug output).
template <bool GATHER, int kBK, int ACT, int BN, bool HEAD = false, bool CLS = false>
__global__ __launch_bounds__(256) void dense_tanh_kernel(const DenseParams p) {
  static_assert(BN == 128 || (BN == 32 && kBK == 32), "tile shapes");
  static_assert(!(HEAD && CLS) && (!CLS || BN == 128), "one epilogue at a time; the class epilogue has the 128-column tile only");
  constexpr int kBN = BN;                   // shadows the namespace constant inside this kernel
  constexpr int WT = BN == 128 ? 2 : 1;     // 32 x 32 MFMA tiles per wave, per dimension
  constexpr int kLdsStride = kBK + 4;
  constexpr int kPasses = kBK / 8;          // float4 loads per thread per k-tile of A (128 rows)
  constexpr int kPassesW = BN == 128 ? kPasses : 1;  // ... of W (BN rows)
  constexpr int kRowsPerPass = 1024 / kBK;  // 32 (BK 32) or 16 (BK 64)
  __shared__ __attribute__((aligned(16))) float lds_all[(kBM + kBN) * kLdsStride];
  float *const lds_a = lds_all;
  float *const lds_w = lds_all + kBM * kLdsStride;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware block -> tile map.  Blocks b and b+8 are dispatched to the same XCD (its own L2),
  // in order.  The nblk_m column blocks of one row block read the same A rows, so they are
  // given consecutive slots of ONE XCD: the A k-slabs are then fetched once into that L2 and
  // hit by the other column blocks, instead of being fetched by four different L2s.
  // (A placement guess can only cost speed: every block computes a fixed, disjoint tile.)
  const int64_t blk = blockIdx.x;
  const int xcd = static_cast<int>(blk & 7);
  const int64_t slot = blk >> 3;
  int col_blk;
  int64_t row_blk;
  if (HEAD && p.col_group > 0) {
    // Bit-wise bank: MANY column blocks (128 for 512 positions: a 68 MB stacked first layer) against a
    // small A row block (0.5 MB).  Walking all column blocks per row block re-streams W once per row
    // block (168 GB fetched per launch, PMC round 2).  Instead an XCD takes the column blocks in groups
    // of `col_group` (W slab of col_group x 0.54 MB, L2 resident) and walks its ROW blocks fastest
    // inside a group: W is fetched once per group and XCD, A once per group (its col_group blocks are
    // consecutive slots, so they run together and share the rows in L2).
    const int64_t n_r8 = (p.nblk_rows + 7) / 8;                  // row blocks per XCD (grid padded)
    const int64_t per_group = n_r8 * p.col_group;
    const int groups = (p.nblk_m + p.col_group - 1) / p.col_group;
    int64_t g = slot / per_group;
    if (g > groups - 1) g = groups - 1;                           // the last, smaller group
    const int gsz = g == groups - 1 ? p.nblk_m - static_cast<int>(g) * p.col_group : p.col_group;
    const int64_t rem = slot - g * per_group;
    col_blk = static_cast<int>(g) * p.col_group + static_cast<int>(rem % gsz);
    row_blk = (rem / gsz) * 8 + xcd;
  } else {
    col_blk = static_cast<int>(slot % p.nblk_m);
    row_blk = (slot / p.nblk_m) * 8 + xcd;
  }
  if (row_blk >= p.nblk_rows) return;  // grid is padded to a multiple of 8 row blocks
  const int64_t row0 = row_blk * kBM;
  const int col0 = col_blk * kBN;

  // global -> register staging: 128 rows x 32 k = 1024 float4, 4 per thread
  const int ld_row = tid / (kBK / 4);        // + kRowsPerPass per pass
  const int ld_kq = (tid % (kBK / 4)) * 4;   // 0, 4, .., BK-4
  f32x4 ra[kPasses], rw[kPassesW];

  auto load_tiles = [&](int kt) {
    const int k = kt * kBK + ld_kq;
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      int64_t r = row0 + ld_row + kRowsPerPass * ps;
      r = r < p.n_rows ? r : p.n_rows - 1;
      const float *src = p.A + r * p.a_row_stride + k;
      if (GATHER && (kt + 1) * kBK <= p.K) {  // whole k-tile inside the row
        if (p.a16) {                            // rows 16-byte aligned (matrix rows, K % 4 == 0): one dwordx4
          ra[ps] = *reinterpret_cast<const f32x4 *>(src);
        } else {                                // windows of a stream (8-byte aligned): two plain 8-byte loads
          const f32x2 t0 = *reinterpret_cast<const f32x2 *>(src);
          const f32x2 t1 = *reinterpret_cast<const f32x2 *>(src + 2);
          const f32x4 v = {t0[0], t0[1], t1[0], t1[1]};
          ra[ps] = v;
        }
      } else if (GATHER) {  // last k-tile: the row ends inside it, guarded loads, zero fill
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k + 1 < p.K) {
          const f32x2 t = *reinterpret_cast<const f32x2 *>(src);
          v[0] = t[0];
          v[1] = t[1];
        } else if (k < p.K) {
          v[0] = src[0];
        }
        if (k + 3 < p.K) {
          const f32x2 t = *reinterpret_cast<const f32x2 *>(src + 2);
          v[2] = t[0];
          v[3] = t[1];
        } else if (k + 2 < p.K) {
          v[2] = src[2];
        }
        ra[ps] = v;
      } else {
        ra[ps] = *reinterpret_cast<const f32x4 *>(src);  // activations: [rows][Kpad], 16-byte aligned
      }
      if (ps < kPassesW)
        rw[ps] = *reinterpret_cast<const f32x4 *>(p.Wt + static_cast<int64_t>(col0 + ld_row + kRowsPerPass * ps) * p.Kpad + k);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int ps = 0; ps < kPasses; ++ps) {
      *reinterpret_cast<f32x4 *>(lds_a + (ld_row + kRowsPerPass * ps) * kLdsStride + ld_kq) = ra[ps];
      if (ps < kPassesW)
        *reinterpret_cast<f32x4 *>(lds_w + (ld_row + kRowsPerPass * ps) * kLdsStride + ld_kq) = rw[ps];
    }
  };

  f32x16 acc[WT][WT];
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // wave grid: 2 x 2 waves of 64 x 64 (BN 128) or 4 x 1 waves of 32 x 32 (BN 32)
  const int wrow = BN == 128 ? (wave >> 1) : wave;   // wave's row group
  const int wr = wrow * 32 * WT, wc = BN == 128 ? (wave & 1) * 64 : 0;
  const int nk = p.Kpad / kBK;
  load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // every wave is done reading the previous tile
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles(kt + 1);
    // software pipeline: group g + 1's fragments are read while group g's 16 MFMAs issue
    f32x4 fa[2][WT], fb[2][WT];
    auto read_frags = [&](int g, int buf) {
#pragma unroll
      for (int t = 0; t < WT; ++t) {
        fa[buf][t] = *reinterpret_cast<const f32x4 *>(lds_a + (wr + 32 * t + li) * kLdsStride + 8 * g + 4 * lh);
        fb[buf][t] = *reinterpret_cast<const f32x4 *>(lds_w + (wc + 32 * t + li) * kLdsStride + 8 * g + 4 * lh);
      }
    };
    read_frags(0, 0);
#pragma unroll
    for (int g = 0; g < kBK / 8; ++g) {
      if (g + 1 < kBK / 8) read_frags(g + 1, (g + 1) & 1);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < WT; ++a)
#pragma unroll
          for (int b = 0; b < WT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][a][e], fb[g & 1][b][e], acc[a][b], 0, 0, 0);
      if (BN == 128 && g + 1 < kBK / 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one ds_read
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // four MFMAs
        }
      }
    }
  }

  // epilogue: + bias, activation, store.  D has the column (neuron) on the lane and the rows in
  // the registers, so storing it as it stands is 64 four-byte stores per lane whose issue takes
  // ~18 % of the launch (measured: the same kernel without its epilogue).  Instead the tile goes
  // through the operand tiles' LDS, 64 rows at a time (register tile a of every wave), and
  // leaves as whole 512-byte rows, 16 bytes per lane.  Staged row stride 136 floats: the two
  // lane halves of one ds_write_b32 (rows 4 apart) land 32 banks apart -> conflict free; so are
  // the 16-byte reads.
  constexpr int kOutStride = kBN + 8;
  constexpr int kStageRows = (BN == 128 ? 2 : 4) * 32;   // one register tile `a` of every wave row group
  static_assert(kStageRows * kOutStride <= (kBM + kBN) * kLdsStride, "output staging must fit the operand tiles");
  constexpr int kChunksPerRow = kBN / 4;                  // 16-byte chunks per staged row
  float *const stage = lds_all;
#pragma unroll
  for (int a = 0; a < WT; ++a) {
    __syncthreads();  // operand tiles (a = 0) / previous half (a = 1) fully consumed
#pragma unroll
    for (int b = 0; b < WT; ++b) {
      const int lcol = wc + 32 * b + li;
      const float bias = p.bias[col0 + lcol];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = wrow * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[a][b][r] + bias;
        if constexpr (ACT == 1) v = dense_tanh_f32(v);
        else if constexpr (ACT == 2) v = v > 0.f ? v : 0.f;
        stage[lrow * kOutStride + lcol] = v;
      }
    }
    __syncthreads();
    if constexpr (HEAD) {
      // one (row, position) per thread: 64 staged rows x 4 positions (BN 128) or 128 x 1 (BN 32);
      // a wave holds ONE position (weights through scalar loads), its lanes consecutive rows
      constexpr int kItems = kStageRows * (kBN / 32);
      if (tid < kItems) {
        const int lrow = tid % kStageRows;
        const int lpos = __builtin_amdgcn_readfirstlane(tid / kStageRows);
        const int64_t pos = static_cast<int64_t>(col_blk) * (kBN / 32) + lpos;
        const int64_t row = row0 + (lrow >> 5) * 32 * WT + 32 * a + (lrow & 31);
        if (pos < p.n_pos) {
          const float *hw = p.head_w + pos * 64;   // [32 hidden][2 outputs], zero rows beyond n_hidden
          const float *hv = stage + lrow * kOutStride + 32 * lpos;
          float s0 = 0.f, s1 = 0.f;                // op order of the reference: matmul, then add, then relu
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
          v0 = v0 > 0.f ? v0 : 0.f;                // ReLU on the logits too (src/fnn_viterbi_bitwise.py:397)
          v1 = v1 > 0.f ? v1 : 0.f;
          const unsigned bit = v1 > v0 ? 1u : 0u;  // tf.argmax over 2 outputs, first maximum on ties
          const bool live = row < p.n_rows;
          if (live) {
            p.bits[pos * p.n_rows + row] = static_cast<uint8_t>(bit);
            if (p.head_act) {
              const f32x2 o = {v0, v1};
              *reinterpret_cast<f32x2 *>(p.head_act + (pos * p.n_rows + row) * 2) = o;
            }
          }
          if (p.errors) {                          // needs labels
            const bool wrong = live && p.labels[pos * p.n_rows + row] != bit;
            const unsigned long long m = __ballot(wrong);
            if (lane == 0 && m) atomicAdd(p.errors + pos, static_cast<unsigned long long>(__popcll(m)));
          }
        }
      }
      continue;
    }
    if constexpr (CLS) {
      // 64 staged rows x 128 activated columns: 4 threads per row, 32 columns each, merged by two xor-shuffles
      const int lrow = tid >> 2, seg = tid & 3;
      const float *hv = stage + lrow * kOutStride + 32 * seg;
      float best = -__builtin_inff();
      int bidx = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(hv + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = col0 + 32 * seg + 4 * q + e;
          const bool take = c < p.n_valid && v4[e] > best;   // strict >: the lowest index keeps a tie
          best = take ? v4[e] : best;
          bidx = take ? c : bidx;
        }
      }
#pragma unroll
      for (int off = 1; off <= 2; off <<= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bidx, off);
        const bool take = ov > best || (ov == best && oi < bidx);
        best = take ? ov : best;
        bidx = take ? oi : bidx;
      }
      const int64_t row = row0 + (lrow >> 5) * 32 * WT + 32 * a + (lrow & 31);
      if (seg == 0 && row < p.n_rows) {
        p.part_val[row * p.nblk_m + col_blk] = best;
        p.part_idx[row * p.nblk_m + col_blk] = bidx;
      }
      continue;
    }
#pragma unroll
    for (int it = 0; it < (kStageRows * kChunksPerRow) / 256; ++it) {
      const int chunk = tid + 256 * it;
      const int lrow = chunk / kChunksPerRow, c4 = (chunk % kChunksPerRow) * 4;
      const int64_t row = row0 + (lrow >> 5) * 32 * WT + 32 * a + (lrow & 31);
      if (row < p.n_rows)
        *reinterpret_cast<f32x4 *>(p.C + row * p.ldc + col0 + c4) =
            *reinterpret_cast<const f32x4 *>(stage + lrow * kOutStride + c4);
    }
  }
}

// Narrow class layer (n_classes <= 32, e.g. the 2-way bit decision of the sliding-window
// decoder): one half-wave per row, float4 reads of the row, per-class partial dot products,
// xor-shuffle reduction, tanh, first-maximum argmax (src/fnn_viterbi_seq_predict.py:497).
__global__ __launch_bounds__(256) void head_argmax_kernel(const HeadParams p) {
  const int lane = threadIdx.x & 31;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  const bool live = row < p.n_rows;
  const float *a = p.A + (live ? row : 0) * p.lda;
  float best = -__builtin_inff();
  int bidx = 0;
  for (int c = 0; c < p.n_classes; ++c) {
    const float *w = p.Wt + static_cast<int64_t>(c) * p.Kpad;
    float s = 0.f;
    for (int k = lane * 4; k < p.Kpad; k += 128) {
      const f32x4 av = *reinterpret_cast<const f32x4 *>(a + k);
      const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + k);
      s = __builtin_fmaf(av[0], wv[0], s);
      s = __builtin_fmaf(av[1], wv[1], s);
      s = __builtin_fmaf(av[2], wv[2], s);
      s = __builtin_fmaf(av[3], wv[3], s);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 32);
    const float z = s + p.bias[c];
    const float v = p.act_fn == 2 ? (z > 0.f ? z : 0.f) : dense_tanh_f32(z);
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

// one 32-lane group per row: first maximum over its nblk (value, class) pairs
__global__ __launch_bounds__(256) void partial_argmax_kernel(const float *pv, const int32_t *pi, int64_t n_rows, int nblk,
                                                             int32_t *cls32, int64_t *cls64) {
  const int lane = threadIdx.x & 31;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  if (row >= n_rows) return;
  float best = -__builtin_inff();
  int bidx = 0x7fffffff;
  for (int c = lane; c < nblk; c += 32) {
    const float v = pv[row * nblk + c];
    const int i = pi[row * nblk + c];
    if (v > best || (v == best && i < bidx)) {
      best = v;
      bidx = i;
    }
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 32);
    const int oi = __shfl_xor(bidx, off, 32);
    if (ov > best || (ov == best && oi < bidx)) {
      best = ov;
      bidx = oi;
    }
  }
  if (lane == 0) {
    if (bidx == 0x7fffffff) bidx = 0;   // a row of NaNs: unspecified, but a class
    if (cls32) cls32[row] = bidx;
    if (cls64) cls64[row] = bidx;
  }
}

nnv_status partial_argmax_launch(const float *part_val, const int32_t *part_idx, int64_t n_rows, int32_t nblk,
                                 int32_t *cls32, int64_t *cls64, hipStream_t stream) {
  const int64_t blocks = (n_rows * 32 + 255) / 256;
  if (blocks <= 0 || blocks > 0x7fffffffLL || nblk < 1) return NNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(partial_argmax_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, part_val, part_idx,
                     n_rows, nblk, cls32, cls64);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

nnv_status dense_launch(const DenseParams &p, bool gather, hipStream_t stream, bool cls) {
  const int64_t blocks = (p.nblk_rows + 7) / 8 * 8 * p.nblk_m;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  // BK = 64 measured 6 % slower than BK = 32 on the 512x512 layers (more staging registers and LDS,
  // same MFMA count); kept as an instantiation for A/B runs only.
  static const bool use64 = ab_env("NNV_DENSE_BK64") != nullptr;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  if (p.act != 1 && p.act != 2) return NNV_ERR_INVALID_ARGUMENT;
  if (p.bn != 128 && p.bn != 32) return NNV_ERR_INVALID_ARGUMENT;
  const bool tanh_act = p.act == 1;
  if (cls) {  // a wide class layer with the argmax in the epilogue (never the first layer: depth >= 2)
    if (gather || p.bn != 128 || !p.part_val || !p.part_idx || p.n_valid < 1) return NNV_ERR_INVALID_ARGUMENT;
    if (tanh_act) hipLaunchKernelGGL((dense_tanh_kernel<false, 32, 1, 128, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((dense_tanh_kernel<false, 32, 2, 128, false, true>), grid, block, 0, stream, p);
    return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
  }
#define NNV_DENSE_LAUNCH(G, BK, BNV)                                                              \
  do {                                                                                            \
    if (tanh_act) hipLaunchKernelGGL((dense_tanh_kernel<G, BK, 1, BNV>), grid, block, 0, stream, p); \
    else hipLaunchKernelGGL((dense_tanh_kernel<G, BK, 2, BNV>), grid, block, 0, stream, p);          \
  } while (0)
  if (p.bn == 32) {  // narrow layer
    if (gather) NNV_DENSE_LAUNCH(true, 32, 32);
    else NNV_DENSE_LAUNCH(false, 32, 32);
  } else if (gather) {  // layer 1: raw rows, K padded to 32
    NNV_DENSE_LAUNCH(true, 32, 128);
  } else if (p.Kpad % 64 == 0 && use64) {
    NNV_DENSE_LAUNCH(false, 64, 128);
  } else {
    NNV_DENSE_LAUNCH(false, 32, 128);
  }
#undef NNV_DENSE_LAUNCH
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

// The bit-wise decoder bank: rows gathered from [n_rows][K] (any even K), ReLU, fused per-position head.
nnv_status bitwise_launch(const DenseParams &p, hipStream_t stream) {
  const int64_t blocks = (p.nblk_rows + 7) / 8 * 8 * p.nblk_m;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  if (p.bn != 128 && p.bn != 32) return NNV_ERR_INVALID_ARGUMENT;
  if (!p.bits || !p.head_w || !p.head_b || (p.errors && !p.labels)) return NNV_ERR_INVALID_ARGUMENT;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  // One position (32-column tile, HBM-shaped): 9 KB of unused dynamic LDS caps the kernel at 5 workgroups per
  // CU instead of 6.  Measured on the reference's default shape (300 k x 1036): 6 per CU 0.306 ms, 5 / 4 / 3 / 2
  // per CU 0.287 / 0.288 / 0.290 / 0.287 ms (0.54 of the HBM peak): the sixth workgroup only adds contention.
  constexpr int kCapLds = 9 * 1024;
  if (p.bn == 128) hipLaunchKernelGGL((dense_tanh_kernel<true, 32, 2, 128, true>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((dense_tanh_kernel<true, 32, 2, 32, true>), grid, block, kCapLds, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

nnv_status head_launch(const HeadParams &p, hipStream_t stream) {
  const int64_t blocks = (p.n_rows * 32 + 255) / 256;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return NNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(head_argmax_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
