// K3: synthetic received-block generator (benchmark harness / on-device data
// path).  One thread produces one row (tile) of the reference's [N, 2b] input
// matrix: random info bits -> rate-1/2 convolutional code from zero state ->
// BPSK -> AWGN.  Replaces the per-row Python loop of train_test_dataset
// (src/fnn_viterbi.py:343-388) for random rows.
//
// All randomness is Philox4x32-10 keyed by (seed) with counter
// (row_lo, row_hi, draw, stream): the data of GLOBAL row r are the same
// whatever the sharding over GPUs and whatever the launch geometry.
#include "nnv_internal.h"

namespace nnv {

struct GenParams {
  int32_t code;
  int32_t tile_bits;
  int64_t n_rows;
  int64_t first_row;
  int64_t rows_per_block;
  int32_t n_sigma;
  float sigma[16];
  uint32_t seed_lo, seed_hi;
  float *x;
  int32_t *labels;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                              uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Box-Muller on 24-bit uniforms: u1 in (0,1], u2 in [0,1), on the hardware transcendentals:
// v_log_f32 (log2), v_sqrt_f32, and v_sin_f32 / v_cos_f32, which take their argument in
// REVOLUTIONS, i.e. sin(2 pi u2) is v_sin_f32(u2) with no range reduction.  Contract (the numpy
// model in oracle/awgn_model.py evaluates the same formulas in float64): x within 1e-5.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &n0, float &n1) {
  const float u1 = (static_cast<float>(a >> 8) + 1.0f) * 5.9604644775390625e-08f;
  const float u2 = static_cast<float>(b >> 8) * 5.9604644775390625e-08f;
  // -2 ln(u1) = -2 ln(2) log2(u1)
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}

// TB > 0: tile_bits known at compile time (7, 8, 10: the reference default and BASELINE's tilings):
// the row is built in registers and leaves as 16-byte stores (4 per 64-byte row instead of 16
// four-byte ones).  TB = 0: any tile_bits, scalar stores.  Same draws, same arithmetic, same data.
template <int TB>
__global__ __launch_bounds__(256) void awgn_generate_kernel(const GenParams p) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
       i < p.n_rows; i += stride) {
    const uint64_t r = static_cast<uint64_t>(p.first_row + i);
    const uint32_t r_lo = static_cast<uint32_t>(r), r_hi = static_cast<uint32_t>(r >> 32);
    uint32_t rnd[4];
    philox4x32_10(r_lo, r_hi, 0u, 0u, p.seed_lo, p.seed_hi, rnd);
    const int b = TB > 0 ? TB : p.tile_bits;
    const uint32_t cls = rnd[0] & ((1u << b) - 1u);  // MSB-first value of the info bits
    const int64_t blk = static_cast<int64_t>(r / static_cast<uint64_t>(p.rows_per_block));
    // Continuous mode (src/fnn_viterbi_mixed_snr.py:464-475: one stream per block, chopped into rows): the encoder
    // memory at the start of this row is the END of the previous row of the same block -- that row's info bits are
    // regenerated from its own counter, never read, so a row still depends on (seed, global row index) only.
    uint32_t prev = 0u;       // bit k = info bit at time -(k + 1)
    const bool first_of_block = r % static_cast<uint64_t>(p.rows_per_block) == 0;
    if ((p.code & NNV_CODE_CONTINUOUS) && !first_of_block) {
      const uint64_t rp = r - 1;
      uint32_t q[4];
      philox4x32_10(static_cast<uint32_t>(rp), static_cast<uint32_t>(rp >> 32), 0u, 0u, p.seed_lo, p.seed_hi, q);
      prev = q[0] & ((1u << b) - 1u);   // its last bit in time is the LSB
    }
    const int base_code = p.code & ~NNV_CODE_CONTINUOUS;
    // u(t) = info bit t (t = 0 is the first bit in time = the MSB of cls; t < 0 reaches into the previous row)
    auto u = [&](int t) -> uint32_t { return t < 0 ? (prev >> (-t - 1)) & 1u : (cls >> (b - 1 - t)) & 1u; };
    const float sigma = p.sigma[blk % p.n_sigma];
    float *px = p.x + i * (2 * b);
    float row[TB > 0 ? 2 * TB + 2 : 4];   // TB > 0: the whole row (+2: odd TB draws a spare pair)
    auto two_bits = [&](int t0) {  // 2 info bits -> 4 coded symbols -> 1 Philox draw
      uint32_t g[4];
      philox4x32_10(r_lo, r_hi, static_cast<uint32_t>(t0 >> 1), 1u, p.seed_lo, p.seed_hi, g);
      float nz[4];
      box_muller(g[0], g[1], nz[0], nz[1]);
      box_muller(g[2], g[3], nz[2], nz[3]);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int t = t0 + d;
        if (t >= b) break;
        uint32_t e0, e1;
        if (base_code == NNV_CODE_133171) {
          // src/fnn_viterbi.py:207-215: taps 133 -> {0,2,3,5,6}, 171 -> {0,1,2,3,6}
          e0 = u(t) ^ u(t - 2) ^ u(t - 3) ^ u(t - 5) ^ u(t - 6);
          e1 = u(t) ^ u(t - 1) ^ u(t - 2) ^ u(t - 3) ^ u(t - 6);
        } else {
          // src/fnn_viterbi.py:142-152: even = u[t]+u[t-2], odd = u[t]+u[t-1]+u[t-2],
          // except output index 1 which is u[0]+u[1] (:143)
          e0 = u(t) ^ u(t - 2);
          // (the quirk belongs to output index 1 of the STREAM: in continuous mode only a block's first row has it)
          e1 = (t == 0 && (first_of_block || !(p.code & NNV_CODE_CONTINUOUS))) ? (u(0) ^ u(1)) : (u(t) ^ u(t - 1) ^ u(t - 2));
        }
        const float v0 = (e0 ? 1.0f : -1.0f) + sigma * nz[2 * d];
        const float v1 = (e1 ? 1.0f : -1.0f) + sigma * nz[2 * d + 1];
        if constexpr (TB > 0) {
          row[2 * t] = v0;
          row[2 * t + 1] = v1;
        } else {
          px[2 * t] = v0;
          px[2 * t + 1] = v1;
        }
      }
    };
    if constexpr (TB > 0) {
#pragma unroll
      for (int t0 = 0; t0 < TB; t0 += 2) two_bits(t0);
    } else {
      for (int t0 = 0; t0 < b; t0 += 2) two_bits(t0);
    }
    if constexpr (TB > 0) {
      using f32x4 = __attribute__((ext_vector_type(4))) float;
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      if constexpr ((2 * TB) % 4 == 0) {   // rows are 16-byte aligned: x comes from hipMalloc, 2 TB * 4 B % 16 == 0
#pragma unroll
        for (int q = 0; q < 2 * TB / 4; ++q) {
          const f32x4 v = {row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]};
          *reinterpret_cast<f32x4 *>(px + 4 * q) = v;
        }
      } else {                             // 8-byte aligned rows (TB odd)
#pragma unroll
        for (int q = 0; q < TB; ++q) {
          const f32x2 v = {row[2 * q], row[2 * q + 1]};
          *reinterpret_cast<f32x2 *>(px + 2 * q) = v;
        }
      }
    }
    if (p.labels) p.labels[i] = static_cast<int32_t>(cls);
  }
}

nnv_status generate_launch(int32_t code, int32_t tile_bits, int64_t n_rows, int64_t first_row,
                           int64_t rows_per_block, const float *sigma, int32_t n_sigma,
                           uint64_t seed, float *x, int32_t *labels, hipStream_t stream) {
  GenParams p;
  p.code = code;
  p.tile_bits = tile_bits;
  p.n_rows = n_rows;
  p.first_row = first_row;
  p.rows_per_block = rows_per_block;
  p.n_sigma = n_sigma;
  for (int i = 0; i < 16; ++i) p.sigma[i] = i < n_sigma ? sigma[i] : 0.f;
  p.seed_lo = static_cast<uint32_t>(seed);
  p.seed_hi = static_cast<uint32_t>(seed >> 32);
  p.x = x;
  p.labels = labels;
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  const bool aligned = reinterpret_cast<uintptr_t>(x) % 16 == 0;
  if (tile_bits == 8 && aligned) hipLaunchKernelGGL(awgn_generate_kernel<8>, grid, block, 0, stream, p);
  else if (tile_bits == 10 && aligned) hipLaunchKernelGGL(awgn_generate_kernel<10>, grid, block, 0, stream, p);
  else if (tile_bits == 7 && aligned) hipLaunchKernelGGL(awgn_generate_kernel<7>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(awgn_generate_kernel<0>, grid, block, 0, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv
