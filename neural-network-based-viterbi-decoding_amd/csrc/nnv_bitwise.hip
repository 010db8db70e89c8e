// C ABI of the bit-wise decoder BANK: P independent  n_input -> n_hidden (ReLU) -> 2 (ReLU) -> argmax
// networks, one per information-bit position, evaluated on the same received rows in ONE launch.
//
// The reference trains and evaluates one such network per `bit_position`
// (src/fnn_viterbi_bitwise.py:326-399 build_network, :504-505 argmax eval): decoding a block of L
// bits is L separate Session.runs over the same [N, 2(6+L)] matrix.  Here the P first layers are
// stacked into one [n_input -> 32 P] GEMM (32 columns per position, n_hidden <= 32 used) on the fp32
// matrix cores and each position's 2-logit head is finished in the GEMM epilogue
// (dense_tanh_kernel<.., HEAD> in nnv_dense.hip): every row of x is read once per 4 positions' tile
// from L2/HBM instead of once per position per launch, the hidden matrix never reaches HBM, and one
// decided bit per (position, row) is written.  Host side only in this file.
#include <stdlib.h>

#include <new>
#include <vector>

#include "nnv_internal.h"

struct nnv_bitwise_handle {
  int32_t n_input, n_hidden, n_pos;
  int32_t kpad;    // n_input rounded up to the GEMM k-tile
  int32_t bn;      // 128 (4 positions per column block) or 32 (a single position)
  int32_t mpad;    // 32 * positions rounded up to bn
  int32_t device;
  int32_t precision;  // NNV_PRECISION_F32 | NNV_PRECISION_F16X3
  float *dev;      // f32:   Wt [mpad][kpad] | bias [mpad] | head_w [n_pos][32][2] | head_b [n_pos][2]
                   // f16x3: W_hi [mpad][kpad] f16 | W_lo [mpad][kpad] f16 (together the bytes of the fp32 matrix) | bias | ...
  int64_t off_bias, off_hw, off_hb;
  // K4c (nnv_bitwise_stream.hip): rows resident in LDS, first layers streamed from L2 -- the fp32 bank's default
  int32_t stream_ok;   // 1: stream_dev holds the fragment-order packing of the first layers, K4c may run
  int32_t single_ok;   // 1: ... and K4d may run (one launch per position; the default for small banks)
  int32_t num_cus;
  int32_t groups;      // ceil(n_input / 8)
  float *stream_dev;   // [pos][group][lane][4]
  // K4d in f16x3: hi / lo f16 fragments per step of 16 inputs
  int32_t single16_ok;
  int32_t steps;       // ceil(n_input / 16)
  uint16_t *single16_dev;   // [pos][step][hi | lo][lane 64][8 halfs]
  int32_t path;        // nnv_bitwise_path: AUTO, or the kernel nnv_bitwise_set_path pinned the handle to
};

namespace {
// Banks of up to this many positions run as one K4d launch per position.  300 k rows of 1036 inputs, same box: one
// position 0.254 ms (K4b, 32-column tile: 0.263), two 0.51 (K4b's 4-position column tile: 0.69), three 0.76 (0.68).
constexpr int kSingleMaxPositions = 2;
// f16x3: one position 0.242 ms (the 128-column f16x3 tile: 0.415), two 0.48 (0.414): only a single position
constexpr int kSingle16MaxPositions = 1;
// K4c from this many positions on: below, a wave per position leaves waves idle and quantises the run to
// ceil(P / 16) rounds (measured: P = 32 4.96 vs 4.92 ms for K4b, 48: 6.96 vs 7.31, 64: 8.96 vs 9.77).
constexpr int kStreamMinPositions = 33;

// The kernel a decode call on `h` over rows at `x_dev` runs.  AUTO: small banks one K4d launch per position (rows
// straight from HBM, the contraction split over the waves); large banks K4c (rows resident in LDS, first layers
// streamed); in between, and for anything neither takes, the LDS-tiled GEMM K4b.  A pinned path is taken when the
// handle holds its packing and the rows allow it (K4d needs 16-byte aligned rows), else K4b.
int resolve_path(const nnv_bitwise_handle *h, const float *x_dev) {
  const bool x16 = h->n_input % 4 == 0 && reinterpret_cast<uintptr_t>(x_dev) % 16 == 0;
  const bool may_single = h->path == NNV_BITWISE_PATH_AUTO || h->path == NNV_BITWISE_PATH_SINGLE;
  if ((h->single16_ok || h->single_ok) && x16 && may_single) return NNV_BITWISE_PATH_SINGLE;
  if (h->stream_ok && ((h->path == NNV_BITWISE_PATH_AUTO && h->n_pos >= kStreamMinPositions) ||
                       h->path == NNV_BITWISE_PATH_STREAM))
    return NNV_BITWISE_PATH_STREAM;
  return NNV_BITWISE_PATH_TILED;
}
}  // namespace

extern "C" {

int32_t nnv_bitwise_prepack(const nnv_bitwise_desc *d, nnv_bitwise_handle **out) {
  return nnv_bitwise_prepack_ex(d, NNV_PRECISION_F32, out);
}

int32_t nnv_bitwise_prepack_ex(const nnv_bitwise_desc *d, int32_t precision, nnv_bitwise_handle **out) {
  if (!d || !out) return NNV_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (precision != NNV_PRECISION_F32 && precision != NNV_PRECISION_F16X3) return NNV_ERR_INVALID_ARGUMENT;
  if (d->n_input < 2 || d->n_input > 16384 || (d->n_input & 1)) return NNV_ERR_INVALID_ARGUMENT;
  if (d->n_hidden < 1 || d->n_positions < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (d->n_hidden > 32 || d->n_positions > 65536) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (!d->W1_host || !d->b1_host || !d->Wout_host || !d->bout_host) return NNV_ERR_INVALID_ARGUMENT;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return NNV_ERR_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return NNV_ERR_NO_DEVICE;

  nnv_bitwise_handle *h = new (std::nothrow) nnv_bitwise_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  h->n_input = d->n_input;
  h->n_hidden = d->n_hidden;
  h->n_pos = d->n_positions;
  h->device = dev;
  h->kpad = (d->n_input + nnv::kDenseBK - 1) / nnv::kDenseBK * nnv::kDenseBK;
  h->precision = precision;
  h->path = NNV_BITWISE_PATH_AUTO;
  h->bn = (d->n_positions == 1 && precision == NNV_PRECISION_F32) ? 32 : 128;   // the f16x3 GEMM has the 128-column tile only
  h->mpad = (32 * d->n_positions + h->bn - 1) / h->bn * h->bn;
  const int64_t K = d->n_input, H = d->n_hidden, P = d->n_positions;
  h->off_bias = static_cast<int64_t>(h->mpad) * h->kpad;
  h->off_hw = h->off_bias + h->mpad;
  h->off_hb = h->off_hw + P * 64;
  std::vector<float> host(static_cast<size_t>(h->off_hb + 2 * P), 0.f);
  bool in_range = true;
  for (int64_t pos = 0; pos < P; ++pos) {
    const float *W1 = d->W1_host + pos * K * H;   // [n_input][n_hidden], right-multiply layout
    for (int64_t j = 0; j < H; ++j) {
      if (precision == NNV_PRECISION_F16X3) {   // hi plane [mpad][kpad] halves, then the lo plane
        uint16_t *hi = reinterpret_cast<uint16_t *>(host.data()) + (32 * pos + j) * h->kpad;
        uint16_t *lo = hi + static_cast<int64_t>(h->mpad) * h->kpad;
        for (int64_t k = 0; k < K; ++k) {
          const float v = W1[k * H + j];
          if (!(v > -65504.f && v < 65504.f)) in_range = false;
          hi[k] = nnv::f32_to_f16(v);
          lo[k] = nnv::f32_to_f16(v - nnv::f16_to_f32(hi[k]));
        }
      } else {
        float *row = host.data() + (32 * pos + j) * h->kpad;
        for (int64_t k = 0; k < K; ++k) row[k] = W1[k * H + j];
      }
      host[h->off_bias + 32 * pos + j] = d->b1_host[pos * H + j];
      host[h->off_hw + pos * 64 + 2 * j + 0] = d->Wout_host[(pos * H + j) * 2 + 0];
      host[h->off_hw + pos * 64 + 2 * j + 1] = d->Wout_host[(pos * H + j) * 2 + 1];
    }
    host[h->off_hb + 2 * pos + 0] = d->bout_host[2 * pos + 0];
    host[h->off_hb + 2 * pos + 1] = d->bout_host[2 * pos + 1];
  }
  h->dev = nullptr;
  if (!in_range) {  // a first-layer weight outside the f16 range: f16x3 refuses, as everywhere else
    delete h;
    return NNV_ERR_UNSUPPORTED_SHAPE;
  }
  if (hipMalloc(reinterpret_cast<void **>(&h->dev), host.size() * sizeof(float)) != hipSuccess) {
    delete h;
    return NNV_ERR_OUT_OF_MEMORY;
  }
  if (hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->dev);
    delete h;
    return NNV_ERR_HIP;
  }
  // K4c packing: position p, fragment group g, lane, element e holds W1[p][8g + 4 (lane >> 5) + e][lane & 31]
  h->stream_ok = 0;
  h->stream_dev = nullptr;
  h->num_cus = prop.multiProcessorCount;
  h->groups = static_cast<int32_t>((K + 7) / 8);
  const int64_t frag_floats = P * h->groups * 256;
  h->single_ok = 0;
  const bool k4c = P >= 2 && nnv::bank_stream_lds_bytes(h->groups) <= 160 * 1024;
  const bool k4d = P <= kSingleMaxPositions && nnv::bank_single_eligible(h->groups, K);
  if (precision == NNV_PRECISION_F32 && (k4c || k4d) && frag_floats * 4 < (1LL << 31)) {
    std::vector<float> frag(static_cast<size_t>(frag_floats), 0.f);
    for (int64_t pos = 0; pos < P; ++pos) {
      const float *W1 = d->W1_host + pos * K * H;
      for (int64_t g = 0; g < h->groups; ++g) {
        float *grp = frag.data() + (pos * h->groups + g) * 256;
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int64_t k = 8 * g + 4 * (lane >> 5) + e, j = lane & 31;
            if (k < K && j < H) grp[lane * 4 + e] = W1[k * H + j];
          }
      }
    }
    if (hipMalloc(reinterpret_cast<void **>(&h->stream_dev), frag.size() * sizeof(float)) == hipSuccess) {
      if (hipMemcpy(h->stream_dev, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess) {
        h->stream_ok = (k4c && nnv::bank_stream_prepare(h->groups, dev) == NNV_OK) ? 1 : 0;
        h->single_ok = (k4d && nnv::bank_single_prepare(h->groups, dev) == NNV_OK) ? 1 : 0;
      }
      if (!h->stream_ok && !h->single_ok) {
        (void)hipFree(h->stream_dev);
        h->stream_dev = nullptr;
      }
    }
  }
  // K4d, f16x3: position p, step s, plane (hi | lo), lane, element e holds W1[p][16 s + 8 (lane >> 5) + e][lane & 31]
  h->single16_ok = 0;
  h->single16_dev = nullptr;
  h->steps = static_cast<int32_t>((K + 15) / 16);
  if (precision == NNV_PRECISION_F16X3 && P <= kSingle16MaxPositions && nnv::bank_single16_eligible(h->steps, K)) {
    std::vector<uint16_t> frag(static_cast<size_t>(P * h->steps * 1024), 0);
    for (int64_t pos = 0; pos < P; ++pos) {
      const float *W1 = d->W1_host + pos * K * H;
      for (int64_t st = 0; st < h->steps; ++st) {
        uint16_t *hi = frag.data() + (pos * h->steps + st) * 1024, *lo = hi + 512;
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int64_t k = 16 * st + 8 * (lane >> 5) + e, j = lane & 31;
            if (k < K && j < H) {
              const float v = W1[k * H + j];
              hi[lane * 8 + e] = nnv::f32_to_f16(v);
              lo[lane * 8 + e] = nnv::f32_to_f16(v - nnv::f16_to_f32(hi[lane * 8 + e]));
            }
          }
      }
    }
    if (hipMalloc(reinterpret_cast<void **>(&h->single16_dev), frag.size() * sizeof(uint16_t)) == hipSuccess) {
      if (hipMemcpy(h->single16_dev, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess &&
          nnv::bank_single16_prepare(h->steps, dev) == NNV_OK) {
        h->single16_ok = 1;
      } else {
        (void)hipFree(h->single16_dev);
        h->single16_dev = nullptr;
      }
    }
  }
  *out = h;
  return NNV_OK;
}

int32_t nnv_bitwise_free(nnv_bitwise_handle *h) {
  if (!h) return NNV_OK;
  const hipError_t e = hipFree(h->dev);
  const hipError_t e2 = h->stream_dev ? hipFree(h->stream_dev) : hipSuccess;
  const hipError_t e3 = h->single16_dev ? hipFree(h->single16_dev) : hipSuccess;
  delete h;
  return (e == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_bitwise_set_path(nnv_bitwise_handle *h, int32_t path) {
  if (!h) return NNV_ERR_INVALID_ARGUMENT;
  switch (path) {
    case NNV_BITWISE_PATH_AUTO:
    case NNV_BITWISE_PATH_TILED: break;
    case NNV_BITWISE_PATH_STREAM: if (!h->stream_ok) return NNV_ERR_UNSUPPORTED_SHAPE; break;
    case NNV_BITWISE_PATH_SINGLE: if (!h->single_ok && !h->single16_ok) return NNV_ERR_UNSUPPORTED_SHAPE; break;
    default: return NNV_ERR_INVALID_ARGUMENT;
  }
  h->path = path;
  return NNV_OK;
}

int32_t nnv_bitwise_path_for(const nnv_bitwise_handle *h, const float *x_dev) {
  return h ? resolve_path(h, x_dev) : -NNV_ERR_INVALID_ARGUMENT;
}

int32_t nnv_bitwise_decode_f32(const nnv_bitwise_handle *h, const float *x_dev, int64_t n_rows,
                               uint8_t *bits_dev, float *act_dev, const uint8_t *labels_dev,
                               uint64_t *errors_dev, void *stream) {
  if (!h || n_rows < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_dev || !bits_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (errors_dev && !labels_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (reinterpret_cast<uintptr_t>(x_dev) % 8 != 0) return NNV_ERR_INVALID_ARGUMENT;
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  const bool x16 = h->n_input % 4 == 0 && reinterpret_cast<uintptr_t>(x_dev) % 16 == 0;
  const int path = resolve_path(h, x_dev);
  const bool use_single16 = path == NNV_BITWISE_PATH_SINGLE && h->single16_ok;   // f16x3 handles only
  const bool use_single = path == NNV_BITWISE_PATH_SINGLE && !use_single16;
  const bool use_stream = path == NNV_BITWISE_PATH_STREAM;
  if (use_single || use_stream || use_single16) {
    nnv::BankStreamParams q{};
    q.x = x_dev;
    q.row_stride = h->n_input;
    q.n_rows = n_rows;
    q.n_blocks = (n_rows + 31) / 32;
    q.x16 = x16 ? 1 : 0;
    q.K = h->n_input;
    q.groups = h->groups;
    q.n_pos = h->n_pos;
    q.w_bytes = static_cast<int32_t>(static_cast<int64_t>(h->n_pos) * h->groups * 1024);
    q.w = h->stream_dev;
    q.bias = h->dev + h->off_bias;
    q.head_w = h->dev + h->off_hw;
    q.head_b = h->dev + h->off_hb;
    q.bits = bits_dev;
    q.head_act = act_dev;
    q.labels = labels_dev;
    q.errors = reinterpret_cast<unsigned long long *>(errors_dev);
    if (use_stream) return nnv::bank_stream_launch(q, h->num_cus, h->device, static_cast<hipStream_t>(stream));
    q.n_pos = 1;
    q.w_bytes = h->groups * 1024;
    if (use_single16) {                              // slots are steps of 16 inputs, fragments hi | lo f16
      q.groups = h->steps;
      q.w = reinterpret_cast<const float *>(h->single16_dev);
      q.w_bytes = h->steps * 2048;
    }
    for (int64_t pos = 0; pos < h->n_pos; ++pos) {   // one launch per position, stream-ordered
      const nnv_status e = use_single16 ? nnv::bank_single16_launch(q, h->num_cus, h->device, static_cast<hipStream_t>(stream))
                                        : nnv::bank_single_launch(q, h->num_cus, h->device, static_cast<hipStream_t>(stream));
      if (e != NNV_OK) return e;
      q.w += use_single16 ? static_cast<int64_t>(h->steps) * 512 : static_cast<int64_t>(h->groups) * 256;
      q.bias += 32;
      q.head_w += 64;
      q.head_b += 2;
      q.bits += n_rows;
      if (q.head_act) q.head_act += 2 * n_rows;
      if (q.labels) q.labels += n_rows;
      if (q.errors) q.errors += 1;
    }
    return NNV_OK;
  }
  nnv::DenseParams p{};
  p.A = x_dev;
  p.a_row_stride = h->n_input;
  p.a16 = (h->n_input % 4 == 0 && reinterpret_cast<uintptr_t>(x_dev) % 16 == 0) ? 1 : 0;  // bank: 79.0 -> 77.9 ms
  p.n_rows = n_rows;
  p.K = h->n_input;
  p.Kpad = h->kpad;
  p.Wt = h->dev;
  p.bias = h->dev + h->off_bias;
  p.C = nullptr;
  p.ldc = 0;
  p.bn = h->bn;
  p.nblk_m = h->mpad / h->bn;
  // Column blocks per W slab kept in an XCD's L2: 4 x 128 columns x kpad floats = 2.2 MB at the reference's 1036
  // inputs.  Measured L2 <- fabric bytes per launch (FETCH_SIZE x 2, 300 k rows x 512 positions, profiles/r03/
  // bank_tile_order.json): row-block-major (round 2) 168 GB; groups of 2 / 3 / 4 / 5 / 6 / 8 / 16 / 32 column blocks
  // 93 / 74 / 61 / 63 / 62 / 61 / 92 / 168 GB -- A is re-read once per group ((128 / G) x 1.24 GB), and from G = 5 on
  // the slab no longer stays resident beside the streaming rows (4 MB of L2 per XCD).  Launch time is the same
  // for every order (78.0-78.5 ms: the fp32 matrix pipe is the limiter, 0.78 of peak).  (A/B builds only:
  // NNV_BANK_GROUP overrides; 0 = the round-2 order.)
  static const int forced_group = [] { const char *e = nnv::ab_env("NNV_BANK_GROUP"); return e ? atoi(e) : -1; }();
  const int64_t slab_bytes = static_cast<int64_t>(h->bn) * h->kpad * 4;
  int group = forced_group >= 0 ? forced_group : static_cast<int>((2400LL * 1024) / slab_bytes);
  if (group > p.nblk_m) group = p.nblk_m;
  p.col_group = (p.nblk_m > 8 && group >= 2) ? group : 0;
  p.nblk_rows = (n_rows + nnv::kDenseBM - 1) / nnv::kDenseBM;
  p.act = 2;
  p.n_pos = h->n_pos;
  p.head_w = h->dev + h->off_hw;
  p.head_b = h->dev + h->off_hb;
  p.bits = bits_dev;
  p.head_act = act_dev;
  p.labels = labels_dev;
  p.errors = reinterpret_cast<unsigned long long *>(errors_dev);
  if (h->precision == NNV_PRECISION_F16X3) {
    nnv::DenseF16Params q{};
    q.A_f32 = x_dev;
    q.a_row_stride = h->n_input;
    q.n_rows = n_rows;
    q.K = h->n_input;
    q.Kpad = h->kpad;
    q.W_hi = reinterpret_cast<const _Float16 *>(h->dev);
    q.W_lo = q.W_hi + static_cast<int64_t>(h->mpad) * h->kpad;
    q.bias = p.bias;
    q.nblk_m = p.nblk_m;
    q.nblk_rows = p.nblk_rows;
    q.act = 2;
    q.col_group = p.col_group;
    q.n_pos = p.n_pos;
    q.head_w = p.head_w;
    q.head_b = p.head_b;
    q.bits = p.bits;
    q.head_act = p.head_act;
    q.labels = p.labels;
    q.errors = p.errors;
    return nnv::bitwise_f16x3_launch(q, static_cast<hipStream_t>(stream));
  }
  return nnv::bitwise_launch(p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
