// C ABI of libnnviterbi_hip.so (declared in include/nnviterbi.h).
// Host-side only: argument validation, weight pre-packing, launch planning.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "nnv_internal.h"

using nnv::DecodeParams;
using nnv::LaunchPlan;
using nnv::PackedShape;

namespace {

constexpr int kLdsLimitBytes = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may own all of it

bool is_gfx950(const hipDeviceProp_t &prop) { return strstr(prop.gcnArchName, "gfx950") != nullptr; }

int ceil_div(int a, int b) { return (a + b - 1) / b; }

using nnv::f16_to_f32;
using nnv::f32_to_f16;

nnv_status shape_from_desc(const nnv_mlp_desc *net, int32_t precision, PackedShape *s) {
  if (net->depth < 2 || net->depth > NNV_MAX_LAYERS) return NNV_ERR_UNSUPPORTED_SHAPE;
  for (int l = 0; l <= net->depth; ++l)
    if (net->dims[l] < 1) return NNV_ERR_INVALID_ARGUMENT;
  for (int l = 0; l < net->depth; ++l)
    if (!net->W_host[l] || !net->b_host[l]) return NNV_ERR_INVALID_ARGUMENT;
  if (precision != nnv::kPrecF32 && precision != nnv::kPrecF16x3 && precision != nnv::kPrecF16) return NNV_ERR_INVALID_ARGUMENT;
  memset(s, 0, sizeof(*s));
  s->precision = precision;
  s->depth = net->depth;
  s->nhid = net->depth - 1;
  s->n_in = net->dims[0];
  s->bp = ceil_div(s->n_in, 2);
  s->g1 = precision == nnv::kPrecF32 ? ceil_div(s->bp, 4) : 2 * ceil_div(s->bp, 8);
  for (int i = 0; i < s->nhid; ++i) s->mt[i] = ceil_div(net->dims[i + 1], 32);
  s->n_classes = net->dims[net->depth];
  if (s->n_classes > NNV_MLP_MAX_CLASSES) return NNV_ERR_UNSUPPORTED_SHAPE;
  s->mt_out = ceil_div(s->n_classes, 32);
  s->kt_out = s->mt[s->nhid - 1];
  // fp32: g1 groups per M-tile; f16x3: (g1/2) k-steps x {hi, lo} = g1 groups as well; f16: hi only = g1/2
  const int l1_groups = precision == nnv::kPrecF16 ? s->g1 / 2 : s->g1;
  const int64_t kt_floats = static_cast<int64_t>(nnv::class_tile_kb(precision, 1)) * 256;  // per (M-tile, k-tile)
  s->hid_floats = static_cast<int64_t>(s->mt[0]) * l1_groups * nnv::kFragGroupFloats;
  for (int i = 1; i < s->nhid; ++i)
    s->hid_floats += static_cast<int64_t>(s->mt[i]) * s->mt[i - 1] * kt_floats;
  s->out_floats = static_cast<int64_t>(s->mt_out) * s->kt_out * kt_floats;
  int mts = s->mt_out;
  for (int i = 0; i < s->nhid; ++i) mts += s->mt[i];
  s->bias_floats = 32LL * mts;
  // s->streams (resident or ring-swept class layer) is decided with the workgroup size: nnv::decode_configure
  return nnv::decode_supported(*s);
}

// f16x3 fragment order (nnv_internal.h).  Returns false if a weight is outside the f16 range.
//
// The f16x3 kernel hands r = 1 / (1 + 2^zeta), zeta = 2 log2(e) z, from layer to layer instead of
// tanh(z) = 1 - 2r (act16 in nnv_mlp_decode.hip), so the packed network is the reference's with
//   hidden layers scaled by c = 2 log2(e)        (their accumulators hold zeta)
//   layers >= 2 taking r as input                (W' = -2 W,  b' = b + sum_k W_k)
// i.e. W_eff = s_l * f_l * W and b_eff = s_l * (b + [l >= 2] sum_k W_k) with s_l = c for hidden
// layers, 1 for the class layer, f_l = 1 for layer 1, -2 after.  Computed in double and split
// into hi/lo halves from the double (22 significant bits), the same algebraic function.
// planes = 2: hi and lo groups (f16x3);  planes = 1: the hi groups only (single-product f16 mode)
bool pack_network_f16x3(const nnv_mlp_desc *net, const PackedShape &s, std::vector<float> *out, int planes = 2) {
  out->assign(static_cast<size_t>(s.hid_floats + s.out_floats + s.bias_floats), 0.f);
  uint16_t *w = reinterpret_cast<uint16_t *>(out->data());
  bool in_range = true;
  constexpr double kTwoLog2e = 2.8853900817779268;
  auto emit = [&](double v, uint16_t *hi, uint16_t *lo) {
    if (!(v > -65504.0 && v < 65504.0)) in_range = false;
    const uint16_t h = f32_to_f16(static_cast<float>(v));
    *hi = h;
    if (planes == 2) *lo = f32_to_f16(static_cast<float>(v - static_cast<double>(f16_to_f32(h))));
  };
  const int group_u16 = 512 * planes;  // uint16 per k-step: hi group (+ lo group)
  auto layer_scale = [&](int l) { return l == net->depth - 1 ? 1.0 : kTwoLog2e; };
  // a group = 64 lanes x 8 halves = 512 uint16; hi group then lo group per k-step
  {
    const int M = net->dims[1];
    const float *W = net->W_host[0];
    const int ks1 = s.g1 / 2;
    const double sc = layer_scale(0);
    for (int mt = 0; mt < s.mt[0]; ++mt)
      for (int ks = 0; ks < ks1; ++ks) {
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int t = 8 * ks + e;
            const int k = (lane >> 5) * s.bp + t;
            const int i = 32 * mt + (lane & 31);
            const double v = (t < s.bp && k < s.n_in && i < M) ? sc * W[static_cast<size_t>(k) * M + i] : 0.0;
            emit(v, w + lane * 8 + e, w + 512 + lane * 8 + e);
          }
        w += group_u16;
      }
  }
  for (int l = 1; l < net->depth; ++l) {
    const int K = net->dims[l], M = net->dims[l + 1];
    const int KT = s.mt[l - 1];
    const int MT = (l == net->depth - 1) ? s.mt_out : s.mt[l];
    const float *W = net->W_host[l];
    const double sc = -2.0 * layer_scale(l);
    for (int mt = 0; mt < MT; ++mt)
      for (int kt = 0; kt < KT; ++kt)
        for (int st = 0; st < 2; ++st) {
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const int r = 8 * st + e;
              const int k = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              const int i = 32 * mt + (lane & 31);
              const double v = (k < K && i < M) ? sc * W[static_cast<size_t>(k) * M + i] : 0.0;
              emit(v, w + lane * 8 + e, w + 512 + lane * 8 + e);
            }
          w += group_u16;
        }
  }
  float *bf = reinterpret_cast<float *>(w);
  for (int l = 0; l < net->depth; ++l) {
    const int K = net->dims[l], M = net->dims[l + 1];
    const int MT = (l == net->depth - 1) ? s.mt_out : s.mt[l];
    for (int i = 0; i < 32 * MT; ++i) {
      double b = 0.0;
      if (i < M) {
        b = net->b_host[l][i];
        if (l >= 1)
          for (int k = 0; k < K; ++k) b += net->W_host[l][static_cast<size_t>(k) * M + i];
        b *= layer_scale(l);
      }
      *bf++ = static_cast<float>(b);
    }
  }
  return in_range;
}

// Fragment order: see nnv_internal.h.  The hidden layers are packed scaled by 2 log2(e) (their
// accumulators then hold the argument of the exp2 in tanh z = 1 - 2/(1 + 2^{2 log2(e) z}), act16 in
// nnv_mlp_decode.hip); the class layer keeps its scale (its raw pre-activations feed the argmax).
// The kernel hands r = 1/(1 + 2^zeta) from layer to layer instead of tanh z = 1 - 2r,
// so layers >= 2 are packed as W' = -2 W (exact) and b' = b + sum_k W_k (double, one rounding).
void pack_network(const nnv_mlp_desc *net, const PackedShape &s, std::vector<float> *out) {
  out->assign(static_cast<size_t>(s.hid_floats + s.out_floats + s.bias_floats), 0.f);
  float *w = out->data();
  constexpr double kTwoLog2e = 2.8853900817779268;
  auto scaled = [&](int l, double v) {
    const double f = l >= 1 ? -2.0 : 1.0;
    return static_cast<float>((l == net->depth - 1 ? f : f * kTwoLog2e) * v);
  };
  // layer 1
  {
    const int M = net->dims[1];
    const float *W = net->W_host[0];
    for (int mt = 0; mt < s.mt[0]; ++mt)
      for (int g = 0; g < s.g1; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int step = 4 * g + e;
            const int k = (lane >> 5) * s.bp + step;
            const int i = 32 * mt + (lane & 31);
            const bool ok = step < s.bp && k < s.n_in && i < M;
            *w++ = ok ? scaled(0, W[static_cast<size_t>(k) * M + i]) : 0.f;
          }
  }
  // layers 2..depth (the last one is the output layer; same fragment rule)
  for (int l = 1; l < net->depth; ++l) {
    const int K = net->dims[l], M = net->dims[l + 1];
    const int KT = s.mt[l - 1];
    const int MT = (l == net->depth - 1) ? s.mt_out : s.mt[l];
    const float *W = net->W_host[l];
    for (int mt = 0; mt < MT; ++mt)
      for (int kt = 0; kt < KT; ++kt)
        for (int q = 0; q < 4; ++q)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const int k = 32 * kt + e + 8 * q + 4 * (lane >> 5);
              const int i = 32 * mt + (lane & 31);
              const bool ok = k < K && i < M;
              *w++ = ok ? scaled(l, W[static_cast<size_t>(k) * M + i]) : 0.f;
            }
  }
  // biases, each layer padded to a multiple of 32
  for (int l = 0; l < net->depth; ++l) {
    const int K = net->dims[l], M = net->dims[l + 1];
    const int MT = (l == net->depth - 1) ? s.mt_out : s.mt[l];
    const double sc = l == net->depth - 1 ? 1.0 : kTwoLog2e;
    for (int i = 0; i < 32 * MT; ++i) {
      double b = 0.0;
      if (i < M) {
        b = net->b_host[l][i];
        if (l >= 1)
          for (int k = 0; k < K; ++k) b += net->W_host[l][static_cast<size_t>(k) * M + i];
        b *= sc;
      }
      *w++ = static_cast<float>(b);
    }
  }
}

nnv_status make_params(const nnv_mlp_handle *h, const float *x_dev, int64_t n_rows,
                       DecodeParams *p, LaunchPlan *plan) {
  const PackedShape &s = h->shape;
  const nnv_status dv = nnv::check_current_device(h->device);
  if (dv != NNV_OK) return dv;
  const nnv_status st = nnv::decode_plan(s, n_rows, h->num_cus, h->lds_limit, h->threads, plan);
  if (st != NNV_OK) return st;
  memset(p, 0, sizeof(*p));
  p->x = x_dev;
  p->n_rows = n_rows;
  p->n_tiles = (n_rows + nnv::kRowsPerTile - 1) / nnv::kRowsPerTile;
  p->wpack = h->wpack_dev;
  p->n_in = s.n_in;
  p->bp = s.bp;
  p->xvec = (s.n_in == 2 * s.bp) && (s.bp % 4 == 0) &&
            (reinterpret_cast<uintptr_t>(x_dev) % 16 == 0);
  p->mt_out = s.mt_out;
  p->n_classes = s.n_classes;
  p->out_in_lds = plan->out_in_lds;
  p->out_lds_tiles = plan->out_lds_tiles;
  p->n_bins = 1;
  p->rows_per_block = 1;
  p->first_row = 0;
  return NNV_OK;
}

// K1w (nnv_mlp_split.hip) or K1: pinned by nnv_mlp_set_path, else K1w for short inputs of the shapes it takes
bool use_split(const nnv_mlp_handle *h, int64_t n_tiles) {
  if (h->path == NNV_MLP_PATH_PERSISTENT || !nnv::split_supported(h->shape)) return false;
  return h->path == NNV_MLP_PATH_SPLIT || n_tiles <= nnv::kSplitAutoMaxTiles;
}

}  // namespace

namespace nnv {
nnv_status check_current_device(int device) {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return NNV_ERR_NO_DEVICE;
  return cur == device ? NNV_OK : NNV_ERR_INVALID_ARGUMENT;
}
}  // namespace nnv

extern "C" {

int32_t nnv_abi_version(void) { return NNV_ABI_VERSION; }

int32_t nnv_check_device(int32_t device) { return nnv::check_current_device(device); }

const char *nnv_status_string(int32_t status) {
  switch (status) {
    case NNV_OK: return "ok";
    case NNV_ERR_INVALID_ARGUMENT: return "invalid argument (NULL / bad size / handle packed for another device)";
    case NNV_ERR_UNSUPPORTED_SHAPE: return "unsupported network shape (no kernel instantiation)";
    case NNV_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case NNV_ERR_HIP: return "HIP runtime error";
    case NNV_ERR_OUT_OF_MEMORY: return "out of memory";
    default: return "unknown status";
  }
}

int32_t nnv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && is_gfx950(prop)) ++ok;
  }
  return ok;
}

int32_t nnv_mlp_prepack(const nnv_mlp_desc *net, nnv_mlp_handle **out) {
  return nnv_mlp_prepack_ex(net, NNV_PRECISION_F32, out);
}

int32_t nnv_mlp_prepack_ex(const nnv_mlp_desc *net, int32_t precision, nnv_mlp_handle **out) {
  if (!net || !out) return NNV_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  PackedShape s0;
  const nnv_status st = shape_from_desc(net, precision, &s0);
  if (st != NNV_OK) return st;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NNV_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NNV_ERR_NO_DEVICE;
  if (!is_gfx950(prop)) return NNV_ERR_NO_DEVICE;
  // hidden layers and their biases must fit the LDS (the class layer and ITS biases may be streamed: decode_configure)
  if ((s0.hid_floats + s0.bias_floats - 32LL * s0.mt_out) * 4 > kLdsLimitBytes) return NNV_ERR_UNSUPPORTED_SHAPE;

  std::vector<float> packed;
  if (s0.precision != nnv::kPrecF32) {
    if (!pack_network_f16x3(net, s0, &packed, s0.precision == nnv::kPrecF16x3 ? 2 : 1))
      return NNV_ERR_UNSUPPORTED_SHAPE;  // weight outside f16 range
  } else {
    pack_network(net, s0, &packed);
  }
  nnv_mlp_handle *h = new (std::nothrow) nnv_mlp_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  h->shape = s0;
  for (int l = 0; l <= NNV_MAX_LAYERS; ++l) h->dims[l] = l <= net->depth ? net->dims[l] : 0;
  h->wpack_floats = static_cast<int64_t>(packed.size());
  h->device = dev;
  h->num_cus = prop.multiProcessorCount;
  h->lds_limit = kLdsLimitBytes;
  h->wpack_dev = nullptr;
  h->host_ctx = nullptr;
  h->path = NNV_MLP_PATH_AUTO;
  if (hipMalloc(reinterpret_cast<void **>(&h->wpack_dev), packed.size() * sizeof(float)) != hipSuccess) {
    delete h;
    return NNV_ERR_OUT_OF_MEMORY;
  }
  if (hipMemcpy(h->wpack_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(h->wpack_dev);
    delete h;
    return NNV_ERR_HIP;
  }
  h->threads = nnv::decode_configure(&h->shape, h->lds_limit);
  const PackedShape &s = h->shape;
  const nnv_status prep = nnv::decode_prepare(s, h->num_cus, h->lds_limit, h->threads, dev);
  h->vgprs = nnv::decode_kernel_vgprs(s, h->threads, s.streams == 0 && s.n_classes % 32 == 0);
  if (prep != NNV_OK) {
    (void)hipFree(h->wpack_dev);
    delete h;
    return prep;
  }
  *out = h;
  return NNV_OK;
}

int32_t nnv_mlp_free(nnv_mlp_handle *h) {
  if (!h) return NNV_OK;
  nnv::host_path_release(h);
  const hipError_t e = hipFree(h->wpack_dev);
  delete h;
  return e == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_mlp_decode_f32(const nnv_mlp_handle *h, const float *x_dev, int64_t n_rows,
                           int32_t tile_bits, const int32_t *labels_dev, int32_t *cls32_dev,
                           int64_t *cls64_dev, uint8_t *bits_dev, uint64_t *counters_dev,
                           void *stream) {
  if (!h || n_rows < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (counters_dev && !labels_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (bits_dev || counters_dev) {
    if (tile_bits < 1 || tile_bits > 30) return NNV_ERR_INVALID_ARGUMENT;
    if ((1LL << tile_bits) < h->shape.n_classes) return NNV_ERR_INVALID_ARGUMENT;
  }
  DecodeParams p;
  LaunchPlan plan;
  const nnv_status st = make_params(h, x_dev, n_rows, &p, &plan);
  if (st != NNV_OK) return st;
  p.tile_bits = tile_bits;
  p.labels = labels_dev;
  p.cls32 = cls32_dev;
  p.cls64 = cls64_dev;
  p.bits = bits_dev;
  p.counters = reinterpret_cast<unsigned long long *>(counters_dev);
  if (use_split(h, p.n_tiles)) return nnv::split_launch(h->shape, p, static_cast<hipStream_t>(stream));
  return nnv::decode_launch(h->shape, p, plan, false, h->device, static_cast<hipStream_t>(stream));
}

int32_t nnv_mlp_set_path(nnv_mlp_handle *h, int32_t path) {
  if (!h || (path != NNV_MLP_PATH_AUTO && path != NNV_MLP_PATH_PERSISTENT && path != NNV_MLP_PATH_SPLIT)) return NNV_ERR_INVALID_ARGUMENT;
  if (path == NNV_MLP_PATH_SPLIT && !nnv::split_supported(h->shape)) return NNV_ERR_UNSUPPORTED_SHAPE;
  h->path = path;
  return NNV_OK;
}

int32_t nnv_mlp_path_for(const nnv_mlp_handle *h, int64_t n_rows) {
  if (!h || n_rows < 0) return -NNV_ERR_INVALID_ARGUMENT;
  return use_split(h, (n_rows + nnv::kRowsPerTile - 1) / nnv::kRowsPerTile) ? NNV_MLP_PATH_SPLIT : NNV_MLP_PATH_PERSISTENT;
}

int32_t nnv_mlp_decode_binned_f32(const nnv_mlp_handle *h, const float *x_dev, int64_t n_rows,
                                  int32_t tile_bits, const int32_t *labels_dev, int32_t *cls32_dev,
                                  uint64_t *counters_dev, int64_t first_row, int64_t rows_per_block,
                                  int32_t n_bins, void *stream) {
  if (!h || n_rows < 0 || first_row < 0 || rows_per_block < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (n_bins < 1 || n_bins > nnv::kMaxBins) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_dev || !labels_dev || !counters_dev) return NNV_ERR_INVALID_ARGUMENT;
  if (tile_bits < 1 || tile_bits > 30 || (1LL << tile_bits) < h->shape.n_classes)
    return NNV_ERR_INVALID_ARGUMENT;
  DecodeParams p;
  LaunchPlan plan;
  const nnv_status st = make_params(h, x_dev, n_rows, &p, &plan);
  if (st != NNV_OK) return st;
  p.tile_bits = tile_bits;
  p.labels = labels_dev;
  p.cls32 = cls32_dev;
  p.counters = reinterpret_cast<unsigned long long *>(counters_dev);
  p.n_bins = n_bins;
  p.rows_per_block = rows_per_block;
  p.first_row = first_row;
  return nnv::decode_launch(h->shape, p, plan, false, h->device, static_cast<hipStream_t>(stream));
}

int32_t nnv_mlp_forward_f32(const nnv_mlp_handle *h, const float *x_dev, int64_t n_rows,
                            float *out_dev, void *stream) {
  if (!h || n_rows < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_dev || !out_dev) return NNV_ERR_INVALID_ARGUMENT;
  DecodeParams p;
  LaunchPlan plan;
  const nnv_status st = make_params(h, x_dev, n_rows, &p, &plan);
  if (st != NNV_OK) return st;
  p.act = out_dev;
  return nnv::decode_launch(h->shape, p, plan, true, h->device, static_cast<hipStream_t>(stream));
}

int32_t nnv_awgn_generate_f32(int32_t code, int32_t tile_bits, int64_t n_rows, int64_t first_row,
                              int64_t rows_per_block, const float *sigma_host, int32_t n_sigma,
                              uint64_t seed, float *x_dev, int32_t *labels_dev, void *stream) {
  if (n_rows < 0 || first_row < 0 || rows_per_block < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (!sigma_host || n_sigma < 1 || n_sigma > 16) return NNV_ERR_INVALID_ARGUMENT;
  // the reference encoders index bits[5] / bits[1] unconditionally
  // (src/fnn_viterbi.py:143,201): shorter blocks are an error there too
  const int32_t base_code = code & ~NNV_CODE_CONTINUOUS;
  if (base_code == NNV_CODE_133171) {
    if (tile_bits < 6) return NNV_ERR_INVALID_ARGUMENT;
  } else if (base_code == NNV_CODE_57) {
    if (tile_bits < 2) return NNV_ERR_INVALID_ARGUMENT;
  } else {
    return NNV_ERR_INVALID_ARGUMENT;
  }
  // a continuous stream reaches back 6 (2) bits: they must all lie in the previous row
  if ((code & NNV_CODE_CONTINUOUS) && tile_bits < 6) return NNV_ERR_INVALID_ARGUMENT;
  if (tile_bits > 16) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_dev) return NNV_ERR_INVALID_ARGUMENT;
  return nnv::generate_launch(code, tile_bits, n_rows, first_row, rows_per_block, sigma_host,
                              n_sigma, seed, x_dev, labels_dev, static_cast<hipStream_t>(stream));
}

int32_t nnv_mlp_decode_f32_timed(const nnv_mlp_handle *h, const float *x_dev, int64_t n_rows,
                                 int32_t tile_bits, const int32_t *labels_dev, int32_t *cls32_dev,
                                 uint64_t *counters_dev, void *stream, int32_t iters,
                                 float *elapsed_ms_host) {
  if (!elapsed_ms_host || iters < 1) return NNV_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return NNV_ERR_HIP;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return NNV_ERR_HIP; }
  int32_t st = NNV_OK;
  if (hipEventRecord(e0, s) != hipSuccess) st = NNV_ERR_HIP;
  for (int i = 0; i < iters && st == NNV_OK; ++i)
    st = nnv_mlp_decode_f32(h, x_dev, n_rows, tile_bits, labels_dev, cls32_dev, nullptr, nullptr,
                            counters_dev, stream);
  if (st == NNV_OK && hipEventRecord(e1, s) != hipSuccess) st = NNV_ERR_HIP;
  if (st == NNV_OK && hipEventSynchronize(e1) != hipSuccess) st = NNV_ERR_HIP;
  if (st == NNV_OK && hipEventElapsedTime(elapsed_ms_host, e0, e1) != hipSuccess) st = NNV_ERR_HIP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return st;
}

int32_t nnv_mlp_launch_info(const nnv_mlp_handle *h, int64_t n_rows, int32_t *info_host) {
  if (!h || !info_host || n_rows < 0) return NNV_ERR_INVALID_ARGUMENT;
  LaunchPlan plan;
  const nnv_status st = nnv::decode_plan(h->shape, n_rows, h->num_cus, h->lds_limit, h->threads, &plan);
  if (st != NNV_OK) return st;
  info_host[0] = plan.threads;
  info_host[1] = plan.blocks;
  info_host[2] = plan.lds_bytes;
  info_host[3] = h->vgprs;
  info_host[4] = plan.out_in_lds;
  info_host[5] = nnv::kRowsPerTile;
  info_host[6] = h->shape.depth;
  info_host[7] = h->shape.n_classes;
  info_host[8] = h->shape.precision;
  info_host[9] = plan.out_lds_tiles;
  return NNV_OK;
}

}  // extern "C"
