// C ABI of the layer-at-a-time path (nnv_wide_*): packing, chunking, launches.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "nnv_internal.h"

struct nnv_wide_handle {
  int32_t depth;
  int32_t dims[NNV_WIDE_MAX_LAYERS + 1];
  int32_t kpad[NNV_WIDE_MAX_LAYERS];  // padded input width of layer l (multiple of 32)
  int32_t mpad[NNV_WIDE_MAX_LAYERS];  // padded output width (multiple of bn[l]; class layer: see head)
  int32_t bn[NNV_WIDE_MAX_LAYERS];    // GEMM column tile of layer l: 128, or 32 for narrow fp32 layers
  int32_t head;                       // 1: the class layer runs in head_argmax_kernel (<= 32 classes)
  int32_t act;                        // 1 tanh, 2 relu
  int32_t f16x3;                      // 1: hidden layers' weights are hi/lo f16 planes (w_off = hi, wl_off = lo, in halves)
  int64_t wl_off[NNV_WIDE_MAX_LAYERS];
  float *dev;                         // all packed weights and biases
  int64_t w_off[NNV_WIDE_MAX_LAYERS], b_off[NNV_WIDE_MAX_LAYERS];
  int32_t max_mpad;
  int32_t device;                     // the HIP device the network was packed for
  // K6 (nnv_wide_fused.hip): the whole chain in one persistent launch when the shape allows it
  int32_t fused;                      // 1: fused_dev holds the fragment-order packing
  int32_t num_cus;
  float *fused_dev;
  int32_t path;                       // nnv_wide_path: AUTO (= K6 whenever packed), or what nnv_wide_set_path pinned
  nnv::WideFusedParams fp;            // static part (pointers into fused_dev, shapes); per-call fields filled at launch
};

namespace {

int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline uint16_t f2h(float f) { return nnv::f32_to_f16(f); }
inline float h2f(uint16_t h) { return nnv::f16_to_f32(h); }

// first-maximum argmax over the first n_classes entries of each row (class counts > 32)
__global__ __launch_bounds__(256) void row_argmax_kernel(const float *A, int64_t lda, int64_t n_rows,
                                                         int n_classes, int32_t *cls32, int64_t *cls64,
                                                         float *act) {
  const int lane = threadIdx.x & 31;
  const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  if (row >= n_rows) return;
  const float *a = A + row * lda;
  float best = -__builtin_inff();
  int bidx = 0x7fffffff;
  for (int c = lane; c < n_classes; c += 32) {
    const float v = a[c];
    if (act) act[row * n_classes + c] = v;
    if (v > best) {
      best = v;
      bidx = c;
    }
  }
  for (int off = 16; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 32);
    const int oi = __shfl_xor(bidx, off, 32);
    if (ov > best || (ov == best && oi < bidx)) {
      best = ov;
      bidx = oi;
    }
  }
  if (lane == 0) {
    if (cls32) cls32[row] = bidx;
    if (cls64) cls64[row] = bidx;
  }
}

// K6 eligibility + packing.  MFMA layer l: neuron tile t, fragment group g, lane, element e holds
// W[8g + 4 (lane >> 5) + e][32 t + (lane & 31)] (zero outside the matrix): one wave-load = one contiguous 1 KB,
// the A operand of four consecutive v_mfma_f32_32x32x2_f32.  A class layer of 33..512 classes is packed as one
// more MFMA layer (wide_head); up to 32 classes it stays a [class][input] matrix for the vector-ALU head.
// Not eligible (fused stays 0, the layer-at-a-time path runs): f16x3, any width or the row length above 512.
nnv_status pack_fused(const nnv_wide_desc *net, nnv_wide_handle *h) {
  h->fused = 0;
  h->fused_dev = nullptr;
  const int nh = net->depth - 1;
  const int n_classes = net->dims[net->depth];
  const bool wide_head = n_classes > 32;
  const int nm = nh + (wide_head ? 1 : 0);        // layers on the matrix cores
  if (h->f16x3 || nh < 1 || nm > nnv::kFusedMaxHidden) return NNV_OK;
  for (int l = 0; l <= net->depth; ++l)
    if (net->dims[l] > nnv::kFusedMaxWidth) return NNV_OK;
  nnv::WideFusedParams &fp = h->fp;
  memset(&fp, 0, sizeof(fp));
  fp.n_hidden = nm;
  fp.wide_head = wide_head ? 1 : 0;
  fp.K0 = net->dims[0];
  int64_t w_off[nnv::kFusedMaxHidden], b_off[nnv::kFusedMaxHidden], total = 0;
  int act_quads = 0;
  for (int l = 0; l < nm; ++l) {
    fp.groups[l] = (net->dims[l] + 7) / 8;
    fp.tiles[l] = (net->dims[l + 1] + 31) / 32;
    if (2 * fp.groups[l] > act_quads) act_quads = 2 * fp.groups[l];   // what the layer reads
    if (8 * fp.tiles[l] > act_quads) act_quads = 8 * fp.tiles[l];     // what it writes
    w_off[l] = total;
    total += static_cast<int64_t>(fp.tiles[l]) * fp.groups[l] * 256;
    b_off[l] = total;
    total += 32LL * fp.tiles[l];
  }
  fp.act_quads = act_quads;
  fp.head_quads = 8 * fp.tiles[nm - 1];            // narrow head: its inputs; wide head: the class quads
  fp.n_classes = n_classes;
  const int64_t hw_off = total;
  if (!wide_head) total += static_cast<int64_t>(n_classes) * fp.head_quads * 4;
  const int64_t hb_off = total;
  if (!wide_head) total += round_up(n_classes, 4);
  fp.waves = nnv::wide_fused_pick_waves(fp);
  if (nnv::wide_fused_lds_bytes(fp) > 160 * 1024) return NNV_OK;
  std::vector<float> host(static_cast<size_t>(total), 0.f);
  for (int l = 0; l < nm; ++l) {
    const int K = net->dims[l], M = net->dims[l + 1];
    for (int t = 0; t < fp.tiles[l]; ++t)
      for (int g = 0; g < fp.groups[l]; ++g) {
        float *grp = host.data() + w_off[l] + (static_cast<int64_t>(t) * fp.groups[l] + g) * 256;
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int k = 8 * g + 4 * (lane >> 5) + e, m = 32 * t + (lane & 31);
            if (k < K && m < M) grp[lane * 4 + e] = net->W_host[l][static_cast<size_t>(k) * M + m];
          }
      }
    for (int m = 0; m < M; ++m) host[b_off[l] + m] = net->b_host[l][m];
  }
  if (!wide_head) {
    const int K = net->dims[nh], C = n_classes;
    for (int c = 0; c < C; ++c)
      for (int k = 0; k < K; ++k)
        host[hw_off + static_cast<int64_t>(c) * fp.head_quads * 4 + k] = net->W_host[nh][static_cast<size_t>(k) * C + c];
    for (int c = 0; c < C; ++c) host[hb_off + c] = net->b_host[nh][c];
  }
  // K6 is an OPTION on top of a handle the layered path already serves (as K4c / K4d are for the bank): if its
  // second copy of the weights cannot be allocated or uploaded the handle stays on the layered path
  if (hipMalloc(reinterpret_cast<void **>(&h->fused_dev), host.size() * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();   // clear the sticky error: this is not a failure of the call
    h->fused_dev = nullptr;
    return NNV_OK;
  }
  if (hipMemcpy(h->fused_dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(h->fused_dev);
    h->fused_dev = nullptr;
    return NNV_OK;
  }
  for (int l = 0; l < nm; ++l) {
    fp.w[l] = h->fused_dev + w_off[l];
    fp.bias[l] = h->fused_dev + b_off[l];
  }
  if (!wide_head) {
    fp.head_w = h->fused_dev + hw_off;
    fp.head_b = h->fused_dev + hb_off;
  }
  {
    const nnv_status e = nnv::wide_fused_prepare(fp, h->act, h->device);
    if (e != NNV_OK) {
      (void)hipGetLastError();
      (void)hipFree(h->fused_dev);
      h->fused_dev = nullptr;
      return NNV_OK;
    }
  }
  h->fused = 1;
  return NNV_OK;
}

}  // namespace

extern "C" {

int32_t nnv_wide_prepack(const nnv_wide_desc *net, nnv_wide_handle **out) {
  if (!net || !out) return NNV_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (net->depth < 2 || net->depth > NNV_WIDE_MAX_LAYERS) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (net->activation != NNV_ACT_TANH && net->activation != NNV_ACT_RELU) return NNV_ERR_INVALID_ARGUMENT;
  for (int l = 0; l <= net->depth; ++l)
    if (net->dims[l] < 1) return NNV_ERR_INVALID_ARGUMENT;
  // widths: NNV_WIDE_MAX_WIDTH per layer, NNV_WIDE_MAX_CLASSES for the class layer (2^16: the reference's
  // DecoderConfig goes to block_length 15 = 32768 classes in its own tests, tests/test_utilities.py:342-348)
  for (int l = 0; l < net->depth; ++l)
    if (net->dims[l] > NNV_WIDE_MAX_WIDTH) return NNV_ERR_UNSUPPORTED_SHAPE;
  if (net->dims[net->depth] > NNV_WIDE_MAX_CLASSES) return NNV_ERR_UNSUPPORTED_SHAPE;
  for (int l = 0; l < net->depth; ++l)
    if (!net->W_host[l] || !net->b_host[l]) return NNV_ERR_INVALID_ARGUMENT;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return NNV_ERR_NO_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return NNV_ERR_NO_DEVICE;

  nnv_wide_handle *h = new (std::nothrow) nnv_wide_handle;
  if (!h) return NNV_ERR_OUT_OF_MEMORY;
  memset(h, 0, sizeof(*h));
  h->device = dev;
  h->depth = net->depth;
  h->act = net->activation == NNV_ACT_RELU ? 2 : 1;
  for (int l = 0; l <= net->depth; ++l) h->dims[l] = net->dims[l];
  h->head = net->dims[net->depth] <= 32 ? 1 : 0;
  h->f16x3 = net->precision == NNV_PRECISION_F16X3 ? 1 : 0;
  if (net->precision != NNV_PRECISION_F32 && net->precision != NNV_PRECISION_F16X3) {
    delete h;
    return NNV_ERR_INVALID_ARGUMENT;
  }
  if (h->f16x3 && !h->head) {  // the split path ends in the narrow head kernel
    delete h;
    return NNV_ERR_UNSUPPORTED_SHAPE;
  }
  int64_t total = 0;
  for (int l = 0; l < net->depth; ++l) {
    const bool is_head = h->head && l == net->depth - 1;
    // layer l reads the previous layer's padded output (or the raw rows for l = 0)
    h->kpad[l] = l == 0 ? round_up(net->dims[0], nnv::kDenseBK) : h->mpad[l - 1];
    // narrow fp32 layers (<= 64 outputs, e.g. the bit-wise decoder's 30) use 32-column tiles
    h->bn[l] = (!h->f16x3 && net->dims[l + 1] <= 64) ? 32 : nnv::kDenseBN;
    h->mpad[l] = is_head ? net->dims[l + 1] : round_up(net->dims[l + 1], h->bn[l]);
    if (!is_head && h->mpad[l] > h->max_mpad) h->max_mpad = h->mpad[l];
    h->w_off[l] = total;   // fp32: [mpad][kpad] floats; f16x3 hidden layer: hi plane then lo plane (same bytes)
    total += static_cast<int64_t>(h->mpad[l]) * h->kpad[l];
    h->b_off[l] = total;
    total += round_up(h->mpad[l], 4);
  }
  std::vector<float> host(static_cast<size_t>(total), 0.f);
  for (int l = 0; l < net->depth; ++l) {
    const int K = net->dims[l], M = net->dims[l + 1];
    float *wt = host.data() + h->w_off[l];
    const bool planes = h->f16x3 && !(h->head && l == net->depth - 1);
    if (planes) {
      uint16_t *hi = reinterpret_cast<uint16_t *>(wt);
      uint16_t *lo = hi + static_cast<size_t>(h->mpad[l]) * h->kpad[l];
      for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
          const float v = net->W_host[l][static_cast<size_t>(k) * M + m];
          if (!(v > -65504.f && v < 65504.f)) {
            delete h;
            return NNV_ERR_UNSUPPORTED_SHAPE;
          }
          const uint16_t hh = f2h(v);
          hi[static_cast<size_t>(m) * h->kpad[l] + k] = hh;
          lo[static_cast<size_t>(m) * h->kpad[l] + k] = f2h(v - h2f(hh));
        }
    } else {
      for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) wt[static_cast<size_t>(m) * h->kpad[l] + k] = net->W_host[l][static_cast<size_t>(k) * M + m];
    }
    for (int m = 0; m < M; ++m) host[h->b_off[l] + m] = net->b_host[l][m];
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
  h->num_cus = prop.multiProcessorCount;
  h->path = NNV_WIDE_PATH_AUTO;
  (void)pack_fused(net, h);   // optional: on any failure h->fused stays 0 and the layered path serves the handle
  *out = h;
  return NNV_OK;
}

int32_t nnv_wide_free(nnv_wide_handle *h) {
  if (!h) return NNV_OK;
  const hipError_t e = hipFree(h->dev);
  const hipError_t e2 = h->fused_dev ? hipFree(h->fused_dev) : hipSuccess;
  delete h;
  return (e == hipSuccess && e2 == hipSuccess) ? NNV_OK : NNV_ERR_HIP;
}

int32_t nnv_wide_set_path(nnv_wide_handle *h, int32_t path) {
  if (!h) return NNV_ERR_INVALID_ARGUMENT;
  if (path != NNV_WIDE_PATH_AUTO && path != NNV_WIDE_PATH_FUSED && path != NNV_WIDE_PATH_LAYERED)
    return NNV_ERR_INVALID_ARGUMENT;
  if (path == NNV_WIDE_PATH_FUSED && !h->fused) return NNV_ERR_UNSUPPORTED_SHAPE;
  h->path = path;
  return NNV_OK;
}

int32_t nnv_wide_get_path(const nnv_wide_handle *h) {
  if (!h) return -NNV_ERR_INVALID_ARGUMENT;
  return (h->fused && h->path != NNV_WIDE_PATH_LAYERED) ? NNV_WIDE_PATH_FUSED : NNV_WIDE_PATH_LAYERED;
}

int64_t nnv_wide_workspace_bytes(const nnv_wide_handle *h, int64_t chunk_rows) {
  if (!h || chunk_rows < 1) return -1;
  const int64_t rows = (chunk_rows + 127) / 128 * 128;
  return 2 * rows * h->max_mpad * static_cast<int64_t>(sizeof(float));
}

int32_t nnv_wide_decode_f32(const nnv_wide_handle *h, const float *x_dev, int64_t n_rows,
                            int64_t row_stride, void *workspace_dev, int64_t workspace_bytes,
                            int32_t *cls32_dev, int64_t *cls64_dev, float *act_dev, void *stream) {
  if (!h || n_rows < 0 || row_stride < 1) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_dev || !workspace_dev) return NNV_ERR_INVALID_ARGUMENT;
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  if (reinterpret_cast<uintptr_t>(x_dev) % 8 != 0 || (row_stride % 2) != 0) return NNV_ERR_INVALID_ARGUMENT;
  const int64_t per_row = 2LL * h->max_mpad * static_cast<int64_t>(sizeof(float));
  int64_t chunk = workspace_bytes / per_row / 128 * 128;
  if (chunk < 128) return NNV_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // K6: one persistent launch for the whole chain (no workspace traffic), unless nnv_wide_set_path pinned the
  // handle to the layer-at-a-time path (parity tests hold the two against each other)
  if (nnv_wide_get_path(h) == NNV_WIDE_PATH_FUSED) {
    nnv::WideFusedParams fp = h->fp;
    fp.x = x_dev;
    fp.row_stride = row_stride;
    fp.n_rows = n_rows;
    fp.n_units = (n_rows + 31) / 32;
    fp.x16 = (row_stride % 4 == 0 && reinterpret_cast<uintptr_t>(x_dev) % 16 == 0) ? 1 : 0;
    fp.cls32 = cls32_dev;
    fp.cls64 = cls64_dev;
    fp.act = act_dev;
    return nnv::wide_fused_launch(fp, h->act, h->num_cus, h->device, s);
  }
  float *buf[2] = {static_cast<float *>(workspace_dev),
                   static_cast<float *>(workspace_dev) + chunk * h->max_mpad};
  const int last = h->depth - 1;
  for (int64_t done = 0; done < n_rows; done += chunk) {
    const int64_t rows = n_rows - done < chunk ? n_rows - done : chunk;
    const float *in = x_dev + done * row_stride;
    int64_t in_stride = row_stride;
    int cur = 0;
    for (int l = 0; l < h->depth; ++l) {
      if (h->f16x3) {
        // activations as hi/lo planes: buffer = [chunk][max_mpad] halves hi, then the same for lo
        _Float16 *out_hi = reinterpret_cast<_Float16 *>(buf[cur]);
        _Float16 *out_lo = out_hi + chunk * h->max_mpad;
        const _Float16 *in_hi = reinterpret_cast<const _Float16 *>(in);
        const _Float16 *in_lo = in_hi + chunk * h->max_mpad;
        if (l == last) {
          nnv::HeadF16Params hp;
          hp.A_hi = in_hi;
          hp.A_lo = in_lo;
          hp.lda = in_stride;
          hp.n_rows = rows;
          hp.Kpad = h->kpad[l];
          hp.n_classes = h->dims[h->depth];
          hp.Wt = h->dev + h->w_off[l];
          hp.bias = h->dev + h->b_off[l];
          hp.cls32 = cls32_dev ? cls32_dev + done : nullptr;
          hp.cls64 = cls64_dev ? cls64_dev + done : nullptr;
          hp.act = act_dev ? act_dev + done * hp.n_classes : nullptr;
          hp.act_fn = h->act;
          const nnv_status st = nnv::head_f16x3_launch(hp, s);
          if (st != NNV_OK) return st;
          break;
        }
        nnv::DenseF16Params dp;
        dp.A_f32 = in;
        dp.a_row_stride = in_stride;
        dp.A_hi = in_hi;
        dp.A_lo = in_lo;
        dp.lda = in_stride;
        dp.n_rows = rows;
        dp.K = h->dims[l];
        dp.Kpad = h->kpad[l];
        dp.W_hi = reinterpret_cast<const _Float16 *>(h->dev + h->w_off[l]);
        dp.W_lo = dp.W_hi + static_cast<int64_t>(h->mpad[l]) * h->kpad[l];
        dp.bias = h->dev + h->b_off[l];
        dp.C_hi = out_hi;
        dp.C_lo = out_lo;
        dp.ldc = h->mpad[l];
        dp.nblk_m = h->mpad[l] / nnv::kDenseBN;
        dp.nblk_rows = (rows + nnv::kDenseBM - 1) / nnv::kDenseBM;
        dp.act = h->act;
        const nnv_status st = nnv::dense_f16x3_launch(dp, l == 0, s);
        if (st != NNV_OK) return st;
        in = buf[cur];
        in_stride = h->mpad[l];
        cur ^= 1;
        continue;
      }
      if (l == last && h->head) {
        nnv::HeadParams hp;
        hp.A = in;
        hp.lda = in_stride;
        hp.n_rows = rows;
        hp.Kpad = h->kpad[l];
        hp.n_classes = h->dims[h->depth];
        hp.Wt = h->dev + h->w_off[l];
        hp.bias = h->dev + h->b_off[l];
        hp.cls32 = cls32_dev ? cls32_dev + done : nullptr;
        hp.cls64 = cls64_dev ? cls64_dev + done : nullptr;
        hp.act = act_dev ? act_dev + done * hp.n_classes : nullptr;
        hp.act_fn = h->act;
        const nnv_status st = nnv::head_launch(hp, s);
        if (st != NNV_OK) return st;
        break;
      }
      nnv::DenseParams dp{};
      dp.A = in;
      dp.a_row_stride = in_stride;
      dp.a16 = (l == 0 && in_stride % 4 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0) ? 1 : 0;
      dp.n_rows = rows;
      dp.K = h->dims[l];
      dp.Kpad = h->kpad[l];
      dp.Wt = h->dev + h->w_off[l];
      dp.bias = h->dev + h->b_off[l];
      dp.C = buf[cur];
      dp.ldc = h->mpad[l];
      dp.bn = h->bn[l];
      dp.nblk_m = h->mpad[l] / h->bn[l];
      dp.nblk_rows = (rows + nnv::kDenseBM - 1) / nnv::kDenseBM;
      dp.act = h->act;
      if (l == last && !act_dev && h->bn[l] == 128 && l > 0) {
        // wide class layer: argmax in the GEMM epilogue, one (value, class) pair per (row, 128-column block) instead
        // of the [rows, n_classes] matrix (the pairs reuse the buffer the matrix would have taken: 8 bytes per block
        // against its 512)
        dp.C = nullptr;
        dp.n_valid = h->dims[h->depth];
        dp.part_val = buf[cur];
        dp.part_idx = reinterpret_cast<int32_t *>(buf[cur] + chunk * dp.nblk_m);
        nnv_status st = nnv::dense_launch(dp, false, s, true);
        if (st != NNV_OK) return st;
        st = nnv::partial_argmax_launch(dp.part_val, dp.part_idx, rows, dp.nblk_m, cls32_dev ? cls32_dev + done : nullptr,
                                        cls64_dev ? cls64_dev + done : nullptr, s);
        if (st != NNV_OK) return st;
        break;
      }
      const nnv_status st = nnv::dense_launch(dp, l == 0, s);
      if (st != NNV_OK) return st;
      in = buf[cur];
      in_stride = h->mpad[l];
      cur ^= 1;
      if (l == last) {  // wide class layer with the debug output wanted (or a 33..64-class layer on the 32-column tile): argmax over the stored row
        const int64_t blocks = (rows * 32 + 255) / 256;
        hipLaunchKernelGGL(row_argmax_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, in,
                           in_stride, rows, h->dims[h->depth], cls32_dev ? cls32_dev + done : nullptr,
                           cls64_dev ? cls64_dev + done : nullptr,
                           act_dev ? act_dev + done * h->dims[h->depth] : nullptr);
        if (hipGetLastError() != hipSuccess) return NNV_ERR_HIP;
      }
    }
  }
  return NNV_OK;
}

}  // extern "C"
