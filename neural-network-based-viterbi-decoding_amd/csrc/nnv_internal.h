// Internal declarations shared by the translation units of libnnviterbi_hip.so.
// Not part of the public ABI (that is include/nnviterbi.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "nnviterbi.h"

namespace nnv {

// A/B scaffolding.  The shipped library reads NO environment variable and has no alternative code paths behind
// compile-time switches: the kernel a call gets is a function of the handle and the call's arguments alone
// (include/nnviterbi.h: "no global mutable state").  tools/build_variant.sh builds side libraries with
// -DNNV_AB_BUILD, and only there do the tuning overrides below (workgroup size, band policy, tile sizes ...) exist.
#ifdef NNV_AB_BUILD
inline const char *ab_env(const char *name) { return getenv(name); }
#else
constexpr const char *ab_env(const char *) { return nullptr; }
#endif

// IEEE binary16 <-> binary32 on the host (round-to-nearest-even), used by the packers of the
// hi/lo f16 operand planes (nnv_capi.hip, nnv_wide.hip).
inline uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
  if (x >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);  // rounds to inf
  if (x < 0x33000001u) return static_cast<uint16_t>(sign);              // rounds to 0
  if (x < 0x38800000u) {                                                // subnormal half
    const int shift = 126 - static_cast<int>(x >> 23);                  // 14..24
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    m >>= shift;
    if (rem > halfway || (rem == halfway && (m & 1u))) ++m;
    return static_cast<uint16_t>(sign | m);
  }
  uint32_t h = ((x >> 23) - 112u) << 10 | ((x >> 13) & 0x3ffu);
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return static_cast<uint16_t>(sign | h);
}

inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      e = 113;
      while (!(m & 0x400u)) { m <<= 1; --e; }
      x = sign | (e << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}

// ---------------------------------------------------------------------------
// Packed network layout (built on the host by prepack, read by the kernel)
//
// One MFMA v_mfma_f32_32x32x2_f32 step consumes, per lane l, ONE fp32 of the
// A operand: A[i = l & 31][k = l >> 5].  We compute the transposed problem
//     H_l^T [neurons x rows] = W_l^T [neurons x K] * H_{l-1}^T [K x rows]
// so A fragments are weights and the B operand is the previous layer's
// accumulator register as it stands (see nnv_mlp_decode.hip).
//
// A "fragment group" is 64 lanes x 4 consecutive steps = 256 floats, stored
// [lane][4] so that one ds_read_b128 / global_load_dwordx4 per lane fetches
// the A operands of 4 MFMAs.
//
//  layer 1   groups [mt][g]      g < G1      step s = 4g+e uses input
//                                            k = (l>>5)*bp + s   (bp = ceil(n_in/2))
//  layer >=2 groups [mt][kt][q]  q < 4       step r = 4q+e uses input neuron
//                                            k = 32kt + e + 8q + 4(l>>5)
//            (exactly the neuron that register r of M-tile kt of the previous
//             layer's 32x32 accumulator holds in that lane half)
//  value     W[k][32mt + (l&31)]  (0 where k or the neuron is padding)
//
// wpack = [hidden layers' groups][output layer's groups][biases, each layer
// padded to 32*MT floats]
// ---------------------------------------------------------------------------
// f16x3 packing (NNV_PRECISION_F16X3): the same group size in bytes, holding
// [lane][8 halves]: one group = the hi (or lo) f16 halves of one 16-wide k-step.
//  layer 1   groups [mt][ks][hi|lo]       k-slot (lane>>5, e) <-> input (lane>>5)*bp + 8ks + e
//  layer >=2 groups [mt][kt][s][hi|lo]    k-slot (lane>>5, e) <-> neuron 32kt + ((8s+e)&3)
//                                         + 8((8s+e)>>2) + 4(lane>>5)  (accumulator register 8s+e)
constexpr int kPrecF32 = 0;
constexpr int kPrecF16x3 = 1;
constexpr int kPrecF16 = 2;   // single-product f16: the hi planes of the f16x3 layout only
// In every precision the hidden activations travel as r = 1/(1+2^zeta) with tanh's affine map 1 - 2r folded
// into the next layer's packed weights: kernel (act16) and packers (pack_network*) agree on that.
constexpr int kFragGroupFloats = 256;
constexpr int kRowsPerTile = 32;  // batch rows per wave tile (MFMA N dim)

struct PackedShape {
  int32_t precision;    // kPrecF32 | kPrecF16x3 | kPrecF16
  int32_t depth;        // dense layers incl. output
  int32_t nhid;         // depth - 1
  int32_t n_in;         // dims[0]
  int32_t bp;           // ceil(n_in / 2): layer-1 K steps actually used
  int32_t g1;           // fp32: ceil(bp / 4) layer-1 groups per M-tile; 16-bit modes: 2 * ceil(bp / 8) (kernel template G1)
  int32_t mt[3];        // hidden layers' M-tiles (width padded to 32)
  int32_t n_classes;    // dims[depth]
  int32_t mt_out;       // ceil(n_classes / 32)
  int32_t kt_out;       // mt[nhid-1]
  int64_t hid_floats;   // floats of all hidden-layer groups
  int64_t out_floats;   // floats of the output-layer groups
  int64_t bias_floats;  // 32 * (sum mt + mt_out)
  int32_t streams;      // 1: the class layer does not fit the LDS beside the rest (streaming kernel)
};

struct DecodeParams {
  const float *x;
  int64_t n_rows;
  int64_t n_tiles;
  const float *wpack;  // device
  int32_t n_in;
  int32_t bp;
  int32_t xvec;        // 1: 16-byte vector loads of x are legal
  int32_t mt_out;
  int32_t n_classes;
  int32_t out_in_lds;  // ALL output-layer groups are LDS resident
  int32_t out_lds_tiles;  // class-layer M-tiles resident in LDS (every wave streams the rest from L2 for its own rows)
  int32_t tile_bits;
  const int32_t *labels;
  int32_t *cls32;
  int64_t *cls64;
  uint8_t *bits;
  float *act;
  unsigned long long *counters;  // [4] or, with n_bins > 1, [n_bins][4]
  // mixed-SNR sweeps (BASELINE config 4): counters are kept per SNR bin,
  // bin = ((first_row + row) / rows_per_block) % n_bins
  int32_t n_bins;
  int64_t rows_per_block;
  int64_t first_row;
};

// K1 variants (template parameter VARIANT of mlp_decode_kernel; see nnv_mlp_decode.hip)
constexpr int kVarGeneral = 0, kVarStream = 1, kVarLean = 2, kVarLeanBins = 3;
// The running argmax winner's 16 raw accumulators are parked in a per-lane LDS slot by exec-masked
// ds_write_b128 (no VALU) instead of 16 v_cndmask per class tile: 64 bytes per thread, reserved by decode_plan.
constexpr int kKeepSlotBytes = 64;

// KB of packed weights per class M-tile: fp32 / f16x3 4 per k-tile of 32 inputs, f16 2
constexpr int class_tile_kb(int prec, int kt_out) { return (prec == kPrecF16 ? 2 : 4) * kt_out; }

constexpr int kMaxBins = 16;
constexpr int kLdsTailBytes = 16 + kMaxBins * 16;  // work counter + per-bin {bit, frame, rows, pad}

struct LaunchPlan {
  int32_t threads;
  int32_t blocks;
  int32_t lds_bytes;
  int32_t out_in_lds;
  int32_t out_lds_tiles;
};

// one instantiation of nnv::mlp_decode_kernel (tables in nnv_mlp_decode.hip, lookup in nnv_mlp_launch.hip)
using KernelFn = void (*)(const DecodeParams);
struct KernelEntry {
  int prec, g1, nhid, mta, mtb, mtc, nt;
  bool act;
  int variant;
  KernelFn fn;
};

// nnv_capi.hip: NNV_OK iff `device` is the calling thread's current HIP device.  Every launch entry
// checks its handle's device first: a handle's packed network, kernel attributes and pointers belong to
// the device it was packed for, and launching them on another one is a fault, not an error code.
nnv_status check_current_device(int device);
constexpr int kMaxDevices = 64;  // per-device kernel-attribute cache (one process may drive several GPUs)

// nnv_mlp_decode.hip
nnv_status decode_supported(const PackedShape &s);
// threads: the workgroup size picked ONCE at prepack (decode_pick_threads) and kept in the handle, so a
// launch is table lookups + hipLaunchKernel, without runtime attribute queries.
int decode_pick_threads(const PackedShape &s);
// decides s->streams (class layer LDS resident, or partly streamed from L2 by every wave) and returns the workgroup size
int decode_configure(PackedShape *s, int lds_limit);
nnv_status decode_plan(const PackedShape &s, int64_t n_rows, int num_cus,
                       int lds_limit, int threads, LaunchPlan *plan);
nnv_status decode_launch(const PackedShape &s, const DecodeParams &p,
                         const LaunchPlan &plan, bool write_act, int device,
                         hipStream_t stream);
int decode_kernel_vgprs(const PackedShape &s, int threads, bool lean);
nnv_status decode_prepare(const PackedShape &s, int num_cus, int lds_limit, int threads, int device);

// nnv_dense.hip: layer-at-a-time path for wide networks
constexpr int kDenseBM = 128, kDenseBN = 128, kDenseBK = 32;
struct DenseParams {
  const float *A;        // activations [n_rows][a_row_stride] or, gathered, stream + row * a_row_stride
  int64_t a_row_stride;  // floats between consecutive rows
  int64_t n_rows;
  int32_t a16;           // gather mode: every row start is 16-byte aligned (one dwordx4 per chunk instead of two dwordx2)
  int32_t K;             // valid inputs (gather mode zero-fills k >= K)
  int32_t Kpad;          // multiple of 32; Wt rows have this length
  const float *Wt;       // [Mpad][Kpad] fp32, zero padded
  const float *bias;     // [Mpad]
  float *C;              // [n_rows][ldc]
  int64_t ldc;
  int32_t nblk_m;        // Mpad / bn
  int64_t nblk_rows;     // ceil(n_rows / 128)
  int32_t act;           // 1 tanh, 2 relu
  int32_t bn;            // output columns per block: 128, or 32 for narrow layers
  int32_t col_group;     // bitwise_launch: column blocks per L2-resident W slab (0: row-block-major tile order)
  // fused per-position head (bitwise_launch only): column block c holds positions (bn/32) c ..
  int64_t n_pos;               // bit positions (networks) in the bank
  const float *head_w;         // [n_pos][32][2] fp32, rows >= n_hidden zero
  const float *head_b;         // [n_pos][2]
  uint8_t *bits;               // [n_pos][n_rows] decided bit
  float *head_act;             // nullable [n_pos][n_rows][2] ReLU'd outputs (parity)
  const uint8_t *labels;       // nullable [n_pos][n_rows] true bit
  unsigned long long *errors;  // nullable [n_pos] wrong decisions, ACCUMULATED (needs labels)
  // fused class-layer argmax (dense_launch with cls = true): per (row, column block) the first maximum of the
  // activated outputs over the block's valid columns; C is not written
  int32_t n_valid;             // true class count (columns >= n_valid are padding and never win)
  float *part_val;             // [n_rows][nblk_m]
  int32_t *part_idx;           // [n_rows][nblk_m] global class index of that maximum
};
struct HeadParams {
  const float *A;        // [n_rows][lda]
  int64_t lda;
  int64_t n_rows;
  int32_t Kpad;
  int32_t n_classes;     // <= 32
  const float *Wt;       // [n_classes][Kpad]
  const float *bias;
  int32_t *cls32;
  int64_t *cls64;
  float *act;            // nullable [n_rows][n_classes]
  int32_t act_fn;        // 1 tanh, 2 relu
};
// nnv_dense_f16x3.hip: the same layer with hi/lo f16 operand planes (3 f16 MFMAs per product block)
struct DenseF16Params {
  const float *A_f32;     // gather mode (layer 1): fp32 rows at A_f32 + row * a_row_stride
  int64_t a_row_stride;
  const _Float16 *A_hi;   // otherwise: activation planes [n_rows][lda]
  const _Float16 *A_lo;
  int64_t lda;
  int64_t n_rows;
  int32_t K, Kpad;
  const _Float16 *W_hi;   // [Mpad][Kpad] planes of W^T
  const _Float16 *W_lo;
  const float *bias;
  _Float16 *C_hi;         // [n_rows][ldc]
  _Float16 *C_lo;
  int64_t ldc;
  int32_t nblk_m;
  int64_t nblk_rows;
  int32_t act;            // 1 tanh, 2 relu
  // fused per-position head (bitwise_f16x3_launch only; same meaning as in DenseParams)
  int32_t col_group;
  int64_t n_pos;
  const float *head_w;
  const float *head_b;
  uint8_t *bits;
  float *head_act;
  const uint8_t *labels;
  unsigned long long *errors;
};
struct HeadF16Params {
  const _Float16 *A_hi, *A_lo;
  int64_t lda;
  int64_t n_rows;
  int32_t Kpad, n_classes;
  const float *Wt;
  const float *bias;
  int32_t *cls32;
  int64_t *cls64;
  float *act;
  int32_t act_fn;
};
nnv_status dense_f16x3_launch(const DenseF16Params &p, bool gather, hipStream_t stream);
nnv_status head_f16x3_launch(const HeadF16Params &p, hipStream_t stream);
nnv_status dense_launch(const DenseParams &p, bool gather, hipStream_t stream, bool cls = false);
// first maximum over the nblk partial (value, class) pairs of every row (ties: the lower class index)
nnv_status partial_argmax_launch(const float *part_val, const int32_t *part_idx, int64_t n_rows, int32_t nblk,
                                 int32_t *cls32, int64_t *cls64, hipStream_t stream);
nnv_status bitwise_launch(const DenseParams &p, hipStream_t stream);
nnv_status bitwise_f16x3_launch(const DenseF16Params &p, hipStream_t stream);
nnv_status head_launch(const HeadParams &p, hipStream_t stream);

// nnv_wide_fused.hip (K6): the whole wide network in one persistent launch, activations resident in LDS.
// Eligible: fp32, every layer width -- hidden AND class layer -- and the row length <= 512 (16 waves x one
// 32-neuron tile).  Up to 32 classes the class layer runs on the vector ALU; 33..512 classes make it one more
// MFMA layer followed by an argmax over the LDS buffer (wide_head).
constexpr int kFusedRows = 64, kFusedThreads = 1024, kFusedMaxHidden = 8 /* MFMA layers */, kFusedMaxWidth = 512;
struct WideFusedParams {
  const float *x;          // row i starts at x + i * row_stride
  int64_t row_stride;
  int64_t n_rows;
  int64_t n_units;         // ceil(n_rows / 32): 32-row work units
  int32_t x16;             // every row start 16-byte aligned
  int32_t K0;              // true row length
  int32_t n_hidden;        // layers on the matrix cores (the class layer included when wide_head)
  int32_t wide_head;       // 1: the last MFMA layer is the class layer (33..512 classes), argmax over the LDS buffer
  int32_t waves;           // 16 (one workgroup per CU), 8 (two per CU: every layer <= 8 tiles, LDS <= 80 KB) or 4 (four per CU)
  int32_t groups[kFusedMaxHidden];     // ceil(K_l / 8): fragment groups (8 inputs) per neuron tile
  int32_t tiles[kFusedMaxHidden];      // ceil(M_l / 32): neuron tiles = active waves
  const float *w[kFusedMaxHidden];     // [tile][group][lane 64][4]: W[8g + 4 (lane >> 5) + e][32 tile + (lane & 31)]
  const float *bias[kFusedMaxHidden];  // [32 tiles_l], zero padded
  int32_t act_quads;       // LDS activation buffer, in k-quads of 64 rows x 16 bytes
  int32_t head_quads;      // narrow head: k-quads it reads (8 x tiles of the last hidden layer); wide head: class quads (8 x its tiles)
  int32_t n_classes;       // <= 32 (narrow head) or <= 512 (wide head)
  const float *head_w;     // narrow head only: [n_classes][4 head_quads], zero padded
  const float *head_b;     // narrow head only: [n_classes]
  int32_t *cls32;          // nullable [n_rows]
  int64_t *cls64;          // nullable [n_rows]
  float *act;              // nullable [n_rows][n_classes]
};
int64_t wide_fused_lds_bytes(const WideFusedParams &p);   // needs p.waves
int wide_fused_pick_waves(const WideFusedParams &p);
nnv_status wide_fused_prepare(const WideFusedParams &p, int act, int device);
nnv_status wide_fused_launch(const WideFusedParams &p, int act, int num_cus, int device, hipStream_t stream);

// nnv_bitwise_stream.hip (K4c): the bit-wise bank with 32 received rows resident in LDS and every position's first
// layer streamed from L2 into the MFMA A operand.  Eligible: fp32, >= 2 positions (taken from 33 on), n_hidden <= 32, LDS for one row tile
// ((2 ceil(n_input / 8) + 2) KB) <= 160 KB, packed first layers < 2 GB.
struct BankStreamParams {
  const float *x;          // [n_rows][row_stride]
  int64_t row_stride;
  int64_t n_rows;
  int64_t n_blocks;        // ceil(n_rows / 32)
  int32_t x16;             // every row start 16-byte aligned
  int32_t K;               // n_input
  int32_t groups;          // ceil(K / 8)
  int32_t n_pos;
  int32_t w_bytes;         // n_pos * groups * 1024
  const float *w;          // [pos][group][lane 64][4]: W1[pos][8g + 4 (lane >> 5) + e][lane & 31], zero padded
  const float *bias;       // [pos][32]
  const float *head_w;     // [pos][32 hidden][2 outputs]
  const float *head_b;     // [pos][2]
  uint8_t *bits;           // [n_pos][n_rows]
  float *head_act;         // nullable [n_pos][n_rows][2]
  const uint8_t *labels;   // nullable [n_pos][n_rows]
  unsigned long long *errors;  // nullable [n_pos], ACCUMULATED (needs labels)
};
int64_t bank_stream_lds_bytes(int groups);
nnv_status bank_stream_prepare(int groups, int device);
nnv_status bank_stream_launch(const BankStreamParams &p, int num_cus, int device, hipStream_t stream);
// nnv_bitwise_single.hip (K4d): ONE position, the contraction split over the 16 waves, rows straight from HBM into the
// B operand, first-layer slice in registers.  p describes that one position (every per-position pointer pre-offset).
bool bank_single_eligible(int groups, int64_t row_stride);
nnv_status bank_single_prepare(int groups, int device);
nnv_status bank_single_launch(const BankStreamParams &p, int num_cus, int device, hipStream_t stream);
// ... and its f16x3 variant: p.groups = steps of 16 inputs, p.w = [step][hi | lo][lane 64][8 halfs]
bool bank_single16_eligible(int steps, int64_t row_stride);
nnv_status bank_single16_prepare(int steps, int device);
nnv_status bank_single16_launch(const BankStreamParams &p, int num_cus, int device, hipStream_t stream);

// nnv_generator.hip
nnv_status generate_launch(int32_t code, int32_t tile_bits, int64_t n_rows,
                           int64_t first_row, int64_t rows_per_block,
                           const float *sigma, int32_t n_sigma, uint64_t seed,
                           float *x, int32_t *labels, hipStream_t stream);

}  // namespace nnv

namespace nnv {
struct HostPathCtx;                           // nnv_host_path.hip: persistent pinned staging ring
void host_path_release(nnv_mlp_handle *h);    // frees it (nnv_mlp_free)
}  // namespace nnv

namespace nnv {
// nnv_mlp_split.hip: K1w, the fused decode for short inputs (a workgroup of four waves per 32-row tile, weights from L2)
bool split_supported(const PackedShape &s);
nnv_status split_launch(const PackedShape &s, const DecodeParams &p, hipStream_t stream);
// NNV_MLP_PATH_AUTO takes K1w up to this many 32-row tiles (where K1's persistent workgroups would each walk one tile
// behind a full staging copy of the network); measured crossover: see DESIGN.md section 5
constexpr int64_t kSplitAutoMaxTiles = 1024;
}  // namespace nnv

struct nnv_mlp_handle {
  nnv::PackedShape shape;
  int32_t dims[NNV_MAX_LAYERS + 1];
  float *wpack_dev;
  int64_t wpack_floats;
  int device;
  int num_cus;
  int lds_limit;
  int threads;  // workgroup size of the decode kernels, picked at prepack
  int vgprs;    // of the decode instantiation the benchmark call shape uses (lean if eligible)
  nnv::HostPathCtx *host_ctx;  // created by the first nnv_mlp_decode_host_f32 call on this handle
  int32_t path;                // nnv_mlp_path: AUTO unless nnv_mlp_set_path pinned it
};
