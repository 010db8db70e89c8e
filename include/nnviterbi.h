/*
 * nnviterbi.h -- C ABI of the MI355X-native neural Viterbi decode path.
 *
 * This is the drop-in boundary for ONE path of
 * enessinanparildi/Neural-Network-Based-Viterbi-Decoding: the batched
 * feed-forward forward-pass decoder
 *     received soft bits -> (dense -> +bias -> tanh) x depth -> argmax -> bits
 * The reference has no FFI of its own (it is pure Python on TensorFlow 1.x);
 * the Python call sites each entry point stands in for are cited per function
 * as  src/<file>:<lines>  relative to the reference tree.  The Python-side
 * binding (ctypes) is shown in INTEGRATION.md and implemented in
 * neural-network-based-viterbi-decoding_amd/_cabi.py.
 *
 * Conventions
 *   - plain C types only; every buffer is owned by the caller;
 *   - pointers named *_dev are device (HBM) pointers, *_host are host pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     every launch is stream-ordered and returns without synchronising;
 *   - every function returns an nnv_status (0 = NNV_OK); no exceptions, no
 *     global mutable state besides a kernel-attribute cache kept per (device,
 *     kernel), so one process may drive several GPUs; the library reads NO
 *     environment variable: which kernel a call runs is a function of the
 *     handle and the call's arguments alone (nnv_*_set_path pins it per handle);
 *   - a handle belongs to the HIP device that was current when it was packed:
 *     every launch entry compares that device with the calling thread's current
 *     device FIRST and returns NNV_ERR_INVALID_ARGUMENT on a mismatch (nothing is
 *     launched; foreign-device pointers would fault, not fail);
 *   - there is NO CPU fallback: without a HIP device every compute entry
 *     point returns NNV_ERR_NO_DEVICE / the HIP error.
 */
#ifndef NNVITERBI_H_
#define NNVITERBI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNV_ABI_VERSION 11 /* 11: nnv_dense_train_desc.loss (softmax cross-entropy: feed-forward decoders beyond 256 classes train through K8); 10: + nnv_mlp_set_path / nnv_mlp_path_for (K1w, the decode kernel for short inputs); 9: + nnv_lstm_seq_forward_f32 / _backward_f32 (the recurrent decoder's back-propagation through time); 8: + nnv_dense_train_* (the bit-wise / sliding-window decoders' training step); 7: + nnv_train_* (the training step as two kernels); prior: 2: + nnv_check_device, nnv_mlp_decode_host_f32, nnv_bitwise_*; handles carry their device; 3: + NNV_PRECISION_F16; 4: + nnv_bitwise_prepack_ex; 5: + NNV_CODE_CONTINUOUS; 6: + nnv_wide_set_path / _get_path, nnv_bitwise_set_path / _path_for (no environment switches any more) */
#define NNV_MAX_LAYERS 4 /* dense layers incl. the output layer */
#define NNV_MLP_MAX_INPUTS 32     /* nnv_mlp_*: row length n_input = 2 b (block_length <= 16); hidden widths: the first 1-3 of
                                     the reference's (128, 64, 32); other shapes: NNV_ERR_UNSUPPORTED_SHAPE -> nnv_wide_* */
#define NNV_MLP_MAX_CLASSES 65536 /* nnv_mlp_*: class layer (2^16); beyond: NNV_ERR_UNSUPPORTED_SHAPE */

typedef enum nnv_status {
  NNV_OK = 0,
  NNV_ERR_INVALID_ARGUMENT = 1, /* NULL where not allowed, bad sizes        */
  NNV_ERR_UNSUPPORTED_SHAPE = 2, /* no kernel instantiation for this network */
  NNV_ERR_NO_DEVICE = 3,         /* no HIP device / wrong architecture       */
  NNV_ERR_HIP = 4,               /* a HIP runtime call failed                */
  NNV_ERR_OUT_OF_MEMORY = 5
} nnv_status;

/* Arithmetic of the dense layers (nnv_mlp_prepack_ex). */
typedef enum nnv_precision {
  NNV_PRECISION_F32 = 0,   /* fp32 operands, fp32 MFMA: an exact fp32 fmaf chain  */
  NNV_PRECISION_F16X3 = 1, /* operands split hi+lo into two f16 (22 bits), three
                              f16 MFMAs per product block, fp32 accumulate:
                              ~2^-21 relative error per product, ~3x faster    */
  NNV_PRECISION_F16 = 2    /* nnv_mlp_* only: operands rounded ONCE to f16 (11 bits), one f16 MFMA
                              per product block, fp32 accumulate.  OUTSIDE the 1e-5 parity contract
                              (outputs within 1e-2 of float64 = the tested bound, 7.8e-3 observed on
                              the trained config-3 network; no reference fixture pins this mode; class
                              flips and BER delta are reported by bench.py): the opt-in "fast mode
                              with measured BER delta"                                                */
} nnv_precision;

/* Convolutional codes of the reference's generators. */
typedef enum nnv_code {
  NNV_CODE_57 = 0,     /* encode_57,     src/fnn_viterbi.py:122-154 (quirks kept) */
  NNV_CODE_133171 = 1, /* encode_133171, src/fnn_viterbi.py:157-217               */
  /* OR'ed into either code for nnv_awgn_generate_f32: the info bits of a BLOCK (rows_per_block consecutive
   * rows) are ONE stream, encoded from zero state at the block's first row and chopped into rows -- the
   * reference's "random test set" (src/fnn_viterbi_mixed_snr.py:464-475: encode the whole random stream, then
   * reshape to [size, 2b]).  Without it every row is encoded from zero state, like a class (:351-352).      */
  NNV_CODE_CONTINUOUS = 0x100
} nnv_code;

/*
 * Network description: the reference's weights/biases dictionaries
 * (initialize_weights, src/fnn_viterbi.py:413-468): depth dense layers,
 * W[l] is [dims[l]][dims[l+1]] row-major fp32 (right-multiply x @ W),
 * b[l] is [dims[l+1]] fp32.  dims[0] = n_input, dims[depth] = n_classes.
 * All pointers are HOST pointers (the whole network is <= 0.3 MB).
 */
typedef struct nnv_mlp_desc {
  int32_t depth;                    /* 2..NNV_MAX_LAYERS                     */
  int32_t dims[NNV_MAX_LAYERS + 1];
  const float *W_host[NNV_MAX_LAYERS];
  const float *b_host[NNV_MAX_LAYERS];
} nnv_mlp_desc;

/* Opaque: weights re-laid-out in MFMA A-fragment order, resident in HBM. */
typedef struct nnv_mlp_handle nnv_mlp_handle;

/* ABI version of the loaded library (== NNV_ABI_VERSION of its header). */
int32_t nnv_abi_version(void);

/* Static, NUL-terminated description of a status code. */
const char *nnv_status_string(int32_t status);

/* Number of visible gfx950 devices (0 if none); never fails. */
int32_t nnv_device_count(void);

/* The device rule every launch entry applies to its handle: NNV_OK iff `device`
 * is the calling thread's current HIP device, NNV_ERR_INVALID_ARGUMENT if another
 * device is current, NNV_ERR_NO_DEVICE without a HIP device. */
int32_t nnv_check_device(int32_t device);

/*
 * Replaces: building the TF variables + graph for multilayer_perceptron
 * (src/fnn_viterbi.py:538-545; src/fnn_viterbi_mixed_snr.py:712-720) and
 * tf.train.Saver.restore (mixed_snr:809-811).  Packs the network for the
 * current device and uploads it.  Synchronous.  *out must be released with
 * nnv_mlp_free.
 */
int32_t nnv_mlp_prepack(const nnv_mlp_desc *net, nnv_mlp_handle **out);

/* Same, choosing the arithmetic (nnv_precision).  nnv_mlp_prepack == F32.
 * F16X3 needs every |weight| < 11350 (the packed network carries the weights scaled by up
 * to 4 log2(e), see pack_network_f16x3; f16 range 65504) and inputs within the f16 range;
 * the fused kernel hands 1/(1+e^{2z}) instead of tanh(z) = 1 - 2/(1+e^{2z}) between layers
 * in this mode, with the affine map folded into the next layer's packed weights. */
int32_t nnv_mlp_prepack_ex(const nnv_mlp_desc *net, int32_t precision,
                           nnv_mlp_handle **out);

int32_t nnv_mlp_free(nnv_mlp_handle *h);

/*
 * THE HOT PATH.  Replaces
 *   predictions = tf.argmax(multilayer_perceptron(x, W, b), 1).eval({x: test_set})
 *     src/fnn_viterbi.py:494-510,606-610; src/fnn_viterbi_mixed_snr.py:589-592,787-791,813-818,833
 *   np.binary_repr(pred, width=b) per row        src/fnn_viterbi.py:648-651
 *   the BER/FER counting loops                   src/fnn_viterbi.py:639,654-667;
 *                                                src/fnn_viterbi_mixed_snr.py:326-328,650-655
 * in one fused launch.
 *
 *  x_dev        [n_rows][dims[0]] fp32, C-contiguous, 4-byte aligned
 *  labels_dev   nullable [n_rows] int32 true class per row
 *  cls32_dev    nullable [n_rows] int32 decoded class (argmax, first maximum)
 *  cls64_dev    nullable [n_rows] int64 the same, in tf.argmax's dtype
 *  bits_dev     nullable [n_rows][tile_bits] uint8, MSB first
 *  counters_dev nullable uint64[4] {bit_errors, frame_errors, total_bits,
 *               total_frames}; the launch ADDS to them (zero them once, run
 *               many launches); requires labels_dev
 *  tile_bits    b; must satisfy 2^b >= dims[depth] for bits/counters
 * n_rows == 0 is a no-op returning NNV_OK.
 */
int32_t nnv_mlp_decode_f32(const nnv_mlp_handle *h, const float *x_dev,
                           int64_t n_rows, int32_t tile_bits,
                           const int32_t *labels_dev, int32_t *cls32_dev,
                           int64_t *cls64_dev, uint8_t *bits_dev,
                           uint64_t *counters_dev, void *stream);

/*
 * The reference's own call shape: HOST arrays in, HOST predictions out,
 *   predictions = predictions_tensor.eval(feed_dict={x: test_set})
 *     src/fnn_viterbi.py:606-610 (test_set is a numpy array; so are the results),
 * as one SYNCHRONOUS call.  The rows are streamed through a ring of page-locked
 * staging slots owned by the handle (allocated by the first call, reused by
 * later ones, released by nnv_mlp_free): pageable -> pinned copies on
 * `n_threads` host threads, H2D DMA, the fused decode launch and the D2H of the
 * classes run concurrently on different chunks, so the call is bound by the
 * host link (4 dims[0] + 4 bytes per row), not by staging.
 *  x_host        [n_rows][dims[0]] fp32, any (pageable) host memory
 *  labels_host   nullable [n_rows] int32
 *  cls32_host / cls64_host  nullable [n_rows] outputs
 *  counters_host nullable uint64[4], ADDED to; requires labels_host
 *  chunk_rows    rows per pipeline stage (0 = 64 MB worth of rows)
 *  n_threads     staging threads (0 = the CPUs this process may use, <= 8)
 * Calls on one handle are serialised.  Must be called with the handle's device
 * current.  Uses its own internal streams.
 */
int32_t nnv_mlp_decode_host_f32(nnv_mlp_handle *h, const float *x_host,
                                int64_t n_rows, int32_t tile_bits,
                                const int32_t *labels_host, int32_t *cls32_host,
                                int64_t *cls64_host, uint64_t *counters_host,
                                int64_t chunk_rows, int32_t n_threads);

/*
 * Which kernel nnv_mlp_decode_f32 launches.  K1 (PERSISTENT) stages the packed network into LDS once per workgroup
 * and walks 32-row tiles wave by wave: the throughput shape.  K1w (SPLIT) gives every 32-row tile a workgroup of four
 * waves that share each layer's tiles and read the weights from L2: the latency shape for short inputs (BASELINE
 * configs[0], 2,000 rows: the call the reference makes as predictions_tensor.eval(feed_dict={x: test_set}),
 * src/fnn_viterbi.py:606-610, with a small test_set).  Same packed network, same arithmetic, same decision rule: the
 * two return the same classes bit for bit.  AUTO (the default) takes K1w up to 1,024 tiles for fp32 networks (every
 * shape nnv_mlp_prepack takes) and K1 otherwise; nnv_mlp_decode_binned_f32 and nnv_mlp_forward_f32 always
 * run K1.  nnv_mlp_set_path(SPLIT) on a network K1w does not take: NNV_ERR_UNSUPPORTED_SHAPE, nothing changes.  Not to
 * be called while another thread decodes with the same handle.
 */
typedef enum nnv_mlp_path {
  NNV_MLP_PATH_AUTO = 0,
  NNV_MLP_PATH_PERSISTENT = 1, /* K1  */
  NNV_MLP_PATH_SPLIT = 2       /* K1w */
} nnv_mlp_path;
int32_t nnv_mlp_set_path(nnv_mlp_handle *h, int32_t path);
/* The kernel a decode of n_rows rows runs on this handle: NNV_MLP_PATH_PERSISTENT or NNV_MLP_PATH_SPLIT (negative
 * nnv_status for bad arguments). */
int32_t nnv_mlp_path_for(const nnv_mlp_handle *h, int64_t n_rows);

/*
 * The same launch with error counters kept PER SNR BIN, for mixed-SNR sweeps
 * (BASELINE config 4; the reference evaluates one SNR per run,
 * src/fnn_viterbi_mixed_snr.py:452-460,813-845): row r of this call belongs to
 * block (first_row + r) / rows_per_block and to bin block % n_bins, the rule
 * nnv_awgn_generate_f32 uses to pick sigma.  counters_dev is uint64[n_bins][4],
 * each row laid out like nnv_mlp_decode_f32's counters, accumulated.
 * 1 <= n_bins <= 16.  labels_dev and counters_dev are required.
 */
int32_t nnv_mlp_decode_binned_f32(const nnv_mlp_handle *h, const float *x_dev,
                                  int64_t n_rows, int32_t tile_bits,
                                  const int32_t *labels_dev, int32_t *cls32_dev,
                                  uint64_t *counters_dev, int64_t first_row,
                                  int64_t rows_per_block, int32_t n_bins,
                                  void *stream);

/*
 * Debug / parity entry: the tanh'd output layer itself,
 * multilayer_perceptron(x, W, b) (src/fnn_viterbi.py:471-510).
 *  out_dev  [n_rows][dims[depth]] fp32.  Slow (strided stores).
 */
int32_t nnv_mlp_forward_f32(const nnv_mlp_handle *h, const float *x_dev,
                            int64_t n_rows, float *out_dev, void *stream);

/*
 * Synthetic input generator (benchmark harness and on-device data path).
 * Replaces the per-row Python loops of train_test_dataset
 * (src/fnn_viterbi.py:343-388) for random (not class-major) rows: row r of
 * the GLOBAL stream (r = first_row + local index) draws tile_bits info bits
 * and 2*tile_bits Gaussian samples from Philox4x32-10 keyed by (seed, r), so
 * the data do not depend on how rows are sharded over GPUs; the bits are
 * encoded from zero state with `code` exactly like the reference encodes a
 * class (src/fnn_viterbi.py:351-352), BPSK-mapped (0 -> -1, 1 -> +1,
 * src/fnn_viterbi.py:244-247) and sigma * N(0,1) is added
 * (src/fnn_viterbi.py:250-251).
 *
 * With NNV_CODE_CONTINUOUS in `code` the encoder state runs on from row to row
 * inside a block (rows_per_block rows = one stream from zero state; the labels
 * stay each row's own info bits, as the reference labels its random test set,
 * src/fnn_viterbi_mixed_snr.py:468-475); tile_bits >= 6 then for either code.
 *
 *  sigma_host      n_sigma standard deviations; block i = r / rows_per_block
 *                  uses sigma_host[i % n_sigma] (config 4's mixed-SNR sweep;
 *                  n_sigma == 1 for a single SNR)
 *  x_dev           [n_rows][2*tile_bits] fp32 out
 *  labels_dev      nullable [n_rows] int32 out: class index, MSB-first value
 *                  of the info bits (src/fnn_viterbi.py:281-285)
 */
int32_t nnv_awgn_generate_f32(int32_t code, int32_t tile_bits, int64_t n_rows,
                              int64_t first_row, int64_t rows_per_block,
                              const float *sigma_host, int32_t n_sigma,
                              uint64_t seed, float *x_dev, int32_t *labels_dev,
                              void *stream);

/* ------------------------------------------------------------------------
 * Wide networks (layer-at-a-time path): hidden widths beyond the register-
 * chained kernel, e.g. the reference's sliding-window decoder
 *   200 -> 500 -> 500 -> 500 -> 500 -> 2   src/fnn_viterbi_seq_predict.py:343-414
 * Two kernels behind one entry point:
 *  - the whole chain in ONE persistent launch, activations resident in LDS
 *    (fp32, every layer width -- class layer included -- and the row length
 *    <= 512: the sliding-window network, feed-forward decoders with other hidden
 *    widths; no workspace traffic) -- taken automatically;
 *  - otherwise one fp32-MFMA GEMM launch per layer (bias + activation fused),
 *    activations in the caller-provided HBM workspace, a small head kernel for
 *    <= 32 classes or a row-argmax for wider class layers.
 *    (nnv_wide_set_path pins a handle to either; the default is the first
 *    whenever the shape allows it.)
 * ------------------------------------------------------------------------ */
#define NNV_WIDE_MAX_LAYERS 8
#define NNV_WIDE_MAX_WIDTH 16384   /* row length and hidden widths; wider: NNV_ERR_UNSUPPORTED_SHAPE */
#define NNV_WIDE_MAX_CLASSES 65536 /* class layer (block_length <= 16 as 2^b classes); wider: NNV_ERR_UNSUPPORTED_SHAPE */

typedef enum nnv_activation {
  NNV_ACT_TANH = 0, /* src/fnn_viterbi_seq_predict.py:408-412 */
  NNV_ACT_RELU = 1  /* src/fnn_viterbi_bitwise.py:388-397 (output layer included) */
} nnv_activation;

typedef struct nnv_wide_desc {
  int32_t depth;                         /* 2..NNV_WIDE_MAX_LAYERS dense layers */
  int32_t dims[NNV_WIDE_MAX_LAYERS + 1]; /* dims[0] = row length, dims[depth] = n_classes */
  const float *W_host[NNV_WIDE_MAX_LAYERS]; /* [dims[l]][dims[l+1]] row-major */
  const float *b_host[NNV_WIDE_MAX_LAYERS];
  int32_t activation;                    /* nnv_activation, applied after EVERY layer */
  int32_t precision;                     /* nnv_precision; F16X3 needs n_classes <= 32 and
                                            every |weight| < 65504                        */
} nnv_wide_desc;

typedef struct nnv_wide_handle nnv_wide_handle;

/* Replaces build_network's variables (src/fnn_viterbi_seq_predict.py:343-414). */
int32_t nnv_wide_prepack(const nnv_wide_desc *net, nnv_wide_handle **out);
int32_t nnv_wide_free(nnv_wide_handle *h);

/* Kernel choice of a wide handle (ABI 6). */
typedef enum nnv_wide_path {
  NNV_WIDE_PATH_AUTO = 0,    /* the one-launch chain (K6) when the handle holds its packing, else layered */
  NNV_WIDE_PATH_FUSED = 1,   /* K6, nnv_wide_fused.hip */
  NNV_WIDE_PATH_LAYERED = 2  /* one GEMM launch per layer (K4), nnv_dense.hip */
} nnv_wide_path;

/* Pins the handle to one kernel (parity tests hold the two against each other; A/B timing).  FUSED on a handle
 * whose shape K6 does not take returns NNV_ERR_UNSUPPORTED_SHAPE and changes nothing.  Not to be called while
 * another thread decodes with the same handle. */
int32_t nnv_wide_set_path(nnv_wide_handle *h, int32_t path);
/* The kernel nnv_wide_decode_f32 runs on this handle: NNV_WIDE_PATH_FUSED or NNV_WIDE_PATH_LAYERED
 * (negative nnv_status for a NULL handle). */
int32_t nnv_wide_get_path(const nnv_wide_handle *h);

/* Bytes of workspace needed to process `chunk_rows` rows at a time. */
int64_t nnv_wide_workspace_bytes(const nnv_wide_handle *h, int64_t chunk_rows);

/*
 * Replaces  predictions = tf.argmax(pred, 1).eval({x: test_data})
 * (src/fnn_viterbi_seq_predict.py:497,524,611) AND the Python loop that
 * materialises test_data from the stream (:507-519, get_single_instance
 * :293-336): row i of the input is the `dims[0]` floats starting at
 * x_dev + i * row_stride.  row_stride == dims[0] is an ordinary [n_rows, dims[0]]
 * matrix; row_stride == 2 with x_dev = the received stream is the sliding
 * window (window i = coded[2i : 2i + window_length]).  x_dev must be 8-byte
 * aligned and hold (n_rows - 1) * row_stride + dims[0] floats.
 *  workspace_dev  device scratch, >= nnv_wide_workspace_bytes(h, 128) bytes;
 *                 rows are processed in chunks that fit it
 *  cls32_dev / cls64_dev  nullable [n_rows] decoded class (first maximum)
 *  act_dev        nullable [n_rows][n_classes] tanh'd class layer (parity)
 */
int32_t nnv_wide_decode_f32(const nnv_wide_handle *h, const float *x_dev,
                            int64_t n_rows, int64_t row_stride,
                            void *workspace_dev, int64_t workspace_bytes,
                            int32_t *cls32_dev, int64_t *cls64_dev,
                            float *act_dev, void *stream);

/* ------------------------------------------------------------------------
 * Bit-wise decoder BANK: the reference trains one small ReLU network per
 * information-bit position (`bit_position`),
 *   2(6+L) -> n_hidden_1 (30) -> 2, ReLU after both layers, argmax = the bit
 *   src/fnn_viterbi_bitwise.py:326-399 (build_network), :503-505 (eval)
 * and decodes a block by running them one after the other over the same
 * received rows.  The bank stacks the P first layers into one fp32-MFMA GEMM
 * (32 columns per position) and finishes each position's 2-logit head in the
 * GEMM epilogue: one launch, every row read once per 4 positions' tile, the
 * hidden activations never leave the chip.
 * The fp32 bank picks its kernel by size: >= 33 positions run with 32
 * received rows resident in LDS and every position's first layer streamed
 * from L2 into the MFMA A operand (K4c; same results as the tiled GEMM K4b
 * bit for bit); 1-2 positions (the reference's own call shape is ONE per run)
 * with 16-byte aligned rows run one HBM-streaming launch per position, the
 * contraction split over the waves (K4d; same tolerances, another summation
 * order); everything else the LDS-tiled GEMM.  nnv_bitwise_set_path pins a
 * handle to one of them.
 * ------------------------------------------------------------------------ */
typedef struct nnv_bitwise_desc {
  int32_t n_input;        /* 2 (constraint_length + block_length); even           */
  int32_t n_hidden;       /* n_hidden_1 <= 32 (reference default 30)               */
  int32_t n_positions;    /* P networks                                            */
  const float *W1_host;   /* [P][n_input][n_hidden]  weights['h1'] of each network */
  const float *b1_host;   /* [P][n_hidden]           biases['b1']                  */
  const float *Wout_host; /* [P][n_hidden][2]        weights['out']                */
  const float *bout_host; /* [P][2]                  biases['out']                 */
} nnv_bitwise_desc;

typedef struct nnv_bitwise_handle nnv_bitwise_handle;

int32_t nnv_bitwise_prepack(const nnv_bitwise_desc *net, nnv_bitwise_handle **out);
/* precision: NNV_PRECISION_F32 (= nnv_bitwise_prepack) or NNV_PRECISION_F16X3 (the stacked first
 * layers as hi/lo f16 planes, three f16 MFMAs per product block; the heads stay fp32 on the fp32
 * accumulators; every first-layer |weight| < 65504).  ABI 4. */
int32_t nnv_bitwise_prepack_ex(const nnv_bitwise_desc *net, int32_t precision, nnv_bitwise_handle **out);
int32_t nnv_bitwise_free(nnv_bitwise_handle *h);

/* Kernel choice of a bank handle (ABI 6). */
typedef enum nnv_bitwise_path {
  NNV_BITWISE_PATH_AUTO = 0,   /* by size, as described above */
  NNV_BITWISE_PATH_TILED = 1,  /* K4b: stacked first layers as one LDS-tiled GEMM, heads in its epilogue */
  NNV_BITWISE_PATH_STREAM = 2, /* K4c: rows resident in LDS, first layers streamed from L2 (fp32, >= 2 positions) */
  NNV_BITWISE_PATH_SINGLE = 3  /* K4d: one HBM-streaming launch per position (banks of <= 2 positions; f16x3: 1) */
} nnv_bitwise_path;

/* Pins the handle to one kernel.  A path whose packing the handle does not hold (K4c on a one-position or f16x3
 * bank, K4d on a large one) returns NNV_ERR_UNSUPPORTED_SHAPE and changes nothing.  A pinned K4d still needs
 * 16-byte aligned rows per call; calls that cannot take it run K4b (nnv_bitwise_path_for tells).  Not to be
 * called while another thread decodes with the same handle. */
int32_t nnv_bitwise_set_path(nnv_bitwise_handle *h, int32_t path);
/* The kernel nnv_bitwise_decode_f32 would run for rows at x_dev: NNV_BITWISE_PATH_TILED / _STREAM / _SINGLE
 * (negative nnv_status for a NULL handle). */
int32_t nnv_bitwise_path_for(const nnv_bitwise_handle *h, const float *x_dev);

/*
 * Replaces P runs of  predictions = tf.argmax(pred, 1).eval({x: test_encoded})
 * (src/fnn_viterbi_bitwise.py:503-505), one per trained bit position.
 *  x_dev       [n_rows][n_input] fp32, 8-byte aligned (the noisy codewords)
 *  bits_dev    [P][n_rows] uint8 out: position-major, i.e. one prediction vector
 *              per network; first maximum on ties (both outputs ReLU'd to 0 -> bit 0)
 *  act_dev     nullable [P][n_rows][2] fp32 out: the ReLU'd outputs (parity)
 *  labels_dev  nullable [P][n_rows] uint8 true bits (test_info[:, bit_position])
 *  errors_dev  nullable uint64[P] wrong decisions per position, ACCUMULATED;
 *              needs labels_dev
 */
int32_t nnv_bitwise_decode_f32(const nnv_bitwise_handle *h, const float *x_dev,
                               int64_t n_rows, uint8_t *bits_dev, float *act_dev,
                               const uint8_t *labels_dev, uint64_t *errors_dev,
                               void *stream);

/* ------------------------------------------------------------------------
 * Recurrent decoder (LSTM encoder -> LSTM decoder -> linear), the inference
 * tensors of build_seq2seq_model, src/rnn_viterbi.py:235-340:
 *   encoder LSTMCell(H) + dynamic_rnn over encoder_inputs      :289-297
 *   decoder LSTMCell(H), initial_state = encoder final state   :302-312
 *   decoder_logits = linear(decoder_outputs, vocab_size = 1)   :315-318
 * TF-1.x LSTMCell arithmetic: kernel [1 + H][4H] (row 0 multiplies the scalar
 * input, rows 1.. the previous output m), gate columns in the order i, j, f,
 * o, c = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j), m = sigmoid(o)
 * tanh(c).  The reference script itself cannot be run.
 * One wave carries 32 sequences through the whole recurrence with the state
 * in registers; every array is TIME-MAJOR [T][batch], as the reference's
 * placeholders are declared (:263-279).
 * ------------------------------------------------------------------------ */
typedef struct nnv_seq2seq_desc {
  int32_t hidden;           /* H = encoder_hidden_units = decoder_hidden_units, 1..22 */
  float forget_bias;        /* LSTMCell default 1.0 */
  int32_t precision;        /* nnv_precision: F32 exact fp32 MFMA chain; F16X3 hi/lo f16 operand
                               split on the 16-bit matrix pipe (every scaled |weight| < 65504) */
  const float *enc_kernel;  /* [1 + H][4H] row-major, host */
  const float *enc_bias;    /* [4H] */
  const float *dec_kernel;
  const float *dec_bias;
  const float *proj_w;      /* [H] (the [H, 1] weights of linear()) */
  const float *proj_b;      /* [1] */
} nnv_seq2seq_desc;

typedef struct nnv_seq2seq_handle nnv_seq2seq_handle;

/* Replaces the variables created by build_seq2seq_model (:289-318) /
 * tf.global_variables_initializer (:405): packs both cells for the current
 * device.  Synchronous. */
int32_t nnv_seq2seq_prepack(const nnv_seq2seq_desc *net, nnv_seq2seq_handle **out);
int32_t nnv_seq2seq_free(nnv_seq2seq_handle *h);

/*
 * Replaces  sess.run(model['decoder_prediction'], feed_dict=test_feed_dict)
 * (src/rnn_viterbi.py:471-474), returning the logits the prediction is taken
 * from (tf.argmax over the size-1 vocabulary axis, :321, is identically 0 and
 * is left to the caller).
 *  enc_in_dev   [t_enc][batch] fp32   encoder_inputs (received soft symbols)
 *  dec_in_dev   [t_dec][batch] fp32   decoder_inputs (EOS, info, padding :378-386)
 *  logits_dev   nullable [t_dec][batch] fp32 out
 *  bits_dev     nullable [n_bits][batch] uint8 out: logit > 0 for the first
 *               n_bits decoder steps (NOT in the reference: the decision a
 *               network trained towards +-1 targets takes)
 *  info_dev     nullable [n_bits][batch] uint8 0/1 information bits
 *  counters_dev nullable uint64[4] {bit_errors, frame_errors, total_bits,
 *               total_frames} of bits vs info, ACCUMULATED; needs info_dev
 * Asynchronous on `stream`.
 */
int32_t nnv_seq2seq_forward_f32(const nnv_seq2seq_handle *h, const float *enc_in_dev,
                                const float *dec_in_dev, int32_t t_enc, int32_t t_dec,
                                int64_t batch, float *logits_dev, uint8_t *bits_dev,
                                const uint8_t *info_dev, int32_t n_bits,
                                uint64_t *counters_dev, void *stream);

/* ------------------------------------------------------------------------
 * The TRAINING STEP of the feed-forward decoder (ABI 7), SURVEY.md section 8f rank 1:
 *   cost = reduce_mean(softmax_cross_entropy_with_logits(logits = the TANH'd class layer, labels = y))
 *          + beta (l2_loss(h1) + l2_loss(h2) + l2_loss(h3))                  src/fnn_viterbi.py:549-559
 *   tf.train.AdamOptimizer(learning_rate).minimize(cost)                      src/fnn_viterbi.py:561-563
 *   (TF 1.x: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), epsilon 1e-8 OUTSIDE the bias correction)
 *   optional tf.nn.dropout after every hidden tanh                src/fnn_viterbi_mixed_snr.py:596-627
 * for the reference's architecture n_input -> 128 -> 64 -> 32 -> n_classes with n_input <= 32, n_classes <= 256 and a
 * batch of at most 32,768 rows (other shapes: NNV_ERR_UNSUPPORTED_SHAPE; the Python trainer then runs them through
 * PyTorch autograd).  One step = one forward/backward kernel (four waves per 32 rows, every layer's tiles dealt over
 * them and handed over through LDS) + one Adam kernel; no atomics, every sum in a fixed order: results independent of
 * the launch.
 * ------------------------------------------------------------------------ */
typedef struct nnv_train_desc {
  int32_t n_input, n_classes, batch_size;
  int32_t n_hidden[3];          /* must be {128, 64, 32} */
  float learning_rate;          /* DecoderConfig.learning_rate (1e-4) */
  float beta_regularization;    /* DecoderConfig.beta_regularization (1e-3) */
  float keep_prob;              /* 1 = no dropout; < 1: every step call must bring its masks */
  const float *W_host[4];       /* initial weights [in][out] row-major (initialize_weights, src/fnn_viterbi.py:413-468) */
  const float *b_host[4];
} nnv_train_desc;

typedef struct nnv_train_handle nnv_train_handle;

/* Replaces the variables + optimizer slots of run_network's graph (src/fnn_viterbi.py:538-567).  Synchronous.
 * A train handle is MUTABLE state (weights, Adam moments, the step count): one caller at a time, unlike the read-only
 * decode handles; it belongs to the device that was current at creation (other device: NNV_ERR_INVALID_ARGUMENT). */
int32_t nnv_train_create(const nnv_train_desc *net, nnv_train_handle **out);
int32_t nnv_train_free(nnv_train_handle *h);

/*
 * Replaces n_steps iterations of  sess.run([optimizer, cost], feed_dict={x: batch_x, y: batch_y})
 * (src/fnn_viterbi.py:574-594) over CONSECUTIVE batches: step t takes rows [t B, (t + 1) B) of x_dev / labels_dev.
 *  x_dev       [n_steps * B][n_input] fp32
 *  labels_dev  [n_steps * B] int32: the class of each row (the position of the 1 in the reference's one-hot batch_y)
 *  masks_dev   nullable {[n_steps * B][128], [n_steps * B][64], [n_steps * B][32]} fp32 0/1 dropout masks
 *              (required iff keep_prob < 1: the caller draws them, as tf.nn.dropout draws its own)
 *  costs_dev   nullable [n_steps] fp32 out: the cost of each batch BEFORE its update (what sess.run returns)
 * Asynchronous on `stream`; steps are stream-ordered.
 */
int32_t nnv_train_steps(nnv_train_handle *h, const float *x_dev, const int32_t *labels_dev, int32_t n_steps,
                        const float *const masks_dev[3], float *costs_dev, void *stream);

/* Parity / debug: cost and the eight gradients of ONE batch at the current weights (tf.gradients of the cost above),
 * no update.  Synchronous; host outputs in the reference's shapes ([in][out], [out]); any pointer may be NULL. */
int32_t nnv_train_gradients(nnv_train_handle *h, const float *x_dev, const int32_t *labels_dev,
                            const float *const masks_dev[3], float *cost_host, float *const gW_host[4],
                            float *const gb_host[4]);

/* Current weights to the host (the reference's Saver.save, src/fnn_viterbi_mixed_snr.py:782-784) and, if
 * cost_sum_host is not NULL, the sum of the step costs since the last export (reset by the call).  Synchronous. */
int32_t nnv_train_export(const nnv_train_handle *h, float *const W_host[4], float *const b_host[4], double *cost_sum_host);

/* ------------------------------------------------------------------------
 * K8: the training step of the reference's bit-wise and sliding-window decoders, for a dense network of 1..5 layers of
 * any width with the activation after EVERY layer (the output layer included):
 *   cost = reduce_mean(sigmoid_cross_entropy_with_logits(logits=pred, labels=y)) + beta (l2_loss(h1) + l2_loss(h2)),
 *   tf.train.AdamOptimizer(learning_rate).minimize(cost)
 *   -- src/fnn_viterbi_bitwise.py:388-397,438-452 (1036 -> 30 -> 2 ReLU, batch 1000),
 *      src/fnn_viterbi_seq_predict.py:404-412,475-489 (200 -> 500 x 4 -> 2 tanh, batch 100).
 * One step = 2 L + 1 launches (L forward with the loss in the last one's epilogue, L backward, Adam + staging of the
 * next batch) instead of ~60 PyTorch-autograd
 * kernels; exact fp32; every sum in a fixed order.  Handles are mutable state: one caller at a time.
 * ------------------------------------------------------------------------ */
typedef enum nnv_loss {
  NNV_LOSS_SIGMOID_CE = 0, /* reduce_mean over rows AND outputs of sigmoid_cross_entropy_with_logits on the activated outputs
                              (src/fnn_viterbi_bitwise.py:438-441, src/fnn_viterbi_seq_predict.py:475-478); n_output <= 32 */
  NNV_LOSS_SOFTMAX_CE = 1  /* reduce_mean over rows of softmax_cross_entropy_with_logits on the activated outputs, the
                              feed-forward decoder's (src/fnn_viterbi.py:549-553); n_output <= NNV_WIDE_MAX_CLASSES */
} nnv_loss;

typedef struct nnv_dense_train_desc {
  int32_t n_layers;             /* 1..5 */
  int32_t dims[6];              /* n_input, the hidden widths, n_output */
  int32_t activation;           /* nnv_activation */
  int32_t batch_size;
  float learning_rate, beta_regularization;
  uint32_t reg_layer_mask;      /* bit l: tf.nn.l2_loss(W_l) is part of the cost (both references: layers 0 and 1) */
  const float *W_host[5];       /* [in][out] row-major */
  const float *b_host[5];
  int32_t loss;                 /* nnv_loss */
  const float *extra_host;      /* nullable: variables that ONLY the regulariser names (the bit-wise decoder's unused h2,
                                   src/fnn_viterbi_bitwise.py:388-397 vs :444-447): gradient beta * value, moved by Adam */
  int64_t n_extra;
} nnv_dense_train_desc;

typedef struct nnv_dense_train_handle nnv_dense_train_handle;

/* Replaces the variables + optimizer slots of train_and_evaluate's graph (src/fnn_viterbi_bitwise.py:429-452,
 * src/fnn_viterbi_seq_predict.py:468-489).  Synchronous. */
int32_t nnv_dense_train_create(const nnv_dense_train_desc *net, nnv_dense_train_handle **out);
int32_t nnv_dense_train_free(nnv_dense_train_handle *h);

/*
 * Replaces n_steps iterations of sess.run([optimizer, cost], feed_dict={x: batch_x, y: batch_y})
 * (src/fnn_viterbi_bitwise.py:475-494, src/fnn_viterbi_seq_predict.py:541-577).  Row m of step t is
 * x_dev[t * step_stride + m * row_stride + k], k < n_input: a [rows][n_input] matrix has row_stride = n_input and
 * step_stride = B * n_input; the sliding-window decoder's overlapping windows of ONE stream (window i = stream[2 i ..],
 * src/fnn_viterbi_seq_predict.py:293-336) have row_stride = 2 and step_stride = 2 * B -- no window matrix is built.
 *  labels_dev  [n_steps * B] int32 class index (the position of the 1 in the reference's one-hot batch_y)
 *  costs_dev   nullable [n_steps] fp32 out: the cost of each batch BEFORE its update
 * Asynchronous on `stream`.
 */
int32_t nnv_dense_train_steps(nnv_dense_train_handle *h, const float *x_dev, int64_t row_stride, int64_t step_stride,
                              const int32_t *labels_dev, int32_t n_steps, float *costs_dev, void *stream);

/* Parity / debug: cost and the gradients of ONE batch at the current values, no update.  Synchronous; host outputs in
 * the reference's shapes; any pointer may be NULL. */
int32_t nnv_dense_train_gradients(nnv_dense_train_handle *h, const float *x_dev, int64_t row_stride, const int32_t *labels_dev,
                                  float *cost_host, float *const gW_host[5], float *const gb_host[5], float *gextra_host);

/* Current values to the host (Saver.save, src/fnn_viterbi_bitwise.py:498-501) and, if cost_sum_host is not NULL, the sum
 * of the step costs since the last export (reset by the call).  Synchronous. */
int32_t nnv_dense_train_export(const nnv_dense_train_handle *h, float *const W_host[5], float *const b_host[5], float *extra_host,
                               double *cost_sum_host);

/* ------------------------------------------------------------------------
 * K9: the recurrent half of the recurrent decoder's training step -- one tf.contrib.rnn.LSTMCell unrolled over time by
 * tf.nn.dynamic_rnn (src/rnn_viterbi.py:277-299; trained on the mean squared error by AdamOptimizer, :310-316), forward
 * with everything back-propagation needs kept, and backward.  One launch per LSTM and direction with every time step
 * inside (PyTorch autograd: ~40 kernels per cell, 400 cells per step).  The input projection, the output projection,
 * the loss, dW_h = sum_t m_{t-1}^T dz_t (one matrix product) and Adam stay with the caller.  All arrays are
 * [time][feature][batch] fp32 (batch contiguous); gate order i, j, f, o (BasicLSTMCell); hidden must be 20, the
 * reference's encoder_hidden_units (src/rnn_viterbi.py:36), other sizes: NNV_ERR_UNSUPPORTED_SHAPE.  Asynchronous.
 *  zx     [T][4H][B]  x_t kernel[0] + bias          wh  [H][4H]  kernel[1:]          c0, m0  [H][B]
 *  gates  [T][4H][B]  out: sigmoid(i), tanh(j), sigmoid(f + forget_bias), sigmoid(o)
 *  c_all, m_all [T][H][B] out: the cell and the output of every step (m_all[T-1], c_all[T-1]: the final state)
 * ------------------------------------------------------------------------ */
int32_t nnv_lstm_seq_forward_f32(const float *zx_dev, const float *wh_dev, const float *c0_dev, const float *m0_dev,
                                 int32_t n_steps, int32_t batch, int32_t hidden, float forget_bias, float *gates_dev,
                                 float *c_all_dev, float *m_all_dev, void *stream);
/*  dm_all [T][H][B] gradient w.r.t. every m_t as an output; dcT, dmT [H][B] w.r.t. the final state (the decoder's
 *  initial state is the encoder's final state, src/rnn_viterbi.py:291-296).  Out: dz [T][4H][B] w.r.t. the gate
 *  pre-activations (= w.r.t. zx), dc0, dm0 [H][B] w.r.t. the initial state. */
int32_t nnv_lstm_seq_backward_f32(const float *dm_all_dev, const float *dcT_dev, const float *dmT_dev, const float *gates_dev,
                                  const float *c_all_dev, const float *c0_dev, const float *wh_dev, int32_t n_steps,
                                  int32_t batch, int32_t hidden, float *dz_dev, float *dc0_dev, float *dm0_dev, void *stream);

/*
 * Timing helper used by bench.py: launches nnv_mlp_decode_f32 `iters` times
 * back-to-back on `stream` between two HIP events recorded on that same
 * stream and returns the elapsed milliseconds (synchronises the stream).
 */
int32_t nnv_mlp_decode_f32_timed(const nnv_mlp_handle *h, const float *x_dev,
                                 int64_t n_rows, int32_t tile_bits,
                                 const int32_t *labels_dev, int32_t *cls32_dev,
                                 uint64_t *counters_dev, void *stream,
                                 int32_t iters, float *elapsed_ms_host);

/* Introspection: fills info_host[0..7] = {threads per workgroup, workgroups
 * launched for n_rows, dynamic LDS bytes and VGPRs (0 if unknown) of the kernel
 * variant a classes+counters call on aligned rows gets, weights
 * resident in LDS (1/0 for the output layer), rows per wave tile, depth,
 * n_classes}; info_host[8] = precision; info_host[9] = class-layer M-tiles (32
 * classes each) resident in LDS, the rest being streamed from L2 by every wave
 * for its own rows (info_host must hold 10 ints). */
int32_t nnv_mlp_launch_info(const nnv_mlp_handle *h, int64_t n_rows,
                            int32_t *info_host);

#ifdef __cplusplus
}
#endif
#endif /* NNVITERBI_H_ */
