/*
 * fnn_oracle.c -- CPU restatement of the reference's feed-forward decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file, and only as the checker / the reported CPU baseline.
 *
 * PARITY PINNING.  The arithmetic of this path (MatMul, Add, Tanh, ArgMax)
 * lives in TensorFlow 1.x, which is neither vendored in /root/reference nor
 * installed in this image.  The reference's tests
 * hold no golden vector for the forward pass, so for the forward pass this
 * oracle is "parity unpinned": it is a text-faithful restatement of
 *   src/fnn_viterbi.py:494-510  (matmul -> add bias -> tanh, four times,
 *                                tanh on the output layer too)
 *   src/fnn_viterbi.py:606-610  (argmax over axis 1; first maximum wins,
 *                                the np.argmax convention used at :635)
 *   src/fnn_viterbi.py:648-651  (np.binary_repr(pred, width=b): MSB first)
 *   src/fnn_viterbi.py:639-671 and
 *   src/fnn_viterbi_mixed_snr.py:303-328,634-655 (FER = mismatching rows,
 *                                BER = popcount(pred XOR label) / (N*b)).
 * The numpy half around it (encoders, channel, metrics) IS pinned by golden
 * vectors produced from the reference itself: tests/golden/.
 *
 * Arithmetic: every pre-activation is an fp32 fmaf chain over k in natural
 * order starting from 0, then one fp32 add of the bias, then libm tanhf.
 * An optional per-layer k permutation lets a test replay the exact summation
 * order of the HIP kernel (MFMA is a k-ordered fmaf chain as well), so that
 * only the tanh implementation differs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FNN_ORACLE_MAX_LAYERS 8
#define FNN_ORACLE_MAX_WIDTH 65536 /* class layers up to 2^16 (block_length 16) */

/* one dense layer on one row: out[i] = tanhf( (sum_k in[k]*W[k][i]) + b[i] )
 * W is [n_in][n_out] row-major, the reference's right-multiply layout
 * (src/fnn_viterbi.py:432-456). */
static void dense_tanh_row(const float *in, int n_in, const float *W,
                           const float *bias, int n_out, const int32_t *korder,
                           int relu, float *out) {
  for (int i = 0; i < n_out; ++i) out[i] = 0.0f;
  for (int kk = 0; kk < n_in; ++kk) {
    const int k = korder ? korder[kk] : kk;
    const float xv = in[k];
    const float *w = W + (size_t)k * n_out;
    for (int i = 0; i < n_out; ++i) out[i] = fmaf(xv, w[i], out[i]);
  }
  /* tanh: src/fnn_viterbi.py:496; relu: src/fnn_viterbi_bitwise.py:388,397 */
  if (relu)
    for (int i = 0; i < n_out; ++i) { const float z = out[i] + bias[i]; out[i] = z > 0.0f ? z : 0.0f; }
  else
    for (int i = 0; i < n_out; ++i) out[i] = tanhf(out[i] + bias[i]);
}

/* Returns 0 on success, a negative value on bad arguments.
 *  x        [n_rows][dims[0]] fp32, C-contiguous
 *  W[l]     [dims[l]][dims[l+1]] fp32 ; b[l] [dims[l+1]]
 *  korder   nullable; korder[l] nullable; a permutation of 0..dims[l]-1
 *  labels   nullable [n_rows] int32 class indices
 *  cls_out  nullable [n_rows] int64
 *  bits_out nullable [n_rows][tile_bits] uint8, MSB first
 *  act_out  nullable [n_rows][dims[depth]] fp32 (the tanh'd output layer)
 *  counters nullable int64[4] = {bit_errors, frame_errors, total_bits,
 *           total_frames}; written (not accumulated); needs labels
 *  relu     0: tanh after every layer (fnn_viterbi, seq_predict); 1: ReLU after every
 *           layer, the class layer included (src/fnn_viterbi_bitwise.py:388-397)
 */
int fnn_oracle_decode_f32(const float *x, int64_t n_rows, int32_t depth,
                          const int32_t *dims, const float *const *W,
                          const float *const *b, const int32_t *const *korder,
                          int32_t tile_bits, const int32_t *labels,
                          int64_t *cls_out, uint8_t *bits_out, float *act_out,
                          int64_t *counters, int32_t n_threads, int32_t relu) {
  if (depth < 1 || depth > FNN_ORACLE_MAX_LAYERS || n_rows < 0) return -1;
  for (int l = 0; l <= depth; ++l)
    if (dims[l] < 1 || dims[l] > FNN_ORACLE_MAX_WIDTH) return -2;
  if (tile_bits < 0 || tile_bits > 30) return -3;
  if (counters && !labels) return -4;
  const int n_cls = dims[depth];
  int64_t bit_err = 0, frame_err = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel reduction(+ : bit_err, frame_err)
  {
    float *ping = (float *)malloc(sizeof(float) * FNN_ORACLE_MAX_WIDTH);
    float *pong = (float *)malloc(sizeof(float) * FNN_ORACLE_MAX_WIDTH);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
      const float *cur = x + (size_t)r * dims[0];
      float *nxt = ping;
      for (int l = 0; l < depth; ++l) {
        dense_tanh_row(cur, dims[l], W[l], b[l], dims[l + 1],
                       korder ? korder[l] : NULL, relu, nxt);
        cur = nxt;
        nxt = (nxt == ping) ? pong : ping;
      }
      /* first maximum (strict >), src/fnn_viterbi.py:607 + np.argmax rule */
      int best = 0;
      float bv = cur[0];
      for (int i = 1; i < n_cls; ++i)
        if (cur[i] > bv) { bv = cur[i]; best = i; }
      if (cls_out) cls_out[r] = best;
      if (act_out) memcpy(act_out + (size_t)r * n_cls, cur, sizeof(float) * n_cls);
      if (bits_out)
        for (int t = 0; t < tile_bits; ++t)
          bits_out[(size_t)r * tile_bits + t] =
              (uint8_t)((best >> (tile_bits - 1 - t)) & 1);
      if (labels) {
        const uint32_t diff = (uint32_t)best ^ (uint32_t)labels[r];
        bit_err += __builtin_popcount(diff);
        frame_err += (diff != 0);
      }
    }
    free(ping);
    free(pong);
  }
  if (counters) {
    counters[0] = bit_err;
    counters[1] = frame_err;
    counters[2] = n_rows * (int64_t)tile_bits;
    counters[3] = n_rows;
  }
  return 0;
}

int fnn_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
