/* The training entries of the C ABI used from plain C: no Python, no torch.
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/train_c_abi.c -Iinclude -I/opt/rocm/include \
 *       -Lneural-network-based-viterbi-decoding_amd -lnnviterbi_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/neural-network-based-viterbi-decoding_amd -Wl,-rpath,/opt/rocm/lib -o train_c_abi
 * 1. The reference's default network 14 -> 128 -> 64 -> 32 -> 128 (src/fnn_viterbi.py:27-92) is trained with its own
 *    recipe (softmax cross-entropy on the tanh outputs + beta l2, TF-1 Adam, batch 256, unshuffled passes over one
 *    training set: src/fnn_viterbi.py:549-594) through nnv_train_* on blocks generated on the device at 2 dB; the
 *    epoch's mean cost must fall and the frame error rate of the decode (nnv_mlp_decode_f32, through K1 and through
 *    K1w with identical classes) must end far below the untrained network's.
 * 2. A 32 -> 30 -> 2 ReLU network with the bit-wise decoder's recipe (sigmoid cross-entropy on the ReLU outputs,
 *    src/fnn_viterbi_bitwise.py:438-452) learns the sign of one input through nnv_dense_train_*.
 * Exit status 0 = every check held. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nnviterbi.h"

#define CHECK_HIP(e) do { if ((e) != hipSuccess) { fprintf(stderr, "HIP error line %d\n", __LINE__); return 2; } } while (0)
#define CHECK_NNV(e) do { int s_ = (e); if (s_ != NNV_OK) { fprintf(stderr, "%s (line %d)\n", nnv_status_string(s_), __LINE__); return 3; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform01(void) {   /* xorshift64*: good enough to draw initial weights */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static float gauss(void) {
  const double u = uniform01() + 1e-300, v = uniform01();
  return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
}
static float *init_matrix(int n_in, int n_out) {   /* N(0, 2 / n_in): the scale of initialize_weights, src/fnn_viterbi.py:413-468 */
  float *w = malloc(sizeof(float) * (size_t)n_in * (size_t)n_out);
  const float s = sqrtf(2.0f / (float)n_in);
  for (int i = 0; i < n_in * n_out; ++i) w[i] = s * gauss();
  return w;
}

static int frame_errors(const float *W[4], const float *b[4], const int *dims, const float *x_dev, const int32_t *lab_dev,
                        int64_t n_rows, int tile_bits, uint64_t *frame_err_out) {
  nnv_mlp_desc net;
  memset(&net, 0, sizeof(net));
  net.depth = 4;
  for (int l = 0; l <= 4; ++l) net.dims[l] = dims[l];
  for (int l = 0; l < 4; ++l) { net.W_host[l] = W[l]; net.b_host[l] = b[l]; }
  nnv_mlp_handle *h = NULL;
  CHECK_NNV(nnv_mlp_prepack(&net, &h));
  int32_t *cls_a = NULL, *cls_b = NULL;
  uint64_t *cnt = NULL;
  CHECK_HIP(hipMalloc((void **)&cls_a, (size_t)n_rows * 4));
  CHECK_HIP(hipMalloc((void **)&cls_b, (size_t)n_rows * 4));
  CHECK_HIP(hipMalloc((void **)&cnt, 32));
  CHECK_HIP(hipMemset(cnt, 0, 32));
  CHECK_NNV(nnv_mlp_set_path(h, NNV_MLP_PATH_PERSISTENT));   /* K1 */
  CHECK_NNV(nnv_mlp_decode_f32(h, x_dev, n_rows, tile_bits, lab_dev, cls_a, NULL, NULL, cnt, NULL));
  CHECK_NNV(nnv_mlp_set_path(h, NNV_MLP_PATH_SPLIT));        /* K1w: the same classes bit for bit */
  CHECK_NNV(nnv_mlp_decode_f32(h, x_dev, n_rows, tile_bits, NULL, cls_b, NULL, NULL, NULL, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  int32_t *ha = malloc((size_t)n_rows * 4), *hb = malloc((size_t)n_rows * 4);
  uint64_t c[4];
  CHECK_HIP(hipMemcpy(ha, cls_a, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(hb, cls_b, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(c, cnt, 32, hipMemcpyDeviceToHost));
  if (memcmp(ha, hb, (size_t)n_rows * 4) != 0) { fprintf(stderr, "K1 and K1w disagree\n"); return 4; }
  *frame_err_out = c[1];
  free(ha); free(hb);
  CHECK_HIP(hipFree(cls_a)); CHECK_HIP(hipFree(cls_b)); CHECK_HIP(hipFree(cnt));
  CHECK_NNV(nnv_mlp_free(h));
  return 0;
}

int main(void) {
  if (nnv_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 1; }
  /* ---------------- 1. the feed-forward decoder's step (K7) ---------------- */
  enum { B = 7, BATCH = 256, STEPS = 250, EPOCHS = 40 };
  const int dims[5] = {2 * B, 128, 64, 32, 1 << B};
  const int64_t n_train = (int64_t)BATCH * STEPS, n_test = 64000;
  const float sigma = 0.5616748812614791f;   /* 2 dB, src/fnn_viterbi.py:89-92 */
  float *x_train, *x_test;
  int32_t *l_train, *l_test;
  CHECK_HIP(hipMalloc((void **)&x_train, (size_t)n_train * 2 * B * 4));
  CHECK_HIP(hipMalloc((void **)&l_train, (size_t)n_train * 4));
  CHECK_HIP(hipMalloc((void **)&x_test, (size_t)n_test * 2 * B * 4));
  CHECK_HIP(hipMalloc((void **)&l_test, (size_t)n_test * 4));
  CHECK_NNV(nnv_awgn_generate_f32(NNV_CODE_133171, B, n_train, 0, 1, &sigma, 1, 11, x_train, l_train, NULL));
  CHECK_NNV(nnv_awgn_generate_f32(NNV_CODE_133171, B, n_test, 0, 1, &sigma, 1, 12, x_test, l_test, NULL));
  float *W[4], *b[4];
  for (int l = 0; l < 4; ++l) { W[l] = init_matrix(dims[l], dims[l + 1]); b[l] = calloc((size_t)dims[l + 1], 4); }
  uint64_t fe_before = 0, fe_after = 0;
  int rc = frame_errors((const float **)W, (const float **)b, dims, x_test, l_test, n_test, B, &fe_before);
  if (rc) return rc;

  nnv_train_desc td;
  memset(&td, 0, sizeof(td));
  td.n_input = dims[0]; td.n_classes = dims[4]; td.batch_size = BATCH;
  td.n_hidden[0] = 128; td.n_hidden[1] = 64; td.n_hidden[2] = 32;
  td.learning_rate = 1e-3f;            /* the reference's 1e-4 over 200 epochs, hurried for an example */
  td.beta_regularization = 1e-3f; td.keep_prob = 1.0f;
  for (int l = 0; l < 4; ++l) { td.W_host[l] = W[l]; td.b_host[l] = b[l]; }
  nnv_train_handle *th = NULL;
  CHECK_NNV(nnv_train_create(&td, &th));
  double first = 0.0, last = 0.0;
  for (int e = 0; e < EPOCHS; ++e) {
    CHECK_NNV(nnv_train_steps(th, x_train, l_train, STEPS, NULL, NULL, NULL));
    double cost_sum = 0.0;
    CHECK_NNV(nnv_train_export(th, (float *const *)W, (float *const *)b, &cost_sum));
    last = cost_sum / STEPS;
    if (e == 0) first = last;
  }
  CHECK_NNV(nnv_train_free(th));
  rc = frame_errors((const float **)W, (const float **)b, dims, x_test, l_test, n_test, B, &fe_after);
  if (rc) return rc;
  printf("K7: %d steps, mean cost of the first epoch %.6f, of the last %.6f; frame errors %llu -> %llu of %lld (K1 == K1w)\n",
         EPOCHS * STEPS, first, last, (unsigned long long)fe_before, (unsigned long long)fe_after, (long long)n_test);
  if (!(last < first) || !(fe_after * 4 < fe_before)) { fprintf(stderr, "the network did not learn\n"); return 5; }

  /* ---------------- 2. the bit-wise decoder's recipe (K8) ---------------- */
  enum { NI = 32, NH = 30, NB = 1000, NSTEPS = 400 };
  float *hx = malloc(sizeof(float) * NI * NB * NSTEPS);
  int32_t *hl = malloc(sizeof(int32_t) * NB * NSTEPS);
  for (int r = 0; r < NB * NSTEPS; ++r) {
    for (int k = 0; k < NI; ++k) hx[(size_t)r * NI + k] = gauss();
    hl[r] = hx[(size_t)r * NI] > 0.f ? 1 : 0;
  }
  float *dx;
  int32_t *dl;
  CHECK_HIP(hipMalloc((void **)&dx, sizeof(float) * NI * NB * NSTEPS));
  CHECK_HIP(hipMalloc((void **)&dl, sizeof(int32_t) * NB * NSTEPS));
  CHECK_HIP(hipMemcpy(dx, hx, sizeof(float) * NI * NB * NSTEPS, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dl, hl, sizeof(int32_t) * NB * NSTEPS, hipMemcpyHostToDevice));
  float *V[5] = {init_matrix(NI, NH), init_matrix(NH, 2), NULL, NULL, NULL}, *c[5] = {calloc(NH, 4), calloc(2, 4), NULL, NULL, NULL};
  for (int i = 0; i < NH; ++i) c[0][i] = 0.1f;   /* ReLU units start alive */
  c[1][0] = c[1][1] = 0.1f;
  nnv_dense_train_desc dd;
  memset(&dd, 0, sizeof(dd));
  dd.n_layers = 2; dd.dims[0] = NI; dd.dims[1] = NH; dd.dims[2] = 2;
  dd.activation = NNV_ACT_RELU; dd.batch_size = NB; dd.learning_rate = 2e-3f; dd.beta_regularization = 1e-4f;
  dd.reg_layer_mask = 1u;
  for (int l = 0; l < 2; ++l) { dd.W_host[l] = V[l]; dd.b_host[l] = c[l]; }
  nnv_dense_train_handle *dh = NULL;
  CHECK_NNV(nnv_dense_train_create(&dd, &dh));
  double d_first = 0.0, d_last = 0.0;
  for (int part = 0; part < 4; ++part) {
    CHECK_NNV(nnv_dense_train_steps(dh, dx + (size_t)part * (NSTEPS / 4) * NB * NI, NI, (int64_t)NB * NI,
                                    dl + (size_t)part * (NSTEPS / 4) * NB, NSTEPS / 4, NULL, NULL));
    double cost_sum = 0.0;
    CHECK_NNV(nnv_dense_train_export(dh, (float *const *)V, (float *const *)c, NULL, &cost_sum));
    d_last = cost_sum / (NSTEPS / 4);
    if (part == 0) d_first = d_last;
  }
  CHECK_NNV(nnv_dense_train_free(dh));
  printf("K8: %d steps, mean cost of the first quarter %.6f, of the last %.6f\n", NSTEPS, d_first, d_last);
  if (!(d_last < d_first - 0.02)) { fprintf(stderr, "the dense network did not learn\n"); return 6; }
  printf("ok\n");
  return 0;
}
