/* The drop-in boundary used from plain C: no Python, no torch.
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/decode_c_abi.c -Iinclude -I/opt/rocm/include \
 *       -Lneural-network-based-viterbi-decoding_amd -lnnviterbi_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/neural-network-based-viterbi-decoding_amd -Wl,-rpath,/opt/rocm/lib -o decode_c_abi
 * A matched-filter network of the reference's architecture (pass-through hidden layers, class layer =
 * correlation with every codeword of the (133,171) b = 8 tile code, src/fnn_viterbi.py:157-217) decodes
 * blocks generated on the device: at sigma = 0 every block must come back exactly; at 2 dB the BER of
 * the per-tile maximum-likelihood decision is printed.  Then the same rows through the HOST entry
 * (nnv_mlp_decode_host_f32: host arrays in / out), the device rule, and a three-position bit-wise bank.
 * Exit status 0 = the round trip was exact and every other check held. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nnviterbi.h"

#define CHECK_HIP(e) do { if ((e) != hipSuccess) { fprintf(stderr, "HIP error line %d\n", __LINE__); return 2; } } while (0)
#define CHECK_NNV(e) do { int s_ = (e); if (s_ != NNV_OK) { fprintf(stderr, "%s (line %d)\n", nnv_status_string(s_), __LINE__); return 3; } } while (0)

/* zero-state (133,171) encoder, output order g=133 first (taps {0,2,3,5,6} and {0,1,2,3,6}) */
static void encode_133171(const int *u, int n, int *out) {
  for (int t = 0; t < n; ++t) {
#define U(i) ((i) < 0 ? 0 : u[i])
    out[2 * t] = (U(t) + U(t - 2) + U(t - 3) + U(t - 5) + U(t - 6)) & 1;
    out[2 * t + 1] = (U(t) + U(t - 1) + U(t - 2) + U(t - 3) + U(t - 6)) & 1;
#undef U
  }
}

int main(void) {
  enum { B = 8, NIN = 16, C = 256, H1 = 128, H2 = 64, H3 = 32 };
  const int64_t n_rows = 8 * 1000 * 1000;  /* 1 M blocks of 8 tiles */
  if (nnv_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 1; }

  /* weights [in][out], the reference's layout (src/fnn_viterbi.py:432-456) */
  float *W1 = calloc(NIN * H1, 4), *W2 = calloc(H1 * H2, 4), *W3 = calloc(H2 * H3, 4), *W4 = calloc(H3 * C, 4);
  float *b1 = calloc(H1, 4), *b2 = calloc(H2, 4), *b3 = calloc(H3, 4), *b4 = calloc(C, 4);
  for (int i = 0; i < NIN; ++i) { W1[i * H1 + i] = 0.05f; W2[i * H2 + i] = 1.0f; W3[i * H3 + i] = 1.0f; }
  for (int c = 0; c < C; ++c) {
    int u[B], e[2 * B];
    for (int t = 0; t < B; ++t) u[t] = (c >> (B - 1 - t)) & 1;  /* MSB first, src/fnn_viterbi.py:281-285 */
    encode_133171(u, B, e);
    for (int k = 0; k < NIN; ++k) W4[k * C + c] = e[k] ? 2.0f : -2.0f;
  }
  nnv_mlp_desc net;
  memset(&net, 0, sizeof(net));
  net.depth = 4;
  const int dims[5] = {NIN, H1, H2, H3, C};
  for (int i = 0; i < 5; ++i) net.dims[i] = dims[i];
  net.W_host[0] = W1; net.W_host[1] = W2; net.W_host[2] = W3; net.W_host[3] = W4;
  net.b_host[0] = b1; net.b_host[1] = b2; net.b_host[2] = b3; net.b_host[3] = b4;
  nnv_mlp_handle *h = NULL;
  CHECK_NNV(nnv_mlp_prepack(&net, &h));

  float *x = NULL;
  int32_t *labels = NULL, *cls = NULL;
  uint64_t *counters = NULL, host[4];
  CHECK_HIP(hipMalloc((void **)&x, n_rows * NIN * sizeof(float)));
  CHECK_HIP(hipMalloc((void **)&labels, n_rows * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void **)&cls, n_rows * sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void **)&counters, 4 * sizeof(uint64_t)));

  int exact = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const float sigma = pass == 0 ? 0.0f : 0.5616748812614791f;  /* noiseless, then 2 dB */
    CHECK_HIP(hipMemset(counters, 0, 4 * sizeof(uint64_t)));
    CHECK_NNV(nnv_awgn_generate_f32(NNV_CODE_133171, B, n_rows, 0, 8, &sigma, 1, 2024, x, labels, NULL));
    CHECK_NNV(nnv_mlp_decode_f32(h, x, n_rows, B, labels, cls, NULL, NULL, counters, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(host, counters, sizeof(host), hipMemcpyDeviceToHost));
    printf("sigma %.4f: bit errors %llu / %llu  (BER %.3e), tile errors %llu / %llu\n", sigma,
           (unsigned long long)host[0], (unsigned long long)host[2], (double)host[0] / (double)host[2],
           (unsigned long long)host[1], (unsigned long long)host[3]);
    if (pass == 0) exact = host[0] == 0 && host[1] == 0 && host[2] == (uint64_t)n_rows * B && host[3] == (uint64_t)n_rows;
  }
  /* ---- the reference's own call shape: host rows in, host predictions out (src/fnn_viterbi.py:606-610) ---- */
  int host_ok = 0;
  {
    const int64_t m = 1000 * 1000 + 7;                      /* ragged on purpose */
    float *xh = malloc((size_t)m * NIN * sizeof(float));
    int32_t *lh = malloc((size_t)m * sizeof(int32_t)), *dev_cls = malloc((size_t)m * sizeof(int32_t));
    int64_t *ch = malloc((size_t)m * sizeof(int64_t));
    uint64_t cnt[4] = {0, 0, 0, 0};
    CHECK_HIP(hipMemcpy(xh, x, (size_t)m * NIN * sizeof(float), hipMemcpyDeviceToHost));   /* the 2 dB rows of pass 1 */
    CHECK_HIP(hipMemcpy(lh, labels, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(dev_cls, cls, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    CHECK_NNV(nnv_mlp_decode_host_f32(h, xh, m, B, lh, NULL, ch, cnt, 100000, 3));
    int64_t diff = 0;
    for (int64_t i = 0; i < m; ++i) diff += ch[i] != dev_cls[i];
    printf("host entry: %lld rows, %lld differ from the resident decode, bit errors %llu / %llu\n", (long long)m,
           (long long)diff, (unsigned long long)cnt[0], (unsigned long long)cnt[2]);
    host_ok = diff == 0 && cnt[2] == (uint64_t)m * B && cnt[3] == (uint64_t)m;
    free(xh); free(lh); free(dev_cls); free(ch);
  }
  /* ---- the other arithmetic modes (nnv_mlp_prepack_ex) on the same 2 dB rows: f16x3 keeps the 1e-5 parity rules,
   *      f16 (one rounding to f16 per operand) is the opt-in fast mode outside them ---- */
  int modes_ok = 1;
  {
    int32_t *ref = malloc((size_t)n_rows * sizeof(int32_t)), *got = malloc((size_t)n_rows * sizeof(int32_t));
    CHECK_HIP(hipMemcpy(ref, cls, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int32_t modes[2] = {NNV_PRECISION_F16X3, NNV_PRECISION_F16};
    const char *names[2] = {"f16x3", "f16"};
    for (int k = 0; k < 2; ++k) {
      nnv_mlp_handle *hm = NULL;
      CHECK_NNV(nnv_mlp_prepack_ex(&net, modes[k], &hm));
      CHECK_HIP(hipMemset(counters, 0, 4 * sizeof(uint64_t)));
      CHECK_NNV(nnv_mlp_decode_f32(hm, x, n_rows, B, labels, cls, NULL, NULL, counters, NULL));
      CHECK_HIP(hipDeviceSynchronize());
      CHECK_HIP(hipMemcpy(got, cls, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
      CHECK_HIP(hipMemcpy(host, counters, sizeof(host), hipMemcpyDeviceToHost));
      int64_t diff = 0;
      for (int64_t i = 0; i < n_rows; ++i) diff += got[i] != ref[i];
      printf("mode %s: %lld of %lld classes differ from the fp32 mode, BER %.3e\n", names[k], (long long)diff,
             (long long)n_rows, (double)host[0] / (double)host[2]);
      modes_ok = modes_ok && diff <= (k == 0 ? 8 : n_rows / 100);   /* f16x3: near-ties only */
      CHECK_NNV(nnv_mlp_free(hm));
    }
    free(ref); free(got);
  }
  /* ---- the device rule: a handle is refused on any device but the one it was packed for ---- */
  int dev = -1;
  CHECK_HIP(hipGetDevice(&dev));
  const int rule_ok = nnv_check_device(dev) == NNV_OK && nnv_check_device(dev + 1) == NNV_ERR_INVALID_ARGUMENT;
  printf("device rule: current device %d ok, device %d refused: %s\n", dev, dev + 1, rule_ok ? "yes" : "NO");

  /* ---- bit-wise decoder bank: 3 positions of a 2(6+10) -> 30 -> 2 ReLU network in one launch
   *      (src/fnn_viterbi_bitwise.py:326-399,503-505); network p votes bit 1 iff soft value 12 + 2p > 0 ---- */
  int bank_ok = 0;
  {
    enum { K = 32, HID = 30, P = 3, NR = 5000 };
    float *w1 = calloc(P * K * HID, 4), *bb1 = calloc(P * HID, 4), *wo = calloc(P * HID * 2, 4), *bo = calloc(P * 2, 4);
    for (int p = 0; p < P; ++p) {
      w1[(p * K + 12 + 2 * p) * HID + 0] = 1.0f;            /* hidden unit 0 = relu(+x_k) */
      w1[(p * K + 12 + 2 * p) * HID + 1] = -1.0f;           /* hidden unit 1 = relu(-x_k) */
      wo[(p * HID + 0) * 2 + 1] = 1.0f;                     /* logit 1 <- unit 0 */
      wo[(p * HID + 1) * 2 + 0] = 1.0f;                     /* logit 0 <- unit 1 */
    }
    nnv_bitwise_desc bd = {K, HID, P, w1, bb1, wo, bo};
    nnv_bitwise_handle *bh = NULL;
    CHECK_NNV(nnv_bitwise_prepack(&bd, &bh));
    float *xs = malloc(NR * K * sizeof(float)), *xd = NULL;
    uint8_t *bits_d = NULL, *bits_h = malloc(P * NR);
    for (int i = 0; i < NR * K; ++i) xs[i] = (float)((i * 2654435761u >> 7) % 7) - 3.0f;   /* -3 .. 3, zeros included */
    CHECK_HIP(hipMalloc((void **)&xd, NR * K * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&bits_d, P * NR));
    CHECK_HIP(hipMemcpy(xd, xs, NR * K * sizeof(float), hipMemcpyHostToDevice));
    CHECK_NNV(nnv_bitwise_decode_f32(bh, xd, NR, bits_d, NULL, NULL, NULL, NULL));
    CHECK_HIP(hipMemcpy(bits_h, bits_d, P * NR, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int p = 0; p < P; ++p)
      for (int r = 0; r < NR; ++r) bad += bits_h[p * NR + r] != (xs[r * K + 12 + 2 * p] > 0.0f);   /* ties at 0 -> bit 0 */
    printf("bit-wise bank: %d positions x %d rows, %d wrong\n", P, NR, bad);
    bank_ok = bad == 0;
    CHECK_NNV(nnv_bitwise_free(bh));
    CHECK_NNV(nnv_bitwise_prepack_ex(&bd, NNV_PRECISION_F16X3, &bh));   /* the same bank on the 16-bit matrix pipe */
    CHECK_NNV(nnv_bitwise_decode_f32(bh, xd, NR, bits_d, NULL, NULL, NULL, NULL));
    CHECK_HIP(hipMemcpy(bits_h, bits_d, P * NR, hipMemcpyDeviceToHost));
    bad = 0;
    for (int p = 0; p < P; ++p)
      for (int r = 0; r < NR; ++r) bad += bits_h[p * NR + r] != (xs[r * K + 12 + 2 * p] > 0.0f);
    printf("bit-wise bank f16x3: %d wrong\n", bad);
    bank_ok = bank_ok && bad == 0;
    CHECK_NNV(nnv_bitwise_free(bh));
    hipFree(xd); hipFree(bits_d);
    free(w1); free(bb1); free(wo); free(bo); free(xs); free(bits_h);
  }
  CHECK_NNV(nnv_mlp_free(h));
  hipFree(x); hipFree(labels); hipFree(cls); hipFree(counters);
  return (exact && host_ok && rule_ok && bank_ok && modes_ok) ? 0 : 4;
}
