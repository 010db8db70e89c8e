// Host side of the fused decode kernel (nnv_mlp_decode.hip): lookup in the instantiation tables of
// its two translation units, launch plan, per-(device, kernel) LDS opt-in, launch.
#include <mutex>

#include "nnv_internal.h"

namespace nnv {

// nnv_mlp_decode.hip compiled with -DNNV_TU_PREC=0 / =1
extern const KernelEntry kKernelsF32[];
extern const int kNumKernelsF32;
extern const KernelEntry kKernelsF16x3[];
extern const int kNumKernelsF16x3;

namespace {

constexpr int kMaxKernels = 512;   // bound of the per-device attribute cache (both tables together)

// index of an entry in the concatenation [f32 table | f16x3 table]
int kernel_index(const KernelEntry *k) {
  if (k >= kKernelsF32 && k < kKernelsF32 + kNumKernelsF32) return static_cast<int>(k - kKernelsF32);
  return kNumKernelsF32 + static_cast<int>(k - kKernelsF16x3);
}

// lean: 0 general, 1 lean, 2 lean with per-SNR-bin counters (ignored for streamed / debug kernels)
const KernelEntry *find_kernel(const PackedShape &s, int nt, bool act, int lean = 0) {
  const int mta = s.mt[0];
  const int mtb = s.nhid >= 2 ? s.mt[1] : 1;
  const int mtc = s.nhid >= 3 ? s.mt[2] : 1;
  const int variant = (act || s.streams != 0) ? kVarStream : (lean == 2 ? kVarLeanBins : (lean == 1 ? kVarLean : kVarGeneral));
  const KernelEntry *tab = s.precision == kPrecF32 ? kKernelsF32 : kKernelsF16x3;  // second table: both 16-bit modes
  const int n = s.precision == kPrecF32 ? kNumKernelsF32 : kNumKernelsF16x3;
  for (int i = 0; i < n; ++i) {
    const KernelEntry &k = tab[i];
    if (k.prec == s.precision && k.g1 == s.g1 && k.nhid == s.nhid && k.mta == mta && k.mtb == mtb &&
        k.mtc == mtc && k.nt == nt && k.act == act && k.variant == variant)
      return &k;
  }
  return nullptr;
}

// Workgroup size: 1024 threads = 4 waves per SIMD (<= 128 VGPRs) when that
// instantiation does not spill (more than a handful of registers), else 512 (2 per SIMD).  (A/B builds only:
// NNV_THREADS overrides.)
int pick_threads_impl(const PackedShape &s) {
  static const int forced = [] {
    const char *e = ab_env("NNV_THREADS");
    const int v = e ? atoi(e) : 0;
    return (v == 256 || v == 512 || v == 768 || v == 1024) ? v : 0;
  }();
  if (forced && find_kernel(s, forced, false)) return forced;
  for (int nt : {1024, 768}) {  // most waves per SIMD whose instantiation does not spill
    const KernelEntry *k = find_kernel(s, nt, false);
    if (!k) continue;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k->fn)) != hipSuccess) continue;
    // up to 8 spilled dwords cost less than a wave per SIMD (since round 4 the stream kernels with >= 2 hidden
    // layers spill 0-2 dwords at 1024 threads; the 1-hidden-layer ones hold 64 activation registers, spill
    // 13-139 and fall through to 768 / 512)
    if (attr.localSizeBytes <= 32u) return nt;
  }
  return 512;
}


int64_t keep_slot_bytes(int threads) { return static_cast<int64_t>(threads) * kKeepSlotBytes; }

}  // namespace

nnv_status decode_supported(const PackedShape &s) {
  return find_kernel(s, 512, false) ? NNV_OK : NNV_ERR_UNSUPPORTED_SHAPE;
}

int decode_pick_threads(const PackedShape &s) { return pick_threads_impl(s); }

int decode_kernel_vgprs(const PackedShape &s, int threads, bool lean) {
  const KernelEntry *k = find_kernel(s, threads, false, lean ? 1 : 0);
  if (!k) return 0;
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k->fn)) != hipSuccess) return 0;
  return attr.numRegs;
}

// Resident or streamed?  The whole network stays in LDS (general / lean variants) iff it fits beside the
// argmax winner slots of the workgroup size pick_threads_impl chooses for the resident variant (1024 threads =
// 64 KB of slots unless that instantiation spills); otherwise as many class tiles as fit stay resident and every
// wave streams the rest from L2 (stream variant) at ITS workgroup size.  In between sits a band -- whole network
// 96..128 KB, e.g. 18->128->64->32->512 (120 KB packed) -- that fits resident beside 512 threads' 32 KB of slots
// only.  Measured, round 4, one box, 18->128->64->32->512 over 4 M rows (tools/bench_band.py):
//   stream variant at 1024 threads (every wave streams the class tiles that do not fit): fp32 1.95-1.97 ms, f16x3 0.834-0.844 ms
//   whole network resident at 512 threads                                              : fp32 1.98-2.01 ms, f16x3 0.893-0.894 ms
// so the band is STREAMED (earlier in round 4, against round 3's lock-step ring -- 2.04-2.05 / 1.02-1.03 ms -- the
// resident choice had won).  (A/B builds only: NNV_BAND_RESIDENT_512=1 forces the resident choice.)
int decode_configure(PackedShape *s, int lds_limit) {
  static const bool band_resident = ab_env("NNV_BAND_RESIDENT_512") != nullptr;
  const int64_t whole = (s->hid_floats + s->out_floats + s->bias_floats) * 4 + kLdsTailBytes;
  const int64_t hidden_bias_floats = s->bias_floats - 32LL * s->mt_out;   // the stream variant keeps only these + its resident tiles' in LDS
  s->streams = 0;
  const int t = pick_threads_impl(*s);
  if (whole + keep_slot_bytes(t) <= lds_limit) return t;
  if (band_resident && t > 512 && whole + keep_slot_bytes(512) <= lds_limit) return 512;
  s->streams = 1;
  // streamed: hidden layers + their biases + winner slots must fit (the class layer's biases follow its tiles: resident
  // ones in LDS, streamed ones from global memory)
  int ts = pick_threads_impl(*s);
  const int64_t fixed = (s->hid_floats + hidden_bias_floats) * 4 + kLdsTailBytes;
  if (fixed + keep_slot_bytes(ts) > lds_limit && ts > 512 && fixed + keep_slot_bytes(512) <= lds_limit) ts = 512;
  return ts;
}

nnv_status decode_plan(const PackedShape &s, int64_t n_rows, int num_cus,
                       int lds_limit, int threads, LaunchPlan *plan) {
  if (!find_kernel(s, 512, false)) return NNV_ERR_UNSUPPORTED_SHAPE;
  // hidden layers + their biases; a class tile then costs its weights + 128 bytes of bias (resident variants hold
  // every tile's bias in LDS, the stream variant only its resident tiles')
  const int64_t hidden_bias_floats = s.bias_floats - 32LL * s.mt_out;
  const int64_t base_floats = s.hid_floats + hidden_bias_floats + (s.streams ? 0 : 32LL * s.mt_out);
  // Every variant parks the running argmax winner's accumulators in LDS, 64 bytes per thread; the rest of the LDS
  // holds as many class-layer M-tiles as fit (all of them unless this is the stream variant).
  const int64_t scratch = keep_slot_bytes(threads);
  if (base_floats * 4 + kLdsTailBytes + scratch > lds_limit) return NNV_ERR_UNSUPPORTED_SHAPE;
  // as many class-layer M-tiles as fit stay in LDS; every wave streams the remaining ones from L2
  const int64_t per_tile_bytes = static_cast<int64_t>(class_tile_kb(s.precision, s.kt_out)) * 1024 + (s.streams ? 128 : 0);
  int64_t fit = (lds_limit - kLdsTailBytes - scratch - base_floats * 4) / per_tile_bytes;
  if (fit > s.mt_out) fit = s.mt_out;
  if (s.streams) {  // A/B builds only: cap the resident class tiles of the stream variant
    static const int cap = [] { const char *e = ab_env("NNV_STREAM_MAX_RESIDENT"); return e ? atoi(e) : -1; }();
    if (cap >= 0 && fit > cap) fit = cap;
  }
  if (s.streams == 0 && fit < s.mt_out) return NNV_ERR_UNSUPPORTED_SHAPE;  // decode_configure sized it
  plan->out_lds_tiles = static_cast<int32_t>(fit);
  plan->out_in_lds = fit == s.mt_out ? 1 : 0;
  plan->lds_bytes = static_cast<int32_t>(base_floats * 4 + fit * per_tile_bytes + kLdsTailBytes + scratch);
  plan->threads = threads;
  // Persistent: one workgroup per CU.  Small batches are spread over as many CUs as they have
  // tiles (one busy wave per CU, the others find no work and leave) rather than packed 16 tiles to
  // a CU: BASELINE configs[0] (2,000 rows) 107 -> 39 us per launch, 20,000 rows 110 -> 63 us.
  const int64_t n_tiles = (n_rows + kRowsPerTile - 1) / kRowsPerTile;
  int64_t blocks = n_tiles;
  if (blocks > num_cus) blocks = num_cus;
  if (blocks < 1) blocks = 1;
  plan->blocks = static_cast<int32_t>(blocks);
  return NNV_OK;
}

namespace {
// dynamic LDS above the default cap must be opted into, once per (device, kernel): the attribute
// belongs to the function as loaded on ONE device
nnv_status ensure_lds_optin(int device, const KernelEntry *k, int lds_bytes) {
  static int lds_set[kMaxDevices][kMaxKernels] = {};
  static std::mutex mu;  // prepack / launches may come from several host threads
  if (device < 0 || device >= kMaxDevices) return NNV_ERR_INVALID_ARGUMENT;
  const std::lock_guard<std::mutex> lock(mu);
  const int idx = kernel_index(k);
  if (idx < 0 || idx >= kMaxKernels) return NNV_ERR_INVALID_ARGUMENT;
  if (lds_set[device][idx] < lds_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k->fn),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
      return NNV_ERR_HIP;
    lds_set[device][idx] = lds_bytes;
  }
  return NNV_OK;
}
}  // namespace

// Called from prepack (on the handle's device): every attribute a launch of this network needs is set
// here, so that the launches themselves are pure kernel launches (capturable into a HIP graph from the
// first call).
nnv_status decode_prepare(const PackedShape &s, int num_cus, int lds_limit, int threads, int device) {
  LaunchPlan plan;
  const nnv_status st = decode_plan(s, 1 << 20, num_cus, lds_limit, threads, &plan);
  if (st != NNV_OK) return st;
  for (const int which : {0, 1, 2, 3}) {  // decode, debug forward, lean decode, lean decode with SNR bins
    if (which >= 2 && s.streams != 0) continue;
    const KernelEntry *k = find_kernel(s, which == 1 ? 512 : plan.threads, which == 1, which >= 2 ? which - 1 : 0);
    if (!k) return NNV_ERR_UNSUPPORTED_SHAPE;
    const nnv_status e = ensure_lds_optin(device, k, plan.lds_bytes);
    if (e != NNV_OK) return e;
  }
  return NNV_OK;
}

nnv_status decode_launch(const PackedShape &s, const DecodeParams &p,
                         const LaunchPlan &plan, bool write_act, int device,
                         hipStream_t stream) {
  const int nt = write_act ? 512 : plan.threads;
  static const bool no_lean = ab_env("NNV_NO_LEAN") != nullptr;  // A/B builds only
  const bool lean = !no_lean && s.streams == 0 && !write_act && p.xvec && !p.bits && !p.cls64 && p.n_classes % 32 == 0;
  const KernelEntry *k = find_kernel(s, nt, write_act, lean ? (p.n_bins > 1 ? 2 : 1) : 0);
  if (!k) return NNV_ERR_UNSUPPORTED_SHAPE;
  const int lds_bytes = plan.lds_bytes;
  {
    const nnv_status e = ensure_lds_optin(device, k, lds_bytes);
    if (e != NNV_OK) return e;
  }
  int blocks = plan.blocks;
  if (write_act) {  // 8 waves per workgroup in the debug variant
    const int64_t b = (p.n_tiles + 7) / 8;
    blocks = static_cast<int>(b < plan.blocks ? (b < 1 ? 1 : b) : plan.blocks);
  }
  hipLaunchKernelGGL(k->fn, dim3(blocks), dim3(nt), lds_bytes, stream, p);
  return hipGetLastError() == hipSuccess ? NNV_OK : NNV_ERR_HIP;
}

}  // namespace nnv

