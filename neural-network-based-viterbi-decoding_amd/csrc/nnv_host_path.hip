// The reference's REAL call shape: a host `test_set` in, host predictions out
//   predictions = predictions_tensor.eval(feed_dict={x: test_set})      src/fnn_viterbi.py:606-610
// as one synchronous C-ABI call over host buffers (nnv_mlp_decode_host_f32).
//
// The rows travel through a persistent ring of page-locked staging slots that belongs to the
// handle (allocated on first use, reused by every later call):
//
//   host threads   : x_host chunk c+1  --memcpy (T threads)-->  pinned slot
//   stream "in"    : pinned slot of chunk c  --DMA-->  device slot
//   stream "cmp"   : fused decode kernel over the device slot (classes + error counters)
//   stream "out"   : classes  --DMA-->  pinned slot ; host threads copy (widen) them to the caller
//
// A plain hipMemcpy from pageable memory stages through the runtime's own pinned buffer on ONE
// thread (~30 GB/s measured); here the pageable -> pinned copy is spread over T threads and runs
// while the previous chunk is on the link and the one before that in the kernel.
// Measured (tools/host_path_bench.py, 10 M blocks = 5.12 GB in, 0.32 GB out): 80-90 M blocks/s =
// 44-49 GB/s moved, against 56.5 GB/s for a bare pinned H2D of the same rows; flat from 4 staging
// threads up (the CPU share's path to DRAM carries source read + staging write + DMA read).  Handing
// the caller's pageable rows to hipMemcpyAsync chunk by chunk instead (the runtime pins and unpins
// each chunk in place) ran at 61-71 M blocks/s in the same harness, so the staged ring stays.
// Round 3 probed the remaining route, page-locking the caller's own pages ahead of the DMA
// (tools/host_register_probe.py, profiles/r03/host_register_probe.json): hipHostRegister itself is lazy on
// this driver (>400 GB/s "registered"), and the FIRST DMA out of freshly registered pageable pages moves
// 21.8 GB/s (a second pass over the same pages 54.9, hipHostMalloc memory 56.3): every call presents fresh
// pages, so zero-copy cannot beat the 44-49 GB/s of the staged ring.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <emmintrin.h>
#include <sched.h>
#include <stdlib.h>

#include "nnv_internal.h"

namespace nnv {
namespace {

constexpr int kSlots = 4;

// Minimal fork-join pool: run(f) calls f(t) for t in [0, T) on T - 1 workers + the caller.
class Pool {
 public:
  explicit Pool(int n) : n_(n < 1 ? 1 : n) {
    for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { loop(t); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
  }
  int size() const { return n_; }
  void run(const std::function<void(int)> &f) {
    {
      std::lock_guard<std::mutex> l(mu_);
      job_ = &f;
      pending_ = n_ - 1;
      ++gen_;
    }
    cv_.notify_all();
    f(0);
    std::unique_lock<std::mutex> l(mu_);
    done_.wait(l, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)> *job;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        job = job_;
      }
      (*job)(t);
      {
        std::lock_guard<std::mutex> l(mu_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *job_ = nullptr;
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

int default_threads() {
  static const int cached = [] {   // the environment and the affinity mask are read once per process
    int n = static_cast<int>(std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, CPU_COUNT(&set));
    if (const char *e = nnv::ab_env("NNV_HOST_THREADS")) n = atoi(e);   // A/B builds only; the ABI's n_threads argument is the control
    return std::max(1, std::min(n, 8));   // 8 staging threads saturate the host DRAM beside the DMA engine
  }();
  return cached;
}

// pageable -> pinned copy with non-temporal stores.  glibc's memcpy switches to streaming stores only
// above ~3/4 of the shared cache per thread (tens of MB); a staging slice is a few MB, so plain memcpy
// reads every destination line before overwriting it (read-for-ownership), and the staging traffic
// competes with the DMA engine for the same DRAM: measured 74 M blocks/s with 8 threads of memcpy
// against 78 M with 4 (whose 32 MB slices did stream).  `dst` must be 16-byte aligned.
void stream_copy(void *dst, const void *src, size_t bytes) {
  char *d = static_cast<char *>(dst);
  const char *s = static_cast<const char *>(src);
  if ((reinterpret_cast<uintptr_t>(d) & 15) != 0 || bytes < 256) {
    memcpy(d, s, bytes);
    return;
  }
  size_t i = 0;
  for (; i + 64 <= bytes; i += 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 32));
    const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 48));
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i), a);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i + 48), e);
  }
  if (i < bytes) memcpy(d + i, s + i, bytes - i);
  _mm_sfence();  // the DMA engine reads this memory next
}

void split(int64_t total, int parts, int t, int64_t align, int64_t *lo, int64_t *hi) {
  const int64_t per = ((total + parts - 1) / parts + align - 1) / align * align;
  *lo = std::min<int64_t>(total, per * t);
  *hi = std::min<int64_t>(total, per * (t + 1));
}

}  // namespace

struct HostPathCtx {
  std::mutex mu;  // one host-path call at a time per handle
  int64_t chunk_rows = 0;
  int n_in = 0;
  bool with_labels = false;
  hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
  hipEvent_t ev_in[kSlots] = {}, ev_cmp[kSlots] = {}, ev_out[kSlots] = {};
  float *px[kSlots] = {};     // pinned
  int32_t *pl[kSlots] = {};   // pinned labels
  int32_t *pc[kSlots] = {};   // pinned classes
  float *dx[kSlots] = {};     // device
  int32_t *dl[kSlots] = {};
  int32_t *dc[kSlots] = {};
  unsigned long long *dcnt = nullptr;
  // the staging threads live as long as the handle (a pool per call made thread creation the
  // dominant cost of the reference's small test sets); rebuilt only when another thread count is asked for
  Pool *pool = nullptr;
  Pool *pool_for(int threads) {
    if (pool && pool->size() != threads) {
      delete pool;
      pool = nullptr;
    }
    if (!pool) pool = new (std::nothrow) Pool(threads);
    return pool;
  }

  void release() {
    delete pool;
    pool = nullptr;
    for (int s = 0; s < kSlots; ++s) {
      if (px[s]) (void)hipHostFree(px[s]);
      if (pl[s]) (void)hipHostFree(pl[s]);
      if (pc[s]) (void)hipHostFree(pc[s]);
      if (dx[s]) (void)hipFree(dx[s]);
      if (dl[s]) (void)hipFree(dl[s]);
      if (dc[s]) (void)hipFree(dc[s]);
      if (ev_in[s]) (void)hipEventDestroy(ev_in[s]);
      if (ev_cmp[s]) (void)hipEventDestroy(ev_cmp[s]);
      if (ev_out[s]) (void)hipEventDestroy(ev_out[s]);
      px[s] = nullptr; pl[s] = nullptr; pc[s] = nullptr; dx[s] = nullptr; dl[s] = nullptr; dc[s] = nullptr;
      ev_in[s] = ev_cmp[s] = ev_out[s] = nullptr;
    }
    if (dcnt) (void)hipFree(dcnt);
    dcnt = nullptr;
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_cmp) (void)hipStreamDestroy(s_cmp);
    if (s_out) (void)hipStreamDestroy(s_out);
    s_in = s_cmp = s_out = nullptr;
    chunk_rows = 0;
  }

  // (re)allocate for this chunk size / label use; persistent across calls otherwise
  nnv_status ensure(int64_t rows, int n_in_, bool labels) {
    if (rows <= chunk_rows && n_in_ == n_in && (!labels || with_labels)) return NNV_OK;
    release();
    bool ok = hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking) == hipSuccess;
    for (int s = 0; s < kSlots && ok; ++s) {
      ok = hipEventCreateWithFlags(&ev_in[s], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ev_cmp[s], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ev_out[s], hipEventDisableTiming) == hipSuccess &&
           hipHostMalloc(reinterpret_cast<void **>(&px[s]), rows * n_in_ * sizeof(float), hipHostMallocDefault) == hipSuccess &&
           hipHostMalloc(reinterpret_cast<void **>(&pc[s]), rows * sizeof(int32_t), hipHostMallocDefault) == hipSuccess &&
           hipMalloc(reinterpret_cast<void **>(&dx[s]), rows * n_in_ * sizeof(float)) == hipSuccess &&
           hipMalloc(reinterpret_cast<void **>(&dc[s]), rows * sizeof(int32_t)) == hipSuccess;
      if (ok && labels)
        ok = hipHostMalloc(reinterpret_cast<void **>(&pl[s]), rows * sizeof(int32_t), hipHostMallocDefault) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&dl[s]), rows * sizeof(int32_t)) == hipSuccess;
    }
    ok = ok && hipMalloc(reinterpret_cast<void **>(&dcnt), 4 * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
      release();
      return NNV_ERR_OUT_OF_MEMORY;
    }
    chunk_rows = rows;
    n_in = n_in_;
    with_labels = labels;
    return NNV_OK;
  }
};

void host_path_release(nnv_mlp_handle *h) {
  if (h && h->host_ctx) {
    h->host_ctx->release();
    delete h->host_ctx;
    h->host_ctx = nullptr;
  }
}

}  // namespace nnv

extern "C" int32_t nnv_mlp_decode_host_f32(nnv_mlp_handle *h, const float *x_host, int64_t n_rows,
                                            int32_t tile_bits, const int32_t *labels_host,
                                            int32_t *cls32_host, int64_t *cls64_host,
                                            uint64_t *counters_host, int64_t chunk_rows,
                                            int32_t n_threads) {
  using nnv::kSlots;
  if (!h || n_rows < 0 || chunk_rows < 0 || n_threads < 0) return NNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return NNV_OK;
  if (!x_host) return NNV_ERR_INVALID_ARGUMENT;
  if (counters_host && !labels_host) return NNV_ERR_INVALID_ARGUMENT;
  if (counters_host && (tile_bits < 1 || tile_bits > 30 || (1LL << tile_bits) < h->shape.n_classes))
    return NNV_ERR_INVALID_ARGUMENT;
  {
    const nnv_status dv = nnv::check_current_device(h->device);
    if (dv != NNV_OK) return dv;
  }
  static std::mutex create_mu;
  {
    const std::lock_guard<std::mutex> l(create_mu);
    if (!h->host_ctx) h->host_ctx = new (std::nothrow) nnv::HostPathCtx;
    if (!h->host_ctx) return NNV_ERR_OUT_OF_MEMORY;
  }
  nnv::HostPathCtx &c = *h->host_ctx;
  const std::lock_guard<std::mutex> call_lock(c.mu);

  const int n_in = h->shape.n_in;
  // default stage: 64 MB of rows (measured on the 10 M-block workload, 4 slots: 64 MB 86-90 M blocks/s,
  // 128-512 MB 79-81 M, 4 MB 38 M: per-chunk launch/enqueue cost below, cache footprint above)
  int64_t m = chunk_rows > 0 ? chunk_rows : std::max<int64_t>(32, (64LL << 20) / (4LL * n_in));
  m = std::min(m, n_rows);
  m = (m + 31) / 32 * 32;  // whole wave tiles per chunk
  const bool labels = labels_host != nullptr;
  {
    const nnv_status st = c.ensure(m, n_in, labels);
    if (st != NNV_OK) return st;
  }
  // a call that fits one stage is copied by the calling thread alone: no hand-off to the staging threads
  const int64_t n_chunks = (n_rows + m - 1) / m;
  const int want_threads = n_chunks == 1 ? 1 : (n_threads > 0 ? std::min(n_threads, 64) : nnv::default_threads());
  nnv::Pool *const pool_ptr = want_threads > 1 ? c.pool_for(want_threads) : nullptr;
  if (want_threads > 1 && !pool_ptr) return NNV_ERR_OUT_OF_MEMORY;
  const int T = want_threads;
  auto run_on_threads = [&](const std::function<void(int)> &f) {
    if (pool_ptr) pool_ptr->run(f);
    else f(0);
  };
  if (hipMemsetAsync(c.dcnt, 0, 4 * sizeof(unsigned long long), c.s_cmp) != hipSuccess) return NNV_ERR_HIP;

  int32_t status = NNV_OK;
  auto chunk_span = [&](int64_t i, int64_t *lo, int64_t *cnt) {
    *lo = i * m;
    *cnt = std::min(m, n_rows - *lo);
  };
  // classes of chunk i: pinned slot -> the caller's arrays (after its D2H has completed)
  auto drain = [&](int64_t i) {
    const int s = static_cast<int>(i % kSlots);
    int64_t lo, cnt;
    chunk_span(i, &lo, &cnt);
    if (hipEventSynchronize(c.ev_out[s]) != hipSuccess) {
      status = NNV_ERR_HIP;
      return;
    }
    if (!cls32_host && !cls64_host) return;
    const int32_t *src = c.pc[s];
    run_on_threads([&](int t) {
      int64_t a, b;
      nnv::split(cnt, T, t, 16, &a, &b);
      if (cls32_host && b > a) memcpy(cls32_host + lo + a, src + a, (b - a) * sizeof(int32_t));
      if (cls64_host)
        for (int64_t r = a; r < b; ++r) cls64_host[lo + r] = src[r];
    });
  };

  for (int64_t i = 0; i < n_chunks && status == NNV_OK; ++i) {
    const int s = static_cast<int>(i % kSlots);
    int64_t lo, cnt;
    chunk_span(i, &lo, &cnt);
    if (i >= kSlots) drain(i - kSlots);  // slot s is free once its previous classes are with the caller
    if (status != NNV_OK) break;
    // pageable -> pinned, T threads (16-float = 64-byte aligned slices)
    run_on_threads([&](int t) {
      int64_t a, b;
      nnv::split(cnt * n_in, T, t, 16, &a, &b);
      if (b > a) nnv::stream_copy(c.px[s] + a, x_host + lo * n_in + a, (b - a) * sizeof(float));
      if (labels) {
        nnv::split(cnt, T, t, 16, &a, &b);
        if (b > a) nnv::stream_copy(c.pl[s] + a, labels_host + lo + a, (b - a) * sizeof(int32_t));
      }
    });
    bool ok = hipMemcpyAsync(c.dx[s], c.px[s], cnt * n_in * sizeof(float), hipMemcpyHostToDevice, c.s_in) == hipSuccess;
    if (ok && labels)
      ok = hipMemcpyAsync(c.dl[s], c.pl[s], cnt * sizeof(int32_t), hipMemcpyHostToDevice, c.s_in) == hipSuccess;
    ok = ok && hipEventRecord(c.ev_in[s], c.s_in) == hipSuccess &&
         hipStreamWaitEvent(c.s_cmp, c.ev_in[s], 0) == hipSuccess;
    if (!ok) {
      status = NNV_ERR_HIP;
      break;
    }
    status = nnv_mlp_decode_f32(h, c.dx[s], cnt, tile_bits, labels ? c.dl[s] : nullptr, c.dc[s], nullptr, nullptr,
                                counters_host ? reinterpret_cast<uint64_t *>(c.dcnt) : nullptr, c.s_cmp);
    if (status != NNV_OK) break;
    ok = hipEventRecord(c.ev_cmp[s], c.s_cmp) == hipSuccess &&
         hipStreamWaitEvent(c.s_out, c.ev_cmp[s], 0) == hipSuccess &&
         hipMemcpyAsync(c.pc[s], c.dc[s], cnt * sizeof(int32_t), hipMemcpyDeviceToHost, c.s_out) == hipSuccess &&
         hipEventRecord(c.ev_out[s], c.s_out) == hipSuccess;
    // No further stream edges are needed: slot s is reused by chunk i + kSlots only after
    // drain(i) has waited on ev_out[s] on the host, i.e. after this chunk's H2D, kernel and D2H.
    if (!ok) status = NNV_ERR_HIP;
  }
  for (int64_t i = std::max<int64_t>(0, n_chunks - kSlots); i < n_chunks && status == NNV_OK; ++i) drain(i);
  if (status != NNV_OK) {  // leave no work in flight that still points at the caller's or our buffers
    (void)hipStreamSynchronize(c.s_in);
    (void)hipStreamSynchronize(c.s_cmp);
    (void)hipStreamSynchronize(c.s_out);
    return status;
  }
  if (counters_host) {
    unsigned long long cnt[4];
    if (hipStreamSynchronize(c.s_cmp) != hipSuccess ||
        hipMemcpy(cnt, c.dcnt, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess)
      return NNV_ERR_HIP;
    for (int k = 0; k < 4; ++k) counters_host[k] += cnt[k];
  }
  return NNV_OK;
}
