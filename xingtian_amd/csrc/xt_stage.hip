// Host-side staging of arriving rollout arrays (SURVEY.md section 8 "next" row f1): the copy of a trajectory's
// uint8 frames from the pageable memory the transport deserialised them into, to the PAGE-LOCKED staging buffer the
// DMA engine reads, is the binding term of the plugin-path throughput (one Python np.copyto at ~20 GB/s in round 2).
// Here it is a persistent pool of native worker threads (non-temporal stores: the destination is only ever read by
// the DMA engine, so it should neither be read-for-ownership nor occupy the cache) that works through the buffer in
// chunks; the calling thread enqueues the hipMemcpyAsync of chunk k as soon as it is staged, so the H2D of chunk k
// runs under the staging of chunk k+1.  ctypes releases the GIL for the duration of the call.
//
// Replaces the host side of the reference's `np.concatenate` + feed_dict upload of the rollout
// (xt/algorithm/ppo/ppo.py:66-71, xt/model/ppo/ppo.py:123-129; learner hand-over xt/framework/learner.py:306-313).
// No arithmetic happens here; no device code in this file.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <emmintrin.h>
#include <mutex>
#include <pthread.h>
#include <string.h>
#include <thread>
#include <vector>

#include "xt_common.h"

namespace xt {
namespace {

constexpr int kMaxWorkers = 16;
constexpr int64_t kDefaultChunk = 256 << 10;     // work unit of the pool (balance over the workers)
constexpr int64_t kDefaultShip = 4 << 20;        // H2D granularity: every hipMemcpyAsync costs ~5-10 us of DMA set-up

// 64 bytes per iteration with streaming stores; head/tail through memcpy.  The final sfence makes the data globally
// visible before the caller hands the range to the DMA engine.
void copy_nt(uint8_t* d, const uint8_t* s, size_t n) {
  size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
  if (head > n) head = n;
  if (head) { memcpy(d, s, head); d += head; s += head; n -= head; }
  const size_t blocks = n / 64;
  for (size_t i = 0; i < blocks; ++i) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 32));
    const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 48));
    _mm_stream_si128(reinterpret_cast<__m128i*>(d), a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(d + 48), e);
    s += 64; d += 64;
  }
  const size_t tail = n - blocks * 64;
  if (tail) memcpy(d, s, tail);
  _mm_sfence();
}

struct Job {
  const uint8_t* src = nullptr;
  uint8_t* dst = nullptr;
  int64_t bytes = 0, chunk = 0;
  int nchunks = 0, nt = 1;
};

class Pool {
 public:
  // Leaked on purpose: workers may outlive static destructors.  NOT inherited across fork(): the worker threads do not
  // exist in the child and the mutexes may have been copied locked, so a child handler drops the pointer and the first
  // call in the child builds a fresh pool (the parent's is simply abandoned there).
  static Pool& get() {
    static std::once_flag once;
    std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { instance().store(nullptr, std::memory_order_release); }); });
    Pool* p = instance().load(std::memory_order_acquire);
    if (!p) {
      Pool* fresh = new Pool();
      if (instance().compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) p = fresh;
      else delete fresh;
    }
    return *p;
  }

  // copies with `threads` participants (the caller is NOT one of them: it pipelines the H2D copies); calls
  // on_chunk(k) on the calling thread for k = 0 .. nchunks-1 in order, each as soon as chunk k is staged.
  template <typename F>
  void run(const Job& job, int threads, F on_chunk) {
    std::lock_guard<std::mutex> serial(call_mu_);
    ensure_workers(threads);
    if ((int)flags_.size() < job.nchunks) flags_ = std::vector<std::atomic<int>>(job.nchunks);
    for (int i = 0; i < job.nchunks; ++i) flags_[i].store(0, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = job;
      active_ = threads;
      const uint64_t e = epoch_.load(std::memory_order_relaxed) + 1;
      next_.store(e << 32, std::memory_order_relaxed);      // chunk tickets carry the job's epoch (see loop())
      epoch_.store(e, std::memory_order_release);
    }
    cv_.notify_all();
    for (int k = 0; k < job.nchunks; ++k) {
      int spins = 0;
      while (!flags_[k].load(std::memory_order_acquire)) {
        if (++spins > 2000) std::this_thread::yield(); else _mm_pause();
      }
      on_chunk(k);
    }
    // all chunks are staged.  A worker that wakes up late still holds this job's epoch: its ticket CAS fails as
    // soon as the next job re-tags the counter, so it can never copy with stale pointers.
  }

 private:
  static std::atomic<Pool*>& instance() { static std::atomic<Pool*> p{nullptr}; return p; }
  void ensure_workers(int n) {
    while ((int)workers_.size() < n && (int)workers_.size() < kMaxWorkers) {
      const int id = (int)workers_.size();
      workers_.emplace_back([this, id] { loop(id); });
      workers_.back().detach();
    }
  }
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      // spin briefly for the next job (an ingest burst is 32 back-to-back calls), then sleep
      uint64_t e = epoch_.load(std::memory_order_acquire);
      if (e == seen) {
        const auto t0 = std::chrono::steady_clock::now();
        while ((e = epoch_.load(std::memory_order_acquire)) == seen) {
          _mm_pause();
          if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150)) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return epoch_.load(std::memory_order_acquire) != seen; });
            e = epoch_.load(std::memory_order_acquire);
            break;
          }
        }
      }
      seen = e;
      Job j;
      int active;
      {
        std::lock_guard<std::mutex> lk(mu_);
        j = job_;
        active = active_;
        if (epoch_.load(std::memory_order_relaxed) != seen) continue;   // a newer job was posted meanwhile
      }
      if (id >= active) continue;
      for (;;) {
        uint64_t v = next_.load(std::memory_order_relaxed);
        int k = -1;
        while ((v >> 32) == (seen & 0xffffffffull) && (int)(v & 0xffffffffull) < j.nchunks) {
          if (next_.compare_exchange_weak(v, v + 1, std::memory_order_relaxed)) { k = (int)(v & 0xffffffffull); break; }
        }
        if (k < 0) break;
        const int64_t off = (int64_t)k * j.chunk;
        const int64_t len = (j.bytes - off) < j.chunk ? (j.bytes - off) : j.chunk;
        if (j.nt) copy_nt(j.dst + off, j.src + off, (size_t)len);
        else memcpy(j.dst + off, j.src + off, (size_t)len);
        flags_[k].store(1, std::memory_order_release);
      }
    }
  }

  std::mutex call_mu_, mu_;
  std::condition_variable cv_;
  std::vector<std::thread> workers_;
  std::vector<std::atomic<int>> flags_;
  std::atomic<uint64_t> epoch_{0};
  std::atomic<uint64_t> next_{0};     // (epoch << 32) | next chunk index
  Job job_;
  int active_ = 0;
};

std::atomic<int> g_threads{4}, g_nt{1};

}  // namespace
}  // namespace xt

extern "C" {

int xt_stage_rows(void* dst_pinned, const void* src, int64_t bytes, void* dev_dst, int64_t chunk_bytes,
                  int64_t ship_bytes, int32_t n_threads, void* stream) {
  XT_REQUIRE(dst_pinned && src && bytes >= 0, "xt_stage_rows: null buffer / negative size");
  if (bytes == 0) return 0;
  if (chunk_bytes <= 0) chunk_bytes = xt::kDefaultChunk;
  if (ship_bytes <= 0) ship_bytes = xt::kDefaultShip;
  if (n_threads < 0) n_threads = xt::g_threads.load();
  if (n_threads > xt::kMaxWorkers) n_threads = xt::kMaxWorkers;
  XT_REQUIRE((bytes + chunk_bytes - 1) / chunk_bytes < (1 << 20), "xt_stage_rows: chunk size too small for %lld bytes",
             (long long)bytes);
  xt::Job j;
  j.src = static_cast<const uint8_t*>(src);
  j.dst = static_cast<uint8_t*>(dst_pinned);
  j.bytes = bytes; j.chunk = chunk_bytes;
  j.nchunks = (int)((bytes + chunk_bytes - 1) / chunk_bytes);
  j.nt = xt::g_nt.load();
  hipError_t err = hipSuccess;
  hipStream_t st = xt::as_stream(stream);
  int64_t shipped = 0;            // [0, shipped) has been handed to the DMA engine
  auto ship = [&](int k) {        // chunks 0..k are staged: ship once ship_bytes have accumulated (or at the end)
    if (!dev_dst || err != hipSuccess) return;
    const int64_t staged = (k + 1 == j.nchunks) ? bytes : (int64_t)(k + 1) * chunk_bytes;
    if (staged - shipped < ship_bytes && staged < bytes) return;
    err = hipMemcpyAsync(static_cast<uint8_t*>(dev_dst) + shipped, j.dst + shipped, (size_t)(staged - shipped),
                         hipMemcpyHostToDevice, st);
    shipped = staged;
  };
  if (n_threads == 0) {       // inline: the calling thread copies chunk by chunk (hosts where a hand-over does not pay)
    for (int k = 0; k < j.nchunks; ++k) {
      const int64_t off = (int64_t)k * chunk_bytes;
      const int64_t len = (bytes - off) < chunk_bytes ? (bytes - off) : chunk_bytes;
      if (j.nt) xt::copy_nt(j.dst + off, j.src + off, (size_t)len); else memcpy(j.dst + off, j.src + off, (size_t)len);
      ship(k);
    }
  } else {
    xt::Pool::get().run(j, n_threads, ship);
  }
  XT_CHECK_HIP(err);
  return 0;
}

int xt_stage_tune(int64_t sample_bytes, float* gbps10) {
  XT_REQUIRE(sample_bytes >= (1 << 16) && sample_bytes <= (1ll << 30), "xt_stage_tune: sample of %lld bytes outside [64 KiB, 1 GiB]",
             (long long)sample_bytes);
  std::vector<uint8_t> src((size_t)sample_bytes), dst((size_t)sample_bytes);
  for (size_t i = 0; i < src.size(); i += 4096) src[i] = (uint8_t)i;   // touch every page
  memset(dst.data(), 1, dst.size());
  // the sample is staged the way an ingest burst does it: in trajectory-sized calls (<= 4 MiB), back to back
  static const int kThreads[5] = {0, 1, 2, 4, 8};
  const int64_t piece = sample_bytes < (4 << 20) ? sample_bytes : (4 << 20);
  float best = 0.f;
  int best_t = 0, best_nt = 0;
  for (int nt = 0; nt < 2; ++nt)
    for (int ti = 0; ti < 5; ++ti) {
      double best_s = 1e30;
      for (int rep = 0; rep < 4; ++rep) {      // rep 0 also spawns / wakes the workers
        const auto t0 = std::chrono::steady_clock::now();
        for (int64_t off = 0; off < sample_bytes; off += piece) {
          xt::Job j;
          j.src = src.data() + off; j.dst = dst.data() + off;
          j.bytes = (sample_bytes - off) < piece ? (sample_bytes - off) : piece; j.chunk = xt::kDefaultChunk;
          j.nchunks = (int)((j.bytes + j.chunk - 1) / j.chunk); j.nt = nt;
          if (kThreads[ti] == 0) {
            if (nt) xt::copy_nt(j.dst, j.src, (size_t)j.bytes); else memcpy(j.dst, j.src, (size_t)j.bytes);
          } else {
            xt::Pool::get().run(j, kThreads[ti], [](int) {});
          }
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep > 0 && s < best_s) best_s = s;
      }
      const float g = (float)((double)sample_bytes / best_s / 1e9);
      if (gbps10) gbps10[nt * 5 + ti] = g;
      if (g > best * 1.05f) { best = g; best_t = kThreads[ti]; best_nt = nt; }    // more threads only for a real gain
    }
  XT_REQUIRE(memcmp(src.data(), dst.data(), src.size()) == 0, "xt_stage_tune: staged copy differs from its source");
  xt::g_threads.store(best_t);
  xt::g_nt.store(best_nt);
  return 0;
}

int xt_stage_get(int32_t* threads, int32_t* non_temporal) {
  if (threads) *threads = xt::g_threads.load();
  if (non_temporal) *non_temporal = xt::g_nt.load();
  return 0;
}

int xt_stage_set(int32_t threads, int32_t non_temporal) {
  XT_REQUIRE(threads >= 0 && threads <= xt::kMaxWorkers, "xt_stage_set: threads outside [0,%d]", xt::kMaxWorkers);
  xt::g_threads.store(threads);
  xt::g_nt.store(non_temporal ? 1 : 0);
  return 0;
}

}  // extern "C"
