// ThreadSanitizer tier for the input pipeline's native part: csrc/runtime/pinned_ring.cpp
// against an ASYNCHRONOUS mock of HIP (tests/native/mock_async: one worker thread per stream), driven with the
// protocol of data/prefetch.py::PinnedRingSource(refill=True):
//
//   main thread      consumer_wait(k); h2d(slot of k+1) behind the compute stream; "kernel" on the compute
//                    stream reads the device buffer of batch k; job(k-1+n) -> producer
//   producer thread  host_wait(slot) — the slot's last copy has left the host — then fill_u8 with 3 native
//                    threads; ready(k) -> main
//   copy stream      memcpy slot -> device buffer (reads the pinned slot, writes the device buffer)
//   compute stream   reads the device buffer (checksum)
//
// The only things ordering those memory accesses are the ring's events and the job / ready queues.  TSan
// fails the test on any unordered pair; the checksums prove that every batch arrived intact and in order.
#include <hip/hip_runtime.h>

#include <cassert>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------- async mock
struct MockAEvent {
  std::mutex m;
  std::condition_variable cv;
  uint64_t recorded = 0, completed = 0;
};
struct MockAStream {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  uint64_t submitted = 0, done = 0;
  bool stop = false;
  std::thread worker;
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        f = std::move(q.front());
        q.pop_front();
      }
      f();
      {
        std::lock_guard<std::mutex> l(m);
        ++done;
      }
      cv.notify_all();
    }
  }
  void push(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> l(m);
      q.push_back(std::move(f));
      ++submitted;
    }
    cv.notify_all();
  }
};
static int g_streams = 0, g_events = 0, g_host_allocs = 0;

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = new MockAStream();
  (*s)->worker = std::thread([st = *s] { st->run(); });
  ++g_streams;
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
  std::unique_lock<std::mutex> l(s->m);
  const uint64_t target = s->submitted;
  s->cv.wait(l, [&] { return s->done >= target; });
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  hipStreamSynchronize(s);
  {
    std::lock_guard<std::mutex> l(s->m);
    s->stop = true;
  }
  s->cv.notify_all();
  s->worker.join();
  delete s;
  --g_streams;
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = new MockAEvent();
  ++g_events;
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  --g_events;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> l(e->m);
    ticket = ++e->recorded;
  }
  s->push([e, ticket] {
    {
      std::lock_guard<std::mutex> l(e->m);
      if (e->completed < ticket) e->completed = ticket;
    }
    e->cv.notify_all();
  });
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> l(e->m);
    ticket = e->recorded;          // the record the wait refers to is the latest one at CALL time (HIP semantics)
  }
  s->push([e, ticket] {
    std::unique_lock<std::mutex> l(e->m);
    e->cv.wait(l, [&] { return e->completed >= ticket; });
  });
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
  std::lock_guard<std::mutex> l(e->m);
  return e->completed >= e->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> l(e->m);
  const uint64_t ticket = e->recorded;
  e->cv.wait(l, [&] { return e->completed >= ticket; });
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  *p = std::malloc(bytes);
  ++g_host_allocs;
  return *p ? hipSuccess : 2;
}
hipError_t hipHostFree(void* p) {
  std::free(p);
  --g_host_allocs;
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
  s->push([=] { std::memcpy(dst, src, bytes); });
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------- the ring's C API
extern "C" {
void* dtp_ring_create(int n_slots, size_t slot_bytes, int device);
void* dtp_ring_slot(void* h, int i);
int dtp_ring_h2d(void* h, int i, size_t src_off, void* dev_dst, size_t bytes, hipStream_t consumer_stream, int last_piece);
int dtp_ring_consumer_wait(void* h, int i, hipStream_t consumer_stream);
int dtp_ring_slot_busy(void* h, int i);
int dtp_ring_host_wait(void* h, int i);
int dtp_ring_fill_u8(void* h, int i, int count, size_t image_bytes, size_t label_off, int num_classes,
                     unsigned long long seed, int threads);
int dtp_ring_stats(void* h, unsigned long long* copies, unsigned long long* bytes);
int dtp_ring_destroy(void* h);
}

static uint64_t checksum(const uint8_t* p, size_t n) {
  uint64_t s = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) s = (s ^ p[i]) * 1099511628211ull;
  return s;
}

int main() {
  constexpr int N_SLOTS = 3, COUNT = 6, STEPS = 40, CLASSES = 3;
  constexpr size_t IMAGE = 3 * 24 * 24, LABEL_OFF = 10496 /* >= COUNT * IMAGE, 256-aligned */, SLOT = LABEL_OFF + 256;
  static_assert(COUNT * IMAGE <= LABEL_OFF && LABEL_OFF % 256 == 0, "slot layout");
  void* ring = dtp_ring_create(N_SLOTS, SLOT, 0);
  assert(ring);
  hipStream_t compute;
  hipStreamCreateWithFlags(&compute, hipStreamNonBlocking);

  // reference checksums: the same generator into a private ring, single-threaded
  std::vector<uint64_t> want(STEPS);
  {
    void* ref = dtp_ring_create(1, SLOT, 0);
    for (int k = 0; k < STEPS; ++k) {
      assert(dtp_ring_fill_u8(ref, 0, COUNT, IMAGE, LABEL_OFF, CLASSES, 1000003ull + k, 1) == 0);
      want[k] = checksum((const uint8_t*)dtp_ring_slot(ref, 0), LABEL_OFF + COUNT * 8);
    }
    dtp_ring_destroy(ref);
  }

  auto produce = [&](int k) {
    assert(dtp_ring_fill_u8(ring, k % N_SLOTS, COUNT, IMAGE, LABEL_OFF, CLASSES, 1000003ull + k, 3) == 0);
  };
  for (int k = 0; k < N_SLOTS; ++k) produce(k);

  std::mutex jm;
  std::condition_variable jcv;
  std::deque<int> jobs;
  int ready_upto = N_SLOTS - 1;
  std::thread producer([&] {
    for (;;) {
      int k;
      {
        std::unique_lock<std::mutex> l(jm);
        jcv.wait(l, [&] { return !jobs.empty(); });
        k = jobs.front();
        jobs.pop_front();
      }
      if (k < 0) return;
#ifndef RING_TEST_SKIP_HOST_WAIT   // (defined by the test's negative control: TSan must then report the race)
      assert(dtp_ring_host_wait(ring, k % N_SLOTS) == 0);     // the slot's previous batch has left the host
#endif
      produce(k);
      {
        std::lock_guard<std::mutex> l(jm);
        ready_upto = k;
      }
      jcv.notify_all();
    }
  });

  std::vector<uint8_t> dev[2] = {std::vector<uint8_t>(SLOT), std::vector<uint8_t>(SLOT)};
  std::vector<uint64_t> got(STEPS, 0);
  auto queue_copy = [&](int k) {
    {
      std::unique_lock<std::mutex> l(jm);
      jcv.wait(l, [&] { return k <= ready_upto; });
    }
    uint8_t* d = dev[k % 2].data();
    assert(dtp_ring_h2d(ring, k % N_SLOTS, 0, d, COUNT * IMAGE, compute, 0) == 0);
    assert(dtp_ring_h2d(ring, k % N_SLOTS, LABEL_OFF, d + LABEL_OFF, COUNT * 8, compute, 1) == 0);
  };
  assert(dtp_ring_consumer_wait(ring, 0, compute) == -3);      // nothing copied yet: refused, not a hang
  queue_copy(0);
  for (int k = 0; k < STEPS; ++k) {
    assert(dtp_ring_consumer_wait(ring, k % N_SLOTS, compute) == 0);
    if (k + 1 < STEPS) queue_copy(k + 1);
    // the step's "kernel": reads batch k's device buffer on the compute stream
    const uint8_t* d = dev[k % 2].data();
    uint64_t* out = &got[k];
    compute->push([=] {
      std::vector<uint8_t> tmp(LABEL_OFF + COUNT * 8);
      std::memcpy(tmp.data(), d, COUNT * IMAGE);
      std::memset(tmp.data() + COUNT * IMAGE, 0, LABEL_OFF - COUNT * IMAGE);
      std::memcpy(tmp.data() + LABEL_OFF, d + LABEL_OFF, COUNT * 8);
      *out = checksum(tmp.data(), tmp.size());
    });
    if (k >= 1 && k - 1 + N_SLOTS < STEPS) {
      {
        std::lock_guard<std::mutex> l(jm);
        jobs.push_back(k - 1 + N_SLOTS);                       // slot of batch k-1: regenerate once its copy is done
      }
      jcv.notify_all();
    }
  }
  hipStreamSynchronize(compute);
  {
    std::lock_guard<std::mutex> l(jm);
    jobs.push_back(-1);
  }
  jcv.notify_all();
  producer.join();

  // the reference slots have zero padding between images and labels only if the allocator zeroed it: compare the
  // payload (images + labels), which is what the consumer reads
  {
    void* ref = dtp_ring_create(1, SLOT, 0);
    for (int k = 0; k < STEPS; ++k) {
      assert(dtp_ring_fill_u8(ref, 0, COUNT, IMAGE, LABEL_OFF, CLASSES, 1000003ull + k, 1) == 0);
      std::vector<uint8_t> tmp(LABEL_OFF + COUNT * 8);
      const uint8_t* s = (const uint8_t*)dtp_ring_slot(ref, 0);
      std::memcpy(tmp.data(), s, COUNT * IMAGE);
      std::memset(tmp.data() + COUNT * IMAGE, 0, LABEL_OFF - COUNT * IMAGE);
      std::memcpy(tmp.data() + LABEL_OFF, s + LABEL_OFF, COUNT * 8);
      want[k] = checksum(tmp.data(), tmp.size());
    }
    dtp_ring_destroy(ref);
  }
  for (int k = 0; k < STEPS; ++k) {
    if (got[k] != want[k]) {
      std::fprintf(stderr, "batch %d arrived damaged or out of order\n", k);
      return 1;
    }
  }
  unsigned long long copies = 0, bytes = 0;
  assert(dtp_ring_stats(ring, &copies, &bytes) == 0 && copies == 2ull * STEPS);
  assert(dtp_ring_slot_busy(ring, 0) == 0);
  dtp_ring_destroy(ring);
  hipStreamDestroy(compute);
  if (g_streams != 0 || g_events != 0 || g_host_allocs != 0) {
    std::fprintf(stderr, "leak: %d streams, %d events, %d pinned buffers\n", g_streams, g_events, g_host_allocs);
    return 1;
  }
  std::printf("ring tsan test ok: %d batches through %d slots\n", STEPS, N_SLOTS);
  return 0;
}
