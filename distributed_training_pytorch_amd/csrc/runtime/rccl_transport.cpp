// Native gradient-bucket transport: a dedicated RCCL communicator + HIP communication stream per
// process, event-synchronised with the compute stream.
//
//   producer (backward kernels, compute stream) --event--> comm stream: ncclAllReduce(bucket)
//   comm stream --event--> consumer (optimizer, compute stream)
//
// The bucket all-reduces run on their own stream, so they overlap with the remaining backward
// kernels; nothing here blocks the host.  Averaging is folded into the collective (ncclAvg).
// One process per GPU; the unique id is created on rank 0 and distributed by the caller (through
// the torch.distributed store), so this file has no dependency on PyTorch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <mutex>
#include <vector>

namespace {

struct Transport {
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  int rank = 0, world = 1, device = 0;
  // Reusable events, oldest first.  An event is re-recorded only after hipEventQuery says the work
  // it marks has completed; while the oldest is still pending (a pass with more than kRing / 3
  // buckets in sharded mode, a slow collective) the pool GROWS instead — re-recording a live event
  // would silently move the point a waiting stream is released at.  Events recorded while a stream
  // is being captured into a hipGraph belong to that graph for good and are never reused.
  static constexpr size_t kRing = 64;
  std::deque<hipEvent_t> events;
  std::vector<hipEvent_t> graph_events;
  unsigned long long launched = 0, bytes = 0;
  std::mutex mu;

  static bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return s != nullptr && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
  }

  hipEvent_t event(hipStream_t recorded_on) {
    const bool cap = capturing(recorded_on) || capturing(comm_stream);
    if (!cap && events.size() >= kRing && hipEventQuery(events.front()) == hipSuccess) {
      hipEvent_t e = events.front();
      events.pop_front();
      events.push_back(e);
      return e;
    }
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (cap) graph_events.push_back(e);
    else events.push_back(e);
    return e;
  }
};

inline int fail(const char* what, int code) {
  std::fprintf(stderr, "[dtp_runtime] %s failed (%d)\n", what, code);
  return code ? code : -1;
}

#define DTP_HIP(x)                                   \
  do {                                               \
    hipError_t e_ = (x);                             \
    if (e_ != hipSuccess) return fail(#x, (int)e_);  \
  } while (0)
#define DTP_NCCL(x)                                      \
  do {                                                   \
    ncclResult_t r_ = (x);                               \
    if (r_ != ncclSuccess) return fail(#x, 1000 + (int)r_); \
  } while (0)

ncclDataType_t dtype_of(int kind) { return kind == 1 ? ncclBfloat16 : ncclFloat32; }

// roctx ranges around every bucket launch ("allreduce-bucket-k" in
// rocprofv3 --marker-trace).  Opt-in with DTP_ROCTX=1; the library is looked up at run time so that
// this file links against nothing but HIP and RCCL.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* on = std::getenv("DTP_ROCTX");
    if (!on || on[0] != '1') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
  bool on;
  RoctxRange(const char* what, int index, size_t bytes) : on(roctx().push != nullptr) {
    if (!on) return;
    char buf[96];
    std::snprintf(buf, sizeof(buf), "dtp:%s bucket %d (%zu B)", what, index, bytes);
    roctx().push(buf);
  }
  ~RoctxRange() { if (on) roctx().pop(); }
};

}  // namespace

extern "C" {

int dtp_rccl_unique_id(void* out, int out_bytes) {
  if (out_bytes < (int)sizeof(ncclUniqueId)) return -2;
  ncclUniqueId id;
  DTP_NCCL(ncclGetUniqueId(&id));
  std::memcpy(out, &id, sizeof(id));
  return (int)sizeof(id);
}

int dtp_rccl_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

void* dtp_rccl_create(const void* id_bytes, int rank, int world, int device) {
  if (world < 1 || rank < 0 || rank >= world) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  auto* t = new Transport();
  t->rank = rank; t->world = world; t->device = device;
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  if (ncclCommInitRank(&t->comm, world, id, rank) != ncclSuccess) { delete t; return nullptr; }
  // Default-priority, non-blocking stream (what ProcessGroupNCCL uses too) — by construction, not
  // by option.  Round 1 created this stream with HIGH priority and the device hung (test at world
  // size 1 after a warm-up step; hipDeviceSynchronize never returned).  What is known: the compute
  // kernels of this framework (8-phase implicit GEMM, wgrad2) run ONE workgroup per CU that owns
  // the CU's whole register file (2 waves/SIMD x 256 VGPRs) and 128 of its 160 KB of LDS, so an
  // RCCL kernel can only become resident on a CU between two tiles; at equal priority the
  // dispatcher hands it CUs at those tile boundaries (every ~60 us).  A high-priority queue asks the
  // hardware for more than that — precedence in queue arbitration while its event-gated barrier
  // packet is pending, and preemption of resident lower-priority workgroups — and with full-CU
  // workgroups holding LDS-DMA in flight that did not terminate on gfx950 / ROCm 7.0.  The exact
  // mechanism was not bisected (a hung device is a lost box for everyone on the host); the design
  // no longer depends on it: every stream has default priority, no kernel of this framework waits
  // inside a launch for another queue (cross-stream ordering is events only), and the overlap is
  // measured instead of forced (allreduce_exposed_ms_per_step).  See docs/KERNELS.md.
  if (hipStreamCreateWithFlags(&t->comm_stream, hipStreamNonBlocking) != hipSuccess) {
    ncclCommDestroy(t->comm);
    delete t;
    return nullptr;
  }
  return t;
}

// In-place all-reduce of buf[count] once everything already queued on producer_stream is done.
int dtp_rccl_allreduce_async(void* h, void* buf, size_t count, int dtype_kind, int average,
                             hipStream_t producer_stream) {
  auto* t = static_cast<Transport*>(h);
  if (!t || !buf || count == 0) return -2;
  std::lock_guard<std::mutex> lk(t->mu);
  hipEvent_t ready = t->event(producer_stream);
  if (!ready) return -3;
  DTP_HIP(hipEventRecord(ready, producer_stream));
  DTP_HIP(hipStreamWaitEvent(t->comm_stream, ready, 0));
  DTP_NCCL(ncclAllReduce(buf, buf, count, dtype_of(dtype_kind), average ? ncclAvg : ncclSum, t->comm,
                         t->comm_stream));
  t->launched += 1;
  t->bytes += count * (dtype_kind == 1 ? 2 : 4);
  return 0;
}

// Sharded mode, first half: in-place reduce-scatter (average) of buf[count]; rank r's slice
// buf[r*S, (r+1)*S), S = count / world, receives the result.  count % world must be 0.
int dtp_rccl_reduce_scatter_async(void* h, void* buf, size_t count, int dtype_kind, hipStream_t producer_stream) {
  auto* t = static_cast<Transport*>(h);
  if (!t || !buf || count == 0 || count % (size_t)t->world != 0) return -2;
  std::lock_guard<std::mutex> lk(t->mu);
  const size_t S = count / (size_t)t->world, esz = dtype_kind == 1 ? 2 : 4;
  if (producer_stream != t->comm_stream) {
    hipEvent_t ready = t->event(producer_stream);
    if (!ready) return -3;
    DTP_HIP(hipEventRecord(ready, producer_stream));
    DTP_HIP(hipStreamWaitEvent(t->comm_stream, ready, 0));
  }
  DTP_NCCL(ncclReduceScatter(buf, (char*)buf + (size_t)t->rank * S * esz, S, dtype_of(dtype_kind), ncclAvg, t->comm,
                             t->comm_stream));
  t->launched += 1;
  t->bytes += count * esz;
  return 0;
}

// Sharded mode, second half: in-place all-gather; every rank contributes its slice of buf[count].
// ``producer_stream`` may be the null handle meaning "already ordered on the comm stream".
int dtp_rccl_all_gather_async(void* h, void* buf, size_t count, int dtype_kind, hipStream_t producer_stream,
                              int on_comm_stream) {
  auto* t = static_cast<Transport*>(h);
  if (!t || !buf || count == 0 || count % (size_t)t->world != 0) return -2;
  std::lock_guard<std::mutex> lk(t->mu);
  const size_t S = count / (size_t)t->world, esz = dtype_kind == 1 ? 2 : 4;
  if (!on_comm_stream) {
    hipEvent_t ready = t->event(producer_stream);
    if (!ready) return -3;
    DTP_HIP(hipEventRecord(ready, producer_stream));
    DTP_HIP(hipStreamWaitEvent(t->comm_stream, ready, 0));
  }
  DTP_NCCL(ncclAllGather((char*)buf + (size_t)t->rank * S * esz, buf, S, dtype_of(dtype_kind), t->comm,
                         t->comm_stream));
  t->launched += 1;
  t->bytes += count * esz;
  return 0;
}

void* dtp_rccl_comm_stream(void* h) {
  auto* t = static_cast<Transport*>(h);
  return t ? (void*)t->comm_stream : nullptr;
}

// Make consumer_stream wait for every collective launched so far (no host blocking).
int dtp_rccl_wait(void* h, hipStream_t consumer_stream) {
  auto* t = static_cast<Transport*>(h);
  if (!t) return -2;
  std::lock_guard<std::mutex> lk(t->mu);
  hipEvent_t done = t->event(consumer_stream);
  if (!done) return -3;
  DTP_HIP(hipEventRecord(done, t->comm_stream));
  DTP_HIP(hipStreamWaitEvent(consumer_stream, done, 0));
  return 0;
}

int dtp_rccl_broadcast(void* h, void* buf, size_t count, int dtype_kind, int root, hipStream_t stream) {
  auto* t = static_cast<Transport*>(h);
  if (!t || !buf) return -2;
  std::lock_guard<std::mutex> lk(t->mu);
  hipEvent_t ready = t->event(stream);
  if (!ready) return -3;
  DTP_HIP(hipEventRecord(ready, stream));
  DTP_HIP(hipStreamWaitEvent(t->comm_stream, ready, 0));
  DTP_NCCL(ncclBroadcast(buf, buf, count, dtype_of(dtype_kind), root, t->comm, t->comm_stream));
  hipEvent_t done = t->event(stream);
  if (!done) return -3;
  DTP_HIP(hipEventRecord(done, t->comm_stream));
  DTP_HIP(hipStreamWaitEvent(stream, done, 0));
  return 0;
}

int dtp_rccl_sync(void* h) {
  auto* t = static_cast<Transport*>(h);
  if (!t) return -2;
  DTP_HIP(hipStreamSynchronize(t->comm_stream));
  return 0;
}

int dtp_rccl_stats(void* h, unsigned long long* launched, unsigned long long* bytes) {
  auto* t = static_cast<Transport*>(h);
  if (!t) return -2;
  *launched = t->launched;
  *bytes = t->bytes;
  return 0;
}

// events currently owned by the transport (reusable pool + hipGraph-owned): for tests and leak checks
int dtp_rccl_event_count(void* h) {
  auto* t = static_cast<Transport*>(h);
  if (!t) return -2;
  std::lock_guard<std::mutex> lk(t->mu);
  return (int)(t->events.size() + t->graph_events.size());
}

int dtp_rccl_destroy(void* h) {
  auto* t = static_cast<Transport*>(h);
  if (!t) return 0;
  hipStreamSynchronize(t->comm_stream);
  for (hipEvent_t e : t->events) hipEventDestroy(e);
  for (hipEvent_t e : t->graph_events) hipEventDestroy(e);
  ncclCommDestroy(t->comm);
  hipStreamDestroy(t->comm_stream);
  delete t;
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Bucket scheduler: the hot-path half of the gradient reducer (parallel/reducer.py keeps planning,
// hooks and the gloo path).  Buckets are contiguous ranges of the flat gradient arena; a bucket's
// all-reduce is launched the moment its last parameter reports ready — one call from the backward
// kernel wrapper, no Python per-bucket work — and finalize() hands the comm stream back to the
// consumer with a single event.  A parameter counts once per pass (fused layers notify explicitly
// AND autograd may fire its hook for the same parameter; counting both launched buckets early).
struct DtpBucket {
  void* ptr;
  size_t count;
  int dtype_kind;
  int expected, pending;
  bool launched;
};

// Fused flat-arena optimizer kernels of libdtp_hip.so (csrc/optim.hip), handed over as function
// pointers so that this library does not link against the kernel library.
typedef int (*dtp_sgd_fn)(float* p, const void* g, int grad_bf16, float* mom, void* lp, size_t n, float lr,
                          float momentum, float dampening, float wd, float grad_scale, int nesterov, int first_step,
                          hipStream_t s);
typedef int (*dtp_adamw_fn)(float* p, const void* g, int grad_bf16, float* m, float* v, void* lp, size_t n, float lr,
                            float beta1, float beta2, float eps, float wd, float grad_scale, int step, hipStream_t s);

// Sharded mode: what this rank owns of one bucket (all pointers at the start of the OWNED slice),
// plus the whole bucket's range of the working-weight buffer for the all-gather.
struct DtpShard {
  float* master; void* grad; float* s1; float* s2; void* lp;
  size_t n;           // elements in the owned slice
  void* wbase;        // bucket range of the working weights (all-gather buffer)
  size_t wcount;      // = n * world
};

struct DtpReducer {
  void* transport;
  std::vector<DtpBucket> buckets;
  std::vector<std::vector<int>> param_buckets;
  std::vector<char> ready;
  bool require_sync = true;
  unsigned long long passes = 0, launched = 0, duplicate_notifications = 0;
  // sharded optimizer
  int opt_kind = 0;                 // 0: all-reduce mode; 1: SGD; 2: AdamW
  int shard_world = 0;              // optimizer-in-scheduler modes: 1 = replicated (all-reduce, then the whole
                                    // bucket is updated on every rank), > 1 = sharded over that many ranks
  dtp_sgd_fn sgd = nullptr;
  dtp_adamw_fn adamw = nullptr;
  std::vector<DtpShard> shards;
  int wkind = 1;                    // dtype of the working weights (1 = bf16)
  float hyper[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int iargs[4] = {0, 0, 0, 0};
  bool pass_armed = false;
};

// one bucket: all-reduce, or reduce-scatter -> optimizer on the owned slice -> all-gather
static int launch_bucket(DtpReducer* r, int bi, hipStream_t producer_stream) {
  DtpBucket& b = r->buckets[bi];
  RoctxRange range(r->opt_kind == 0 ? "allreduce" : "reduce-scatter+opt+all-gather", bi,
                   b.count * (b.dtype_kind == 1 ? 2 : 4));
  if (r->opt_kind == 0)
    return dtp_rccl_allreduce_async(r->transport, b.ptr, b.count, b.dtype_kind, 1, producer_stream);
  if (!r->pass_armed) return -30;      // begin_pass() must precede the first launch of a pass
  // replicated mode (the default data-parallel path): all-reduce, then the fused optimizer updates the
  // WHOLE bucket on the communication stream — the 0.5 ms optimizer pass over 134 M parameters hides
  // under the rest of backward bucket by bucket instead of running after the last collective
  int rc = r->shard_world == 1
               ? dtp_rccl_allreduce_async(r->transport, b.ptr, b.count, b.dtype_kind, 1, producer_stream)
               : dtp_rccl_reduce_scatter_async(r->transport, b.ptr, b.count, b.dtype_kind, producer_stream);
  if (rc != 0) return rc;
  const DtpShard& sh = r->shards[bi];
  hipStream_t cs = (hipStream_t)dtp_rccl_comm_stream(r->transport);
  const float* h = r->hyper;
  if (r->opt_kind == 1)
    rc = r->sgd(sh.master, sh.grad, b.dtype_kind == 1, sh.s1, sh.lp, sh.n, h[0], h[1], h[2], h[3], 1.0f, r->iargs[0],
                r->iargs[1], cs);
  else
    rc = r->adamw(sh.master, sh.grad, b.dtype_kind == 1, sh.s1, sh.s2, sh.lp, sh.n, h[0], h[1], h[2], h[3], h[4], 1.0f,
                  r->iargs[0], cs);
  if (rc != 0) return rc < 0 ? rc : -rc;
  if (r->shard_world == 1) return 0;   // every rank updated every element: nothing to gather
  return dtp_rccl_all_gather_async(r->transport, sh.wbase, sh.wcount, r->wkind, cs, 1);
}

void* dtp_reducer_create(void* transport, int n_buckets, void* const* ptrs, const size_t* counts,
                         const int* dtype_kinds, int n_params, const int* pb_offsets, const int* pb_indices) {
  if (!transport || n_buckets <= 0 || n_params <= 0 || !ptrs || !counts || !dtype_kinds || !pb_offsets || !pb_indices)
    return nullptr;
  auto* r = new DtpReducer();
  r->transport = transport;
  r->buckets.resize(n_buckets);
  for (int b = 0; b < n_buckets; ++b) {
    if (!ptrs[b] || counts[b] == 0) { delete r; return nullptr; }
    r->buckets[b] = DtpBucket{ptrs[b], counts[b], dtype_kinds[b], 0, 0, false};
  }
  r->param_buckets.resize(n_params);
  r->ready.assign(n_params, 0);
  for (int p = 0; p < n_params; ++p) {
    for (int k = pb_offsets[p]; k < pb_offsets[p + 1]; ++k) {
      const int b = pb_indices[k];
      if (b < 0 || b >= n_buckets) { delete r; return nullptr; }
      bool dup = false;
      for (int q : r->param_buckets[p]) dup = dup || q == b;
      if (dup) continue;
      r->param_buckets[p].push_back(b);
      r->buckets[b].expected += 1;
    }
  }
  for (auto& b : r->buckets) {
    if (b.expected == 0) { delete r; return nullptr; }   // a bucket nobody feeds would never launch
    b.pending = b.expected;
  }
  return r;
}

// returns the number of buckets launched by this notification (>= 0) or a negative error
int dtp_reducer_mark_ready(void* h, int param, hipStream_t producer_stream) {
  auto* r = static_cast<DtpReducer*>(h);
  if (!r || param < 0 || param >= (int)r->ready.size()) return -2;
  if (r->ready[param]) { r->duplicate_notifications += 1; return 0; }
  r->ready[param] = 1;
  int n = 0;
  for (int bi : r->param_buckets[param]) {
    DtpBucket& b = r->buckets[bi];
    b.pending -= 1;
    if (b.pending == 0 && r->require_sync && !b.launched) {
      const int rc = launch_bucket(r, bi, producer_stream);
      if (rc != 0) return rc < 0 ? rc : -rc;
      b.launched = true;
      r->launched += 1;
      ++n;
    }
  }
  return n;
}

// end of a backward pass: launch what never became ready (unused parameters), make the consumer
// stream wait for every collective, reset for the next pass.  Returns buckets launched here.
int dtp_reducer_finalize(void* h, hipStream_t stream) {
  auto* r = static_cast<DtpReducer*>(h);
  if (!r) return -2;
  int n = 0;
  if (r->require_sync) {
    for (int bi = 0; bi < (int)r->buckets.size(); ++bi) {
      DtpBucket& b = r->buckets[bi];
      if (b.launched) continue;
      const int rc = launch_bucket(r, bi, stream);
      if (rc != 0) return rc < 0 ? rc : -rc;
      b.launched = true;
      r->launched += 1;
      ++n;
    }
    const int rc = dtp_rccl_wait(r->transport, stream);
    if (rc != 0) return rc < 0 ? rc : -rc;
    r->passes += 1;
    r->pass_armed = false;
  }
  for (auto& b : r->buckets) { b.pending = b.expected; b.launched = false; }
  std::fill(r->ready.begin(), r->ready.end(), 0);
  return n;
}

// Switch the scheduler to an optimizer-in-scheduler mode.  Arrays have one entry per bucket; the
// slice pointers address THIS rank's 1/world of the bucket.  opt_kind 1 = SGD (s2 unused), 2 = AdamW.
// world == 1: replicated — all-reduce + update of the whole bucket on every rank, no all-gather
// (whatever the communicator's size); world > 1: reduce-scatter + owned slice + all-gather.
int dtp_reducer_set_shard(void* h, int opt_kind, void* sgd_fn, void* adamw_fn, int n_buckets, void* const* master,
                          void* const* grad, void* const* s1, void* const* s2, void* const* lp,
                          const size_t* slice_counts, void* const* wbase, int wkind, int world) {
  auto* r = static_cast<DtpReducer*>(h);
  if (!r || n_buckets != (int)r->buckets.size() || world < 1) return -2;
  if ((opt_kind == 1 && !sgd_fn) || (opt_kind == 2 && !adamw_fn) || opt_kind < 1 || opt_kind > 2) return -3;
  r->shards.resize(n_buckets);
  for (int b = 0; b < n_buckets; ++b) {
    if (!master[b] || !grad[b] || !lp[b] || !wbase[b] || slice_counts[b] == 0 || (slice_counts[b] & 3)) return -4;
    if (slice_counts[b] * (size_t)world != r->buckets[b].count) return -5;
    r->shards[b] = DtpShard{(float*)master[b], grad[b], (float*)s1[b], (float*)(s2 ? s2[b] : nullptr), lp[b],
                            slice_counts[b], wbase[b], slice_counts[b] * (size_t)world};
  }
  r->sgd = (dtp_sgd_fn)sgd_fn;
  r->adamw = (dtp_adamw_fn)adamw_fn;
  r->wkind = wkind;
  r->opt_kind = opt_kind;
  r->shard_world = world;
  return 0;
}

// Per backward pass, before the first bucket can become ready: the optimizer's scalar arguments.
// SGD: hyper = {lr, momentum, dampening, weight_decay}, iargs = {nesterov, first_step}.
// AdamW: hyper = {lr, beta1, beta2, eps, weight_decay}, iargs = {step}.
int dtp_reducer_begin_pass(void* h, const float* hyper, int n_hyper, const int* iargs, int n_iargs) {
  auto* r = static_cast<DtpReducer*>(h);
  if (!r || n_hyper > 8 || n_iargs > 4) return -2;
  for (int i = 0; i < n_hyper; ++i) r->hyper[i] = hyper[i];
  for (int i = 0; i < n_iargs; ++i) r->iargs[i] = iargs[i];
  r->pass_armed = true;
  return 0;
}

int dtp_reducer_set_sync(void* h, int on) {
  auto* r = static_cast<DtpReducer*>(h);
  if (!r) return -2;
  r->require_sync = on != 0;
  return 0;
}

int dtp_reducer_stats(void* h, unsigned long long* passes, unsigned long long* launched,
                      unsigned long long* duplicates) {
  auto* r = static_cast<DtpReducer*>(h);
  if (!r) return -2;
  *passes = r->passes; *launched = r->launched; *duplicates = r->duplicate_notifications;
  return 0;
}

int dtp_reducer_destroy(void* h) {
  delete static_cast<DtpReducer*>(h);
  return 0;
}

}  // extern "C"
