// Host-logic test of the native RCCL transport against mock HIP/RCCL (built with
// -fsanitize=address,undefined by tests/test_native_host.py).  Checks, for a long run of
// launches: every collective is preceded by "record(producer) -> wait(comm stream)", waits hand
// the comm stream back to the consumer, the event ring is bounded and re-used, statistics add up,
// argument validation rejects bad calls, and destroy releases every event / stream / communicator.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <string>

#include "mock/hip/hip_runtime.h"
#include "mock/rccl/rccl.h"

// ---- mock implementation -----------------------------------------------------------------------
static MockLog g_log;
MockLog& mock_log() { return g_log; }
static void logc(const std::string& s) { g_log.calls.push_back(s); }
static std::vector<MockEvent*> g_live_events;
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) {   // default priority: by construction
  *s = new MockStream{100 + g_log.streams_created++, (flags & hipStreamNonBlocking) != 0, 0, false};
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) { ++g_log.streams_destroyed; delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { logc("sync:" + std::to_string(s ? s->id : 0)); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  assert(e && !e->destroyed && e->records > 0 && "waiting on an event that was never recorded");
  logc("wait:" + std::to_string(s ? s->id : 0) + ":on:" + std::to_string(e->last_stream ? e->last_stream->id : 0));
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  if (g_log.fail_event_create_after >= 0 && g_log.events_created >= g_log.fail_event_create_after) return 2;
  *e = new MockEvent{g_log.events_created++, 0, nullptr, false};
  g_live_events.push_back(*e);
  return hipSuccess;
}
static void mock_complete_all_events() { for (MockEvent* e : g_live_events) e->pending = false; }
hipError_t hipEventQuery(hipEvent_t e) { assert(e && !e->destroyed); return e->pending ? hipErrorNotReady : hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st) {
  *st = (s && s->capturing) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  assert(e && !e->destroyed);
  assert(!e->pending && "re-recording an event whose previous work has not completed");
  e->pending = g_log.keep_events_pending;
  ++e->records;
  e->last_stream = s;
  logc("record:" + std::to_string(s ? s->id : 0));
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  ++g_log.events_destroyed;
  for (auto& p : g_live_events) if (p == e) { p = g_live_events.back(); g_live_events.pop_back(); break; }
  e->destroyed = true;
  delete e;
  return hipSuccess;
}
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 7, sizeof(*id)); return ncclSuccess; }
static int g_comms = 0;
ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId id, int rank) {
  assert(id.internal[5] == 7);
  *c = new MockComm{rank, world, false};
  ++g_comms;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { --g_comms; delete c; return ncclSuccess; }
ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t st) {
  assert(s == r && c && st && st->non_blocking);
  logc("allreduce:" + std::to_string(st->id) + ":" + std::to_string(n) + ":" + std::to_string((int)dt) + ":" +
       std::to_string((int)op));
  return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void*, void*, size_t n, ncclDataType_t, int root, ncclComm_t, hipStream_t st) {
  logc("broadcast:" + std::to_string(st->id) + ":" + std::to_string(n) + ":" + std::to_string(root));
  return ncclSuccess;
}

ncclResult_t ncclReduceScatter(const void* s, void* r, size_t n, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                               hipStream_t st) {
  // in-place contract: recv == send + rank * recvcount * elem
  const size_t esz = dt == ncclBfloat16 ? 2 : 4;
  assert(c && st && st->non_blocking && (const char*)r == (const char*)s + (size_t)c->rank * n * esz);
  logc("reducescatter:" + std::to_string(st->id) + ":" + std::to_string(n) + ":" + std::to_string((int)op));
  return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* s, void* r, size_t n, ncclDataType_t dt, ncclComm_t c, hipStream_t st) {
  const size_t esz = dt == ncclBfloat16 ? 2 : 4;
  assert(c && st && st->non_blocking && (const char*)s == (const char*)r + (size_t)c->rank * n * esz);
  logc("allgather:" + std::to_string(st->id) + ":" + std::to_string(n));
  return ncclSuccess;
}

// ---- unit under test -----------------------------------------------------------------------------
extern "C" {
int dtp_rccl_unique_id(void* out, int out_bytes);
int dtp_rccl_unique_id_bytes();
void* dtp_rccl_create(const void* id_bytes, int rank, int world, int device);
int dtp_rccl_allreduce_async(void* h, void* buf, size_t count, int dtype_kind, int average, hipStream_t producer);
int dtp_rccl_wait(void* h, hipStream_t consumer);
int dtp_rccl_broadcast(void* h, void* buf, size_t count, int dtype_kind, int root, hipStream_t stream);
int dtp_rccl_sync(void* h);
int dtp_rccl_stats(void* h, unsigned long long* launched, unsigned long long* bytes);
int dtp_rccl_destroy(void* h);
int dtp_rccl_event_count(void* h);
void* dtp_reducer_create(void* transport, int n_buckets, void* const* ptrs, const size_t* counts,
                         const int* dtype_kinds, int n_params, const int* pb_offsets, const int* pb_indices);
int dtp_reducer_mark_ready(void* h, int param, hipStream_t producer_stream);
int dtp_reducer_finalize(void* h, hipStream_t stream);
int dtp_reducer_set_sync(void* h, int on);
int dtp_reducer_stats(void* h, unsigned long long* passes, unsigned long long* launched,
                      unsigned long long* duplicates);
int dtp_reducer_destroy(void* h);
int dtp_reducer_set_shard(void* h, int opt_kind, void* sgd_fn, void* adamw_fn, int n_buckets, void* const* master,
                          void* const* grad, void* const* s1, void* const* s2, void* const* lp,
                          const size_t* slice_counts, void* const* wbase, int wkind, int world);
int dtp_reducer_begin_pass(void* h, const float* hyper, int n_hyper, const int* iargs, int n_iargs);
}

static int count_calls(const char* prefix) {
  int n = 0;
  for (const auto& c : g_log.calls) n += c.rfind(prefix, 0) == 0;
  return n;
}

// stand-in for csrc/optim.hip's dtp_sgd_flat: records what the scheduler asked for
static int fake_sgd(float* p, const void* g, int g_bf16, float* mom, void* lp, size_t n, float lr, float momentum,
                    float dampening, float wd, float grad_scale, int nesterov, int first_step, hipStream_t s) {
  assert(p && g && mom && lp && s && s->non_blocking && grad_scale == 1.0f && dampening == 0.0f && wd == 1e-4f);
  logc("sgd:" + std::to_string(s->id) + ":" + std::to_string(n) + ":" + std::to_string(g_bf16) + ":" +
       std::to_string(lr == 0.5f && momentum == 0.9f) + ":" + std::to_string(nesterov) + std::to_string(first_step));
  return 0;
}

// sharded mode: per bucket reduce-scatter -> optimizer on the owned slice -> all-gather, all on the
// comm stream, one producer event per bucket, one consumer wait per pass (rank 1 of world 4)
static void test_sharded_scheduler(void* transport, MockStream* compute) {
  static float master[256], mom[256];
  static unsigned short grad[256], lp[256];
  const int rank = 1, world = 4;
  // two buckets: elements [128, 256) and [0, 128) of the arena (reverse registration order)
  void* ptrs[2] = {grad + 128, grad};
  size_t counts[2] = {128, 128};
  int kinds[2] = {1, 1};
  int offs[3] = {0, 1, 2};
  int idx[2] = {1, 0};                                      // param 0 -> bucket 1, param 1 -> bucket 0
  void* r = dtp_reducer_create(transport, 2, ptrs, counts, kinds, 2, offs, idx);
  assert(r);
  const size_t S = 128 / world;
  void* m_[2] = {master + 128 + rank * S, master + rank * S};
  void* g_[2] = {grad + 128 + rank * S, grad + rank * S};
  void* s1_[2] = {mom + 128 + rank * S, mom + rank * S};
  void* lp_[2] = {lp + 128 + rank * S, lp + rank * S};
  void* wb_[2] = {lp + 128, lp};
  size_t sc[2] = {S, S};
  size_t bad[2] = {S, S + 4};
  assert(dtp_reducer_set_shard(r, 1, nullptr, nullptr, 2, m_, g_, s1_, nullptr, lp_, sc, wb_, 1, world) == -3);
  assert(dtp_reducer_set_shard(r, 1, (void*)fake_sgd, nullptr, 2, m_, g_, s1_, nullptr, lp_, bad, wb_, 1, world) == -5);
  assert(dtp_reducer_set_shard(r, 1, (void*)fake_sgd, nullptr, 2, m_, g_, s1_, nullptr, lp_, sc, wb_, 1, world) == 0);
  g_log.calls.clear();
  assert(dtp_reducer_mark_ready(r, 1, compute) < 0);        // pass not armed: refused, nothing launched
  assert(count_calls("reducescatter") == 0);
  void* r2 = dtp_reducer_create(transport, 2, ptrs, counts, kinds, 2, offs, idx);   // fresh counters
  assert(dtp_reducer_destroy(r) == 0);
  r = r2;
  assert(dtp_reducer_set_shard(r, 1, (void*)fake_sgd, nullptr, 2, m_, g_, s1_, nullptr, lp_, sc, wb_, 1, world) == 0);
  for (int pass = 0; pass < 2; ++pass) {
    const float hyper[4] = {0.5f, 0.9f, 0.0f, 1e-4f};
    const int iargs[2] = {0, pass == 0};
    assert(dtp_reducer_begin_pass(r, hyper, 4, iargs, 2) == 0);
    g_log.calls.clear();
    assert(dtp_reducer_mark_ready(r, 1, compute) == 1);     // bucket 0 complete
    assert(g_log.calls.size() == 5);
    assert(g_log.calls[0] == "record:1" && g_log.calls[1] == "wait:100:on:1");
    assert(g_log.calls[2] == "reducescatter:100:32:4");
    assert(g_log.calls[3] == std::string("sgd:100:32:1:1:0") + (pass == 0 ? "1" : "0"));
    assert(g_log.calls[4] == "allgather:100:32");
    assert(dtp_reducer_finalize(r, compute) == 1);          // bucket 1 never became ready: launched here
    assert(count_calls("reducescatter") == 2 && count_calls("sgd") == 2 && count_calls("allgather") == 2);
    assert(g_log.calls.back() == "wait:1:on:100");
  }
  assert(dtp_reducer_destroy(r) == 0);
}

// replicated optimizer-in-scheduler mode (the default data-parallel path): per bucket all-reduce ->
// fused optimizer over the WHOLE bucket on the comm stream, no all-gather; one consumer wait per pass
static void test_replicated_bucket_optimizer(void* transport, MockStream* compute) {
  static float master[256], mom[256];
  static unsigned short grad[256], lp[256];
  void* ptrs[2] = {grad + 128, grad};
  size_t counts[2] = {128, 128};
  int kinds[2] = {1, 1};
  int offs[3] = {0, 1, 2};
  int idx[2] = {1, 0};
  void* r = dtp_reducer_create(transport, 2, ptrs, counts, kinds, 2, offs, idx);
  assert(r);
  void* m_[2] = {master + 128, master};
  void* g_[2] = {grad + 128, grad};
  void* s1_[2] = {mom + 128, mom};
  void* lp_[2] = {lp + 128, lp};
  size_t sc[2] = {128, 128};
  assert(dtp_reducer_set_shard(r, 1, (void*)fake_sgd, nullptr, 2, m_, g_, s1_, nullptr, lp_, sc, lp_, 1, /*world=*/1) == 0);
  for (int pass = 0; pass < 2; ++pass) {
    const float hyper[4] = {0.5f, 0.9f, 0.0f, 1e-4f};
    const int iargs[2] = {0, pass == 0};
    assert(dtp_reducer_begin_pass(r, hyper, 4, iargs, 2) == 0);
    g_log.calls.clear();
    assert(dtp_reducer_mark_ready(r, 1, compute) == 1);
    assert(g_log.calls.size() == 4);
    assert(g_log.calls[0] == "record:1" && g_log.calls[1] == "wait:100:on:1");
    assert(g_log.calls[2].rfind("allreduce:100:128:", 0) == 0);
    assert(g_log.calls[3] == std::string("sgd:100:128:1:1:0") + (pass == 0 ? "1" : "0"));
    assert(dtp_reducer_finalize(r, compute) == 1);
    assert(count_calls("allreduce") == 2 && count_calls("sgd") == 2 && count_calls("allgather") == 0 &&
           count_calls("reducescatter") == 0);
    assert(g_log.calls.back() == "wait:1:on:100");
  }
  assert(dtp_reducer_destroy(r) == 0);
}

// 40 buckets per pass whose events stay pending for the whole pass (nothing completes until the
// consumer has synchronised): the pool must GROW past its nominal ring size instead of re-recording
// a live event (the mock's hipEventRecord asserts that), and shrink its appetite again afterwards;
// events recorded while the producer stream is being captured are never handed out again
static void test_event_pool_never_rerecords_live_events(void* transport, MockStream* compute) {
  constexpr int NB = 40;
  static float g[NB][16];
  void* ptrs[NB]; size_t counts[NB]; int kinds[NB]; int offs[NB + 1]; int idx[NB];
  for (int b = 0; b < NB; ++b) { ptrs[b] = g[b]; counts[b] = 16; kinds[b] = 0; offs[b] = b; idx[b] = b; }
  offs[NB] = NB;
  void* r = dtp_reducer_create(transport, NB, ptrs, counts, kinds, NB, offs, idx);
  assert(r);
  const int before = dtp_rccl_event_count(transport);
  g_log.keep_events_pending = true;
  for (int pass = 0; pass < 3; ++pass) {
    for (int p = 0; p < NB; ++p) assert(dtp_reducer_mark_ready(r, p, compute) == 1);
    assert(dtp_reducer_finalize(r, compute) == 0);
  }
  const int grown = dtp_rccl_event_count(transport);
  assert(grown >= 3 * (NB + 1) && grown > before && "pending events must not be reused");
  mock_complete_all_events();                 // the device caught up
  g_log.keep_events_pending = false;
  for (int pass = 0; pass < 50; ++pass) {
    for (int p = 0; p < NB; ++p) assert(dtp_reducer_mark_ready(r, p, compute) == 1);
    assert(dtp_reducer_finalize(r, compute) == 0);
  }
  assert(dtp_rccl_event_count(transport) == grown && "completed events are reused, the pool stops growing");
  // capture: every event of a captured pass is fresh and stays out of the reusable pool
  compute->capturing = true;
  for (int p = 0; p < NB; ++p) assert(dtp_reducer_mark_ready(r, p, compute) == 1);
  assert(dtp_reducer_finalize(r, compute) == 0);
  compute->capturing = false;
  assert(dtp_rccl_event_count(transport) == grown + NB + 1);
  for (int p = 0; p < NB; ++p) assert(dtp_reducer_mark_ready(r, p, compute) == 1);
  assert(dtp_reducer_finalize(r, compute) == 0);
  assert(dtp_rccl_event_count(transport) == grown + NB + 1);
  assert(dtp_reducer_destroy(r) == 0);
}

// bucket scheduler: params 0,1 -> bucket 0; param 2 -> buckets 1 and 2 (split); param 3 -> bucket 2
static void test_scheduler(void* transport, MockStream* compute) {
  float g[4][32];
  void* ptrs[3] = {g[0], g[1], g[2]};
  size_t counts[3] = {32, 16, 16};
  int kinds[3] = {0, 1, 1};
  int offs[5] = {0, 1, 2, 4, 5};
  int idx[5] = {0, 0, 1, 2, 2};
  assert(dtp_reducer_create(nullptr, 3, ptrs, counts, kinds, 4, offs, idx) == nullptr);
  int bad_idx[5] = {0, 0, 1, 7, 2};
  assert(dtp_reducer_create(transport, 3, ptrs, counts, kinds, 4, offs, bad_idx) == nullptr);
  int starve[5] = {0, 0, 0, 0, 0};                        // buckets 1 and 2 would never launch
  assert(dtp_reducer_create(transport, 3, ptrs, counts, kinds, 4, offs, starve) == nullptr);
  void* r = dtp_reducer_create(transport, 3, ptrs, counts, kinds, 4, offs, idx);
  assert(r);
  for (int pass = 0; pass < 3; ++pass) {
    g_log.calls.clear();
    assert(dtp_reducer_mark_ready(r, 9, compute) == -2);
    assert(dtp_reducer_mark_ready(r, 1, compute) == 0);     // bucket 0 still waits for param 0
    assert(dtp_reducer_mark_ready(r, 1, compute) == 0);     // duplicate (explicit notify + autograd hook)
    assert(dtp_reducer_mark_ready(r, 1, compute) == 0);
    assert(count_calls("allreduce") == 0);
    assert(dtp_reducer_mark_ready(r, 2, compute) == 1);     // completes bucket 1; bucket 2 waits for param 3
    assert(dtp_reducer_mark_ready(r, 0, compute) == 1);     // completes bucket 0
    assert(dtp_reducer_mark_ready(r, 0, compute) == 0);
    assert(count_calls("allreduce") == 2);
    assert(dtp_reducer_finalize(r, compute) == 1);          // param 3 produced no gradient: bucket 2 goes now
    assert(count_calls("allreduce") == 3);
    assert(g_log.calls.back() == "wait:1:on:100");          // consumer waits on the comm stream, once
  }
  // gradient accumulation: no launches while sync is off, counters still reset per pass
  assert(dtp_reducer_set_sync(r, 0) == 0);
  g_log.calls.clear();
  for (int p = 0; p < 4; ++p) assert(dtp_reducer_mark_ready(r, p, compute) == 0);
  assert(dtp_reducer_finalize(r, compute) == 0 && count_calls("allreduce") == 0 && count_calls("wait") == 0);
  assert(dtp_reducer_set_sync(r, 1) == 0);
  for (int p = 0; p < 4; ++p) dtp_reducer_mark_ready(r, p, compute);
  assert(count_calls("allreduce") == 3);
  assert(dtp_reducer_finalize(r, compute) == 0);
  unsigned long long passes = 0, launched = 0, dups = 0;
  assert(dtp_reducer_stats(r, &passes, &launched, &dups) == 0);
  assert(passes == 4 && launched == 12 && dups == 9);
  assert(dtp_reducer_destroy(r) == 0);
}

int main() {
  char id[256];
  assert(dtp_rccl_unique_id_bytes() == (int)sizeof(ncclUniqueId));
  assert(dtp_rccl_unique_id(id, 8) == -2);                       // buffer too small
  assert(dtp_rccl_unique_id(id, sizeof(id)) == (int)sizeof(ncclUniqueId));
  assert(dtp_rccl_create(id, 3, 2, 0) == nullptr);               // rank out of range
  assert(dtp_rccl_create(id, 0, 0, 0) == nullptr);
  void* t = dtp_rccl_create(id, 1, 4, 0);
  assert(t && g_comms == 1 && g_log.streams_created == 1);

  MockStream compute{1, false, 0, false};
  float buf[64];
  assert(dtp_rccl_allreduce_async(nullptr, buf, 64, 0, 1, &compute) == -2);
  assert(dtp_rccl_allreduce_async(t, nullptr, 64, 0, 1, &compute) == -2);
  assert(dtp_rccl_allreduce_async(t, buf, 0, 0, 1, &compute) == -2);

  // 1000 steps x (7 bucket launches + 7 waits): far more events than the ring holds
  unsigned long long want_bytes = 0;
  for (int step = 0; step < 1000; ++step) {
    for (int b = 0; b < 7; ++b) {
      const size_t before = g_log.calls.size();
      const int kind = b & 1;
      assert(dtp_rccl_allreduce_async(t, buf, 64, kind, 1, &compute) == 0);
      want_bytes += 64 * (kind ? 2 : 4);
      // exactly: record on the producer, comm stream waits on it, collective on the comm stream
      assert(g_log.calls.size() == before + 3);
      assert(g_log.calls[before] == "record:1");
      assert(g_log.calls[before + 1] == "wait:100:on:1");
      assert(g_log.calls[before + 2].rfind("allreduce:100:64:", 0) == 0);
      assert(g_log.calls[before + 2].substr(g_log.calls[before + 2].size() - 2) == ":4");   // ncclAvg
    }
    for (int b = 0; b < 7; ++b) {
      const size_t before = g_log.calls.size();
      assert(dtp_rccl_wait(t, &compute) == 0);
      assert(g_log.calls[before] == "record:100" && g_log.calls[before + 1] == "wait:1:on:100");
    }
    if (step % 100 == 0) g_log.calls.clear();
  }
  assert(g_log.events_created <= 64 && "event ring must be bounded");
  unsigned long long launched = 0, bytes = 0;
  assert(dtp_rccl_stats(t, &launched, &bytes) == 0 && launched == 7000 && bytes == want_bytes);

  test_scheduler(t, &compute);
  test_sharded_scheduler(t, &compute);
  test_replicated_bucket_optimizer(t, &compute);
  test_event_pool_never_rerecords_live_events(t, &compute);
  g_log.calls.clear();
  assert(dtp_rccl_broadcast(t, buf, 64, 0, 0, &compute) == 0);
  assert(g_log.calls.size() == 5 && g_log.calls[2] == "broadcast:100:64:0" && g_log.calls[4] == "wait:1:on:100");
  assert(dtp_rccl_sync(t) == 0);
  assert(dtp_rccl_destroy(t) == 0);
  assert(g_comms == 0 && g_log.streams_destroyed == 1 && g_log.events_destroyed == g_log.events_created);
  assert(dtp_rccl_destroy(nullptr) == 0);

  // event creation failure surfaces as an error code, not a crash or a leak
  g_log = MockLog();
  g_log.fail_event_create_after = 0;
  void* t2 = dtp_rccl_create(id, 0, 1, 0);
  assert(t2 && dtp_rccl_allreduce_async(t2, buf, 64, 0, 1, &compute) == -3);
  assert(dtp_rccl_wait(t2, &compute) == -3);
  assert(dtp_rccl_destroy(t2) == 0);
  std::puts("transport host test ok");
  return 0;
}
