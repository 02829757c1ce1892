// Host-only stand-in for <hip/hip_runtime.h>: just enough of the stream/event API for
// csrc/runtime/rccl_transport.cpp, recording every call so that the transport's host logic
// (event ring, ordering of record/wait/collective, error paths, teardown) can run under
// AddressSanitizer/UBSan on a CPU-only box.  Not used in any product build.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorNotReady = 600;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
struct MockStream { int id; bool non_blocking; int priority; bool capturing = false; };
struct MockEvent { int id; int records; const MockStream* last_stream; bool destroyed; bool pending = false; };
typedef MockStream* hipStream_t;
typedef MockEvent* hipEvent_t;
constexpr unsigned hipEventDisableTiming = 2;
constexpr unsigned hipStreamNonBlocking = 1;

struct MockLog {
  std::vector<std::string> calls;
  int events_created = 0, events_destroyed = 0, streams_created = 0, streams_destroyed = 0;
  int fail_event_create_after = -1;     // inject hipEventCreate failures
  bool keep_events_pending = false;     // recorded events stay "not ready" until mock_complete_all_events()
};
MockLog& mock_log();

hipError_t hipSetDevice(int);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st);
