// Host-only ASYNCHRONOUS stand-in for <hip/hip_runtime.h>, for tests/native/ring_tsan_test.cpp: every stream
// is a worker thread that executes its queue in order (copies, event records, event waits), so the
// "device" really runs concurrently with the host threads and ThreadSanitizer sees the hand-overs of
// csrc/runtime/pinned_ring.cpp as what they are: a pinned slot is read by the copy stream's thread while
// producer threads refill other slots, a device buffer is written by the copy stream and read by the
// compute stream.  A missing wait in the ring's protocol is a data race here.  Not used in any product build.
#pragma once
#include <cstddef>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorNotReady = 600;
struct MockAStream;
struct MockAEvent;
typedef MockAStream* hipStream_t;
typedef MockAEvent* hipEvent_t;
constexpr unsigned hipEventDisableTiming = 2;
constexpr unsigned hipStreamNonBlocking = 1;
constexpr unsigned hipHostMallocDefault = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1 };

hipError_t hipSetDevice(int);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
