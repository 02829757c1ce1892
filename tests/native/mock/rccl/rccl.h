// Host-only stand-in for <rccl/rccl.h> (see ../hip/hip_runtime.h).
#pragma once
#include <cstddef>
#include "../hip/hip_runtime.h"

typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
struct MockComm { int rank, world; bool destroyed; };
typedef MockComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum ncclDataType_t { ncclFloat32 = 7, ncclBfloat16 = 9 };
enum ncclRedOp_t { ncclSum = 0, ncclAvg = 4 };

ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root,
                           ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduceScatter(const void* send, void* recv, size_t recvcount, ncclDataType_t dt, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream);
