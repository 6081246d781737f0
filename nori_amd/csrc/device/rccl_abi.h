/*
 * rccl_abi.h -- the handful of RCCL types, constants and entry points the device group uses (group.hip), declared by hand:
 * librccl is dlopen'ed at the first group of more than one device, so a one-GPU build needs neither <rccl/rccl.h> nor the
 * library.  Hand-copied declarations are only as good as their agreement with the real header: tests/abi/rccl_abi_check.cpp
 * includes BOTH this file and /opt/rocm/include/rccl/rccl.h and static_asserts, for every entry point below, the same number
 * of parameters of the same ABI class (pointer / integer of the same size) and the same return class, and the values of the
 * constants -- it runs in the CPU suite (tests/test_build_guards.py), where a multi-GPU node has never been available.
 * Everything lives in namespace nori_rccl so that both sets of names can meet in one translation unit.
 */
#pragma once
#include <cstddef>

#include <hip/hip_runtime_api.h>

namespace nori_rccl {

typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                               /* rccl.h: an enum (int-sized) */
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclFloat = 7, ncclSum = 0;              /* rccl.h: ncclDataType_t ncclFloat32 = ncclFloat = 7, ncclRedOp_t ncclSum = 0 */
constexpr int kMinVersion = 20000;                      /* the values above are those of NCCL / RCCL 2.x */

typedef ncclResult_t (*CommInitAll_t)(ncclComm_t *comm, int ndev, const int *devlist);
typedef ncclResult_t (*CommDestroy_t)(ncclComm_t comm);
typedef const char *(*GetErrorString_t)(ncclResult_t result);
typedef ncclResult_t (*GetVersion_t)(int *version);
typedef ncclResult_t (*Reduce_t)(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, int root, ncclComm_t comm, hipStream_t stream);
typedef ncclResult_t (*Send_t)(const void *sendbuff, size_t count, int datatype, int peer, ncclComm_t comm, hipStream_t stream);
typedef ncclResult_t (*Recv_t)(void *recvbuff, size_t count, int datatype, int peer, ncclComm_t comm, hipStream_t stream);
typedef ncclResult_t (*GroupStart_t)();
typedef ncclResult_t (*GroupEnd_t)();

} // namespace nori_rccl
