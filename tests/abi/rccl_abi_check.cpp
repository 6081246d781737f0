// TEST ONLY -- compiled (not run) by tests/test_build_guards.py: the hand-declared RCCL interface of
// nori_amd/csrc/device/rccl_abi.h against the installed <rccl/rccl.h>.  A static_assert that fails is a compile error.
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -fsyntax-only tests/abi/rccl_abi_check.cpp
#include <type_traits>

#include <rccl/rccl.h>

#include "../../nori_amd/csrc/device/rccl_abi.h"

// two parameter (or return) types are the same to the calling convention: both pointers, or both integers / enums of one size
template <class A, class B> constexpr bool abi_same =
    (std::is_pointer_v<A> && std::is_pointer_v<B>) ||
    ((std::is_integral_v<A> || std::is_enum_v<A>) && (std::is_integral_v<B> || std::is_enum_v<B>) && sizeof(A) == sizeof(B));

template <class F, class G> struct same_signature : std::false_type {};
template <class R1, class... A1, class R2, class... A2>
struct same_signature<R1 (*)(A1...), R2 (*)(A2...)> {
    template <bool SameCount, class = void> struct each : std::false_type {};
    template <class D> struct each<true, D> : std::bool_constant<(abi_same<A1, A2> && ...)> {};
    static constexpr bool value = abi_same<R1, R2> && each<sizeof...(A1) == sizeof...(A2)>::value;
};

#define CHECK(ours, theirs) static_assert(same_signature<nori_rccl::ours, decltype(&theirs)>::value, #theirs ": rccl_abi.h disagrees with rccl.h")
CHECK(CommInitAll_t, ncclCommInitAll);
CHECK(CommDestroy_t, ncclCommDestroy);
CHECK(GetErrorString_t, ncclGetErrorString);
CHECK(GetVersion_t, ncclGetVersion);
CHECK(Reduce_t, ncclReduce);
CHECK(Send_t, ncclSend);
CHECK(Recv_t, ncclRecv);
CHECK(GroupStart_t, ncclGroupStart);
CHECK(GroupEnd_t, ncclGroupEnd);

static_assert((int) ::ncclSuccess == nori_rccl::ncclSuccess, "ncclSuccess");
static_assert((int) ::ncclFloat32 == nori_rccl::ncclFloat && (int) ::ncclFloat == nori_rccl::ncclFloat, "ncclFloat32");
static_assert((int) ::ncclSum == nori_rccl::ncclSum, "ncclSum");
static_assert(sizeof(::ncclResult_t) == sizeof(nori_rccl::ncclResult_t) && sizeof(::ncclDataType_t) == sizeof(int) && sizeof(::ncclRedOp_t) == sizeof(int), "enum sizes");
static_assert(sizeof(::ncclComm_t) == sizeof(nori_rccl::ncclComm_t), "ncclComm_t is a pointer");
static_assert(NCCL_VERSION_CODE >= nori_rccl::kMinVersion, "the installed RCCL is older than the interface declared in rccl_abi.h");
// the negative control of the checker itself: a signature with one parameter less, or an integer where a pointer belongs, must NOT pass
static_assert(!same_signature<nori_rccl::Send_t, decltype(&ncclReduce)>::value, "the checker accepts a wrong arity");
static_assert(!same_signature<nori_rccl::ncclResult_t (*)(int, size_t, int, int, nori_rccl::ncclComm_t, hipStream_t), decltype(&ncclSend)>::value, "the checker accepts an integer for a pointer");

int main() { return 0; }
