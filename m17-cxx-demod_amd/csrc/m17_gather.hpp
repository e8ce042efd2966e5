// Multi-GPU exchange of the decoded frame records over RCCL (host code).  SURVEY §8(e): channels are independent, a node is
// sharded by contiguous channel ranges, one process (rank) per GPU, and the only exchange is the gather of the 64-byte frame
// records at the end of a run.  librccl is bound at first use (dlopen of the SONAME: a host process that already carries an
// RCCL — e.g. through torch — shares it, a plain C++ host gets /opt/rocm/lib's); nothing else of the library needs it.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

namespace m17 {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;   // optional: a communicator whose exchange did not end in time is given up through it
    bool ok = false;
};

inline const Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) return;
#define M17_RCCL_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, #sym))
        M17_RCCL_SYM(GetUniqueId, ncclGetUniqueId);
        M17_RCCL_SYM(CommInitRank, ncclCommInitRank);
        M17_RCCL_SYM(CommDestroy, ncclCommDestroy);
        M17_RCCL_SYM(AllGather, ncclAllGather);
        M17_RCCL_SYM(Send, ncclSend);
        M17_RCCL_SYM(Recv, ncclRecv);
        M17_RCCL_SYM(GroupStart, ncclGroupStart);
        M17_RCCL_SYM(GroupEnd, ncclGroupEnd);
        M17_RCCL_SYM(CommAbort, ncclCommAbort);
#undef M17_RCCL_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    });
    return r;
}

}  // namespace m17
