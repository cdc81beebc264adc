// RCCL, bound at run time.
//
// The single-GPU library has no link-time dependency on RCCL: the collective
// entry points are looked up with dlopen the first time a slab communicator is
// created.  Transport between the z-slabs of different ranks is
// ncclSend/ncclRecv (point-to-point over xGMI: each pair of neighbouring slabs
// has its own direct link) and ncclAllReduce for the convergence scalars.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace fsamd {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommSplit) CommSplit = nullptr;   // optional: absent from old builds

    // Returns the process-wide binding or nullptr (err says why).
    static Rccl* get(std::string& err) {
        static Rccl inst;
        static bool tried = false;
        static std::string failure;
        static std::mutex mu;   // handles may be created from several host threads
        std::lock_guard<std::mutex> lock(mu);
        if (!tried) {
            tried = true;
            // FSAMD_RCCL_LIBRARY names the library to bind instead of the system's RCCL (a site build, or the
            // shared-memory stand-in the tests use to run several ranks on one GPU); if set it must load.
            const char* forced = getenv("FSAMD_RCCL_LIBRARY");
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            if (forced && *forced) {
                inst.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            } else {
                for (const char* n : names) {
                    inst.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                    if (inst.handle) break;
                }
            }
            if (!inst.handle) {
                failure = std::string("cannot load RCCL: ") + dlerror();
            } else {
                bool ok = true;
                auto sym = [&](const char* name) {
                    void* p = dlsym(inst.handle, name);
                    if (!p) { ok = false; failure = std::string("RCCL symbol missing: ") + name; }
                    return p;
                };
                inst.GetUniqueId = (decltype(inst.GetUniqueId))sym("ncclGetUniqueId");
                inst.CommInitRank = (decltype(inst.CommInitRank))sym("ncclCommInitRank");
                inst.CommDestroy = (decltype(inst.CommDestroy))sym("ncclCommDestroy");
                inst.Send = (decltype(inst.Send))sym("ncclSend");
                inst.Recv = (decltype(inst.Recv))sym("ncclRecv");
                inst.GroupStart = (decltype(inst.GroupStart))sym("ncclGroupStart");
                inst.GroupEnd = (decltype(inst.GroupEnd))sym("ncclGroupEnd");
                inst.AllReduce = (decltype(inst.AllReduce))sym("ncclAllReduce");
                inst.GetErrorString = (decltype(inst.GetErrorString))sym("ncclGetErrorString");
                if (ok) inst.CommSplit = (decltype(inst.CommSplit))dlsym(inst.handle, "ncclCommSplit");
                if (!ok) { dlclose(inst.handle); inst.handle = nullptr; }
            }
        }
        if (!inst.handle) { err = failure; return nullptr; }
        return &inst;
    }
};

}  // namespace fsamd
