// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/fluidsim_amd.h"

namespace fsamd {

inline std::string hip_err(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return buf;
}

// Evaluates a HIP call; on failure records the text in `errstr` and returns FS_EHIP
// from the enclosing function.
#define FS_HIP(errstr, call)                                                     \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) {                                                 \
            (errstr) = ::fsamd::hip_err(#call, e__, __FILE__, __LINE__);         \
            return FS_EHIP;                                                      \
        }                                                                        \
    } while (0)

#define FS_TRY(call)                  \
    do {                              \
        int rc__ = (call);            \
        if (rc__ != FS_OK) return rc__; \
    } while (0)

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// RAII device selection for one entry point: a handle may be driven from any
// host thread, so every call makes the handle's device current and restores
// the caller's afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

}  // namespace fsamd
