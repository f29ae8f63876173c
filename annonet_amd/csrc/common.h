// common.h — error plumbing and small device-memory helpers shared by the library's translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/annonet_hip.h"

namespace anh {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& message);  // api.cpp: thread-local text behind anh_last_error()

[[noreturn]] inline void fail(int code, const std::string& msg) { throw Error(code, msg); }

#define ANH_REQUIRE(cond, msg) \
    do { if (!(cond)) ::anh::fail(ANH_ERR_INVALID, std::string(msg) + " [" #cond "]"); } while (0)

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e == hipSuccess) return;
    const int code = (e == hipErrorOutOfMemory) ? ANH_ERR_OOM : ANH_ERR_DEVICE;
    (void)hipGetLastError();
    fail(code, std::string(what) + ": " + hipGetErrorString(e) + " (" + file + ":" + std::to_string(line) + ")");
}
#define HIP_CHECK(expr) ::anh::hip_check((expr), #expr, __FILE__, __LINE__)

// Host-side waits of a handle that drives SEVERAL devices are bounded (host B of DESIGN.md §6): a collective that never completes —
// a replica that did not join, a failed link — would otherwise leave StartTraining / synchronize blocked for ever, where the
// one-process-per-GPU host has torch.distributed's collective timeout.  Deadline: ANH_REPLICA_TIMEOUT_S (default 180 s); a wait that
// passes it throws ANH_ERR_DEVICE, which the C++ shim rethrows and the reference's mains turn into a positive exit code
// (annonet_train_main.cpp:616-620,640-644).  bounded = false is the plain blocking call (single-device handles: no collective).
int replica_timeout_seconds();                       // api.cpp
void wait_event(hipEvent_t e, bool bounded);         // api.cpp
void wait_stream(hipStream_t s, bool bounded);       // api.cpp

// RAII device buffer (grow-only scratch)
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    void reserve(size_t n) {
        if (n <= bytes) return;
        release();
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel (and again only if a launch needs more)
void ensure_dynamic_lds(const void* kernel, size_t bytes);

}  // namespace anh
