// Shared host-side helpers: error plumbing, HIP status checks, RAII device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace xrl {

// Every HIP return code becomes a C++ exception, which the extern "C" layer turns into
// xrl_last_error() (SURVEY.md 8b "Error convention": the reference aborts instead).
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

[[noreturn]] inline void fail(const std::string& msg) { throw Error(msg); }

#define XRL_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            ::xrl::fail(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr + \
                        " (" __FILE__ ":" + std::to_string(__LINE__) + ")");               \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        release();
        XRL_HIP(hipMalloc(&p, bytes ? bytes : 16));
        cap = bytes;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
    template <class T> void upload(const std::vector<T>& v) {
        reserve(v.size() * sizeof(T));
        if (!v.empty()) XRL_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void upload_raw(const void* src, size_t bytes) {
        reserve(bytes);
        if (bytes) XRL_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    }
};

// Pinned host staging buffer (grow-only).
struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        XRL_HIP(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault));
        cap = bytes;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace xrl
