// common.h — host-side plumbing shared by the C ABI implementation.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "slideo_amd.h"

namespace slideo {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define HIP_CHECK(expr)                                                                       \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            ::slideo::fail(SLIDEO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                               \
    } while (0)

// Device buffer that only ever grows; contents are NOT preserved across growth.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        release();
        size_t want = bytes + bytes / 8 + 256;
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    void reserve_cap(size_t want) {            // exactly `want` bytes of capacity (no growth slack)
        if (want <= cap) return;
        release();
        HIP_CHECK(hipMalloc(&p, want));
        cap = want;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host buffer (grows, not preserved).
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        HIP_CHECK(hipHostMalloc(&p, bytes + 256, hipHostMallocDefault));
        cap = bytes + 256;
    }
    void reserve_cap(size_t want) { if (want > cap) reserve(want - 256); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace slideo
