// mage_common.h -- status plumbing and small RAII helpers shared by the host side of libmageslam_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mage_ba.h"

#define MAGE_EXPORT extern "C" __attribute__((visibility("default")))

namespace mage {

std::string& last_error_ref();

inline mage_status fail(mage_status s, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return s;
}

#define MAGE_HIP(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess)                                                                                  \
            return ::mage::fail(_e == hipErrorOutOfMemory ? MAGE_ERR_OUT_OF_MEMORY : MAGE_ERR_DEVICE,         \
                                "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
    } while (0)

#define MAGE_TRY(expr)                                  \
    do {                                                \
        mage_status _s = (expr);                        \
        if (_s != MAGE_OK) return _s;                   \
    } while (0)

// Checks that a usable gfx950 device exists; the HIP path never falls back to a CPU.
mage_status select_device(int requested, int* chosen);

// Grow-only device buffer.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    mage_status reserve(size_t n)
    {
        if (n <= cap) return MAGE_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 16;
        MAGE_HIP(hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
        cap = want;
        return MAGE_OK;
    }
    mage_status upload(const T* src, size_t n, hipStream_t st)
    {
        MAGE_TRY(reserve(n));
        if (n) MAGE_HIP(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
        return MAGE_OK;
    }
};

}  // namespace mage
