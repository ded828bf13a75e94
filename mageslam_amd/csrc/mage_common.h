// mage_common.h -- status plumbing and small RAII helpers shared by the host side of libmageslam_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <string>
#include <vector>

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

// Makes `device` the calling thread's current HIP device for the lifetime of the object and puts the caller's device back
// afterwards: a host process that drives several GPUs from one thread (or keeps its own HIP work on another device) must not
// find its current device changed by a library call or by a handle's destructor.
// Wait for everything queued on `st` so far.  hipStreamSynchronize may put the thread to sleep and costs tens of microseconds
// to wake; a frame's worth of front-end work is ~50 us, so the wait first polls an event for a short while (the BA scalar
// read-back does the same) and only then blocks.
inline hipError_t wait_stream_briefly_spinning(hipStream_t st, hipEvent_t ev, int spin_us = 300)
{
    hipError_t e = hipEventRecord(ev, st);
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        e = hipEventQuery(ev);
        if (e == hipSuccess) return hipSuccess;
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
    return hipEventSynchronize(ev);
}

struct DeviceScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            changed = (err == hipSuccess);
        }
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
    ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
};
#define MAGE_DEVICE_SCOPE(dev)            \
    ::mage::DeviceScope _mage_scope(dev); \
    MAGE_HIP(_mage_scope.err)

// Checks that a usable gfx950 device exists; the HIP path never falls back to a CPU.
mage_status select_device(int requested, int* chosen);

// Device-memory cache.  The reference builds a bundler per optimisation and destroys it afterwards (BundleAdjust.cpp:348-351);
// a 1k-pose problem holds ~0.8 GB, and a fresh hipMalloc of that size costs ~25 ms per handle (allocation, first use, hipFree).
// Buffers of a destroyed handle are therefore parked here (per device, bounded, MAGE_DEVICE_CACHE_MB, default 4096; 0 turns
// the cache off) and handed to the next handle.  Contents are undefined, as with hipMalloc.
mage_status cached_device_alloc(void** p, size_t bytes, int* device, size_t* granted);
void cached_device_release(void* p, size_t bytes, int device);     // caller guarantees no work in flight touches p

// Streams are parked the same way: the first kernel on a new stream pays for its hardware queue and scratch (~16 ms).
mage_status cached_stream_acquire(int device, hipStream_t* out);
void cached_stream_release(int device, hipStream_t st);             // caller has synchronised st

// Pinned host memory, parked like the device buffers.  hipMemcpyAsync from pageable memory makes the runtime pin and
// unpin the source around every copy (several ms per 100 MB, paid at the next synchronisation); lists that are built for
// upload are therefore built directly in pinned blocks, so the copy is one asynchronous DMA that overlaps the host work after it.
mage_status cached_pinned_alloc(void** p, size_t bytes, size_t* granted);
void cached_pinned_release(void* p, size_t bytes);                   // caller guarantees no copy in flight reads p

// Bump allocator over pinned blocks; everything taken from it lives until release().
struct PinnedArena {
    struct Block { char* p; size_t cap, used; };
    std::vector<Block> blocks;
    size_t min_block = (size_t)32 << 20;       // the structure build stages tens of MB; small users (a state read-back) say what they need
    PinnedArena() = default;
    explicit PinnedArena(size_t min_block_bytes) : min_block(min_block_bytes) {}
    PinnedArena(const PinnedArena&) = delete;
    PinnedArena& operator=(const PinnedArena&) = delete;
    ~PinnedArena() { release(); }
    void release()
    {
        for (Block& b : blocks) cached_pinned_release(b.p, b.cap);
        blocks.clear();
    }
    template <typename T>
    mage_status take(size_t n, T** out)
    {
        const size_t need = (n * sizeof(T) + 255) & ~(size_t)255;
        if (blocks.empty() || blocks.back().used + need > blocks.back().cap) {
            void* q = nullptr; size_t got = 0;
            MAGE_TRY(cached_pinned_alloc(&q, std::max<size_t>(need, min_block), &got));
            blocks.push_back({ static_cast<char*>(q), got, 0 });
        }
        Block& b = blocks.back();
        *out = reinterpret_cast<T*>(b.p + b.used);
        b.used += need;
        return MAGE_OK;
    }
};

// A fixed-size array in pinned host memory (from the pinned cache): what the host keeps AND what is uploaded as it is -- the
// observation records of a handle go to the device with one DMA straight from where the setters wrote them.
template <typename T>
struct PinnedVec {
    T* p = nullptr;
    size_t n = 0, bytes = 0;
    PinnedVec() = default;
    PinnedVec(const PinnedVec&) = delete;
    PinnedVec& operator=(const PinnedVec&) = delete;
    ~PinnedVec() { if (p) cached_pinned_release(p, bytes); }
    mage_status assign(size_t count, const T& v)
    {
        if (count * sizeof(T) > bytes) {
            if (p) { cached_pinned_release(p, bytes); p = nullptr; bytes = 0; }
            void* q = nullptr;
            MAGE_TRY(cached_pinned_alloc(&q, std::max<size_t>(count, 1) * sizeof(T), &bytes));
            p = static_cast<T*>(q);
        }
        n = count;
        std::fill(p, p + count, v);
        return MAGE_OK;
    }
    // the same without initialising the elements: for a caller that is about to overwrite all of them (or fills them later)
    mage_status resize_uninitialized(size_t count)
    {
        if (count * sizeof(T) > bytes) {
            if (p) { cached_pinned_release(p, bytes); p = nullptr; bytes = 0; }
            void* q = nullptr;
            MAGE_TRY(cached_pinned_alloc(&q, std::max<size_t>(count, 1) * sizeof(T), &bytes));
            p = static_cast<T*>(q);
        }
        n = count;
        return MAGE_OK;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* begin() { return p; }
    T* end() { return p + n; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
};

// Grow-only device buffer.  It either owns its memory (from the device cache) or is a VIEW into a larger buffer somebody else owns
// (alias(): the small-problem image of ba_host.hip puts every list of a problem into ONE device buffer filled by ONE copy).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    size_t bytes = 0;
    int device = 0;
    bool aliased = false;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p && !aliased) cached_device_release(p, bytes, device); }       // handle destructors synchronise their stream first
    // view of `count` elements at q (owned elsewhere).  The caller has made sure nothing in flight still uses memory this buffer owned.
    void alias(T* q, size_t count)
    {
        if (p && !aliased) cached_device_release(p, bytes, device);
        p = q; cap = count; bytes = 0; aliased = true;
    }
    void drop_alias() { if (aliased) { p = nullptr; cap = 0; bytes = 0; aliased = false; } }     // a view that was not renewed by the build that re-cut the image
    mage_status reserve(size_t n)
    {
        if (n <= cap) return MAGE_OK;
        if (aliased) { p = nullptr; cap = 0; bytes = 0; aliased = false; }   // a view is never grown: take memory of its own
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }                // growth inside a live handle: hipFree synchronises
        const size_t want = n + n / 8 + 16;
        void* q = nullptr;
        MAGE_TRY(cached_device_alloc(&q, want * sizeof(T), &device, &bytes));
        p = static_cast<T*>(q);
        cap = bytes / sizeof(T);
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
