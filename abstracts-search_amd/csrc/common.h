// common.h -- host-side plumbing shared by the C-ABI translation units:
// error capture (no exception crosses the ABI), grow-only device buffers,
// host/device pointer classification.
#pragma once
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>

namespace mi {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline std::string &last_error() {
    static thread_local std::string e;
    return e;
}

#define MI_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t mi_e_ = (expr);                                                    \
        if (mi_e_ != hipSuccess)                                                      \
            throw mi::Error(std::string(#expr) + " failed: " + hipGetErrorString(mi_e_)); \
    } while (0)

#define MI_REQUIRE(cond, msg)                         \
    do {                                              \
        if (!(cond)) throw mi::Error(std::string(msg)); \
    } while (0)

template <class F>
int guard(F &&f) {
    try {
        f();
        last_error().clear();
        return 0;
    } catch (const std::exception &e) {
        last_error() = e.what();
        return 1;
    } catch (...) {
        last_error() = "unknown error";
        return 1;
    }
}

// Grow-only device allocation.  hipFree synchronises the device, so growing a
// workspace that an in-flight kernel still reads is safe (just slow): sizes
// settle after the first call of a given shape.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) {
        o.p = nullptr;
        o.cap = 0;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    void *reserve(size_t bytes) {
        if (bytes > cap) {
            release();
            size_t want = bytes + bytes / 8 + 256;
            MI_HIP(hipMalloc(&p, want));
            cap = want;
        }
        return p;
    }
    template <class T>
    T *as(size_t count) {
        return static_cast<T *>(reserve(count * sizeof(T)));
    }
    template <class T>
    T *get() const {
        return static_cast<T *>(p);
    }
};

// true if `ptr` is dereferenceable by a kernel on the current device.
inline bool is_device_ptr(const void *ptr) {
    if (!ptr) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'd host memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// roctx ranges around the stages of a call (coarse / lut / scan / rerank / exchange; embed / layer / pool), so that a
// rocprofv3 --marker-trace summary groups the launches by stage.  The roctx library is bound at run time and only when a
// profiler brought it into the process (or MI_ROCTX=1 asks for it): without one a Range is a branch on a cached flag.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    Roctx() {
        const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        const char *e = std::getenv("MI_ROCTX");
        if (e && e[0] == '0') return;
        const bool force = e && e[0] != '0';
        for (const char *n : names) {
            void *lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (force ? 0 : RTLD_NOLOAD));
            if (!lib) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            on = push && pop;
            if (on) return;
        }
    }
};
inline Roctx &roctx() {
    static Roctx r;
    return r;
}
struct Range {
    bool on;
    explicit Range(const char *name) : on(roctx().on) {
        if (on) roctx().push(name);
    }
    ~Range() {
        if (on) roctx().pop();
    }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device): several host threads may launch through one
// handle, and a second device's first launch needs the attribute as much as the first one's (a plain `static bool` was
// both a data race and process-wide).
inline void set_max_dynamic_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    MI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return;
    MI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, dev});
}

struct DeviceGuard {
    int prev = 0;
    explicit DeviceGuard(int dev) {
        MI_HIP(hipGetDevice(&prev));
        if (dev != prev) MI_HIP(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace mi
