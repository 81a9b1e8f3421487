// Shared host-side helpers for libspeedyml_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/speedyml_hip.h"

namespace sml {

inline std::string &last_error_ref()
{
    static thread_local std::string s;
    return s;
}

inline int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define SML_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return sml::fail(SML_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define SML_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return sml::fail(SML_ERR_ARG, __VA_ARGS__); \
    } while (0)

template <class T>
inline int dev_upload(T **dst, const T *src, size_t count)
{
    SML_HIP(hipMalloc((void **)dst, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    if (count) SML_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return SML_OK;
}

template <class T>
inline int dev_zeros(T **dst, size_t count)
{
    SML_HIP(hipMalloc((void **)dst, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    SML_HIP(hipMemset(*dst, 0, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    return SML_OK;
}

inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

}  // namespace sml
