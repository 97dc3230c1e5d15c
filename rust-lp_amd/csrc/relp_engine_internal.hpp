// relp_engine_internal.hpp -- helpers shared by the relp_engine*.cpp translation units.
#pragma once
#include "relp_engine.hpp"

namespace relp {

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        if (!hip_ok((expr), #expr)) return RELP_E_HIP;                       \
    } while (0)

static inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Zero-initialised device buffer.  hipMemset runs on the null stream, which does not order with the
// engine's non-blocking stream: the device is synchronised before the buffer is handed out, so a kernel
// enqueued on stream_ right afterwards cannot be overtaken by the memset.
template <class T>
static hipError_t dev_alloc(T** p, int64_t count) {
    if (count < 1) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T));
    if (e != hipSuccess) return e;
    e = hipMemset(*p, 0, (size_t)count * sizeof(T));
    if (e != hipSuccess) return e;
    return hipDeviceSynchronize();
}

}  // namespace relp
