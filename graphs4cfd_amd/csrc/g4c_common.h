// Shared helpers for libg4c.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/g4c.h"

namespace g4c {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return G4C_ELAUNCH;
    }
    return G4C_OK;
}

#define G4C_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            g4c::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)

// torch's SELU constants
__device__ __forceinline__ float selu_f(float x) {
    const float alpha = 1.6732632423543772848170429916717f;
    const float scale = 1.0507009873554804934193349852946f;
    return x > 0.f ? scale * x : (scale * alpha) * (__expf(x) - 1.0f);
}

__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == G4C_ACT_SELU) return selu_f(x);
    if (act == G4C_ACT_TANH) return tanhf(x);
    return x;
}

}  // namespace g4c
