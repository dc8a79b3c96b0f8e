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

// Every launching entry point runs on the device that owns its buffers, whatever the caller's current device is (the reference
// API takes `device=cuda:N`; the stream argument alone cannot say which device — torch's default stream of any device is the
// null handle).  With one visible device (one process per GPU, the normal deployment) the guard costs nothing.
int visible_devices();
// compute units of the CURRENT device (the one a launch that follows goes to); cached per device ordinal, thread-safe
int cu_count();
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const void *device_ptr) {
        if (visible_devices() <= 1 || !device_ptr) return;
        hipPointerAttribute_t attr;
        int cur = 0;
        if (hipPointerGetAttributes(&attr, device_ptr) != hipSuccess || hipGetDevice(&cur) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        if (attr.device != cur && hipSetDevice(attr.device) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define G4C_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            g4c::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)

// a / count for the mean of a segment (count >= 1), BIT FOR BIT the IEEE quotient `a / (float)count`, in 4 vector instructions per
// value + 3 per count instead of the ~11 of the compiler's division (v_div_scale x2, v_rcp, five FMAs, v_div_fmas, v_div_fixup), of
// which only the reciprocal is shared between the values of a count.  y = the correctly rounded 1 / count (v_rcp_f32 + one Newton
// step: checked against the exact reciprocal for every count up to MEAN_DIV_MAX_COUNT by test_mean_div_is_the_ieee_quotient);
// q0 = a y, r = a - count q0 exactly (FMA), q1 = q0 + r y rounded once: the correctly rounded quotient (Markstein's theorem — y
// correctly rounded, q0 within an ulp) as long as nothing underflows.  Everything else — larger counts, quotients below 2^-100 (the
// residual could underflow), zeros, infinities, NaNs — takes the division itself, for the whole wave (a wave-uniform branch that
// data of O(1) never takes).
constexpr int MEAN_DIV_MAX_COUNT = 4096;
typedef float mean_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mean_f32x4 mean_div4(mean_f32x4 a, int count) {
    const float c = (float)count;
    const float y0 = __builtin_amdgcn_rcpf(c);
    const float y = fmaf(fmaf(-c, y0, 1.f), y0, y0);
    mean_f32x4 q;
    bool rare = count > MEAN_DIV_MAX_COUNT;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float q0 = a[e] * y;
        q[e] = fmaf(fmaf(-c, q0, a[e]), y, q0);
        // (true for NaN as well; a numerator of +0 — an empty segment, a padding row — gives the exact quotient +0 on the fast path and
        // must not send the whole wave down the slow one)
        rare |= (__builtin_bit_cast(unsigned, a[e]) != 0u) && !(fabsf(q[e]) >= 0x1p-100f);
    }
    if (__builtin_amdgcn_ballot_w64(rare) != 0ull) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = a[e] / c;
    }
    return q;
}

// Branch-free activations on the hardware transcendental unit (v_exp_f32 / v_rcp_f32).
// Absolute error vs torch's F.selu / torch.tanh is ~1e-7 (fp32 round-off class), far inside the
// 1e-4 per-block parity tolerance; a divergent `x > 0 ? ... : exp(...)` costs two branches and an
// exec-mask round trip per element in the fused epilogues.
__device__ __forceinline__ float selu_f(float x) {
    const float alpha = 1.6732632423543772848170429916717f;   // torch's SELU constants
    const float scale = 1.0507009873554804934193349852946f;
    // exp(min(x, 0)) as v_exp_f32 with the clamp modifier (the median folds into the instruction: no v_min), then
    // scale * max(x, 0) + (scale * alpha * e - scale * alpha), whose second term is exactly 0 for x >= 0: four vector
    // instructions + the exponential, and the same bits as the select form (x > 0 ? scale * x : scale * alpha * (e - 1))
    const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * 1.4426950408889634f), 0.f, 1.f);
    return fmaf(fmaxf(x, 0.f), scale, fmaf(scale * alpha, e, -scale * alpha));
}

__device__ __forceinline__ float tanh_f(float x) {
    const float ax = fminf(fabsf(x), 20.f);
    const float t = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);    // exp(2|x|)
    const float r = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
    return copysignf(r, x);
}

__device__ __forceinline__ float apply_act(float x, int act) {
    const float s = selu_f(x), t = tanh_f(x);
    return act == G4C_ACT_SELU ? s : (act == G4C_ACT_TANH ? t : x);
}

template <int ACT>
__device__ __forceinline__ float act_t(float x) {
    if (ACT == G4C_ACT_SELU) return selu_f(x);
    if (ACT == G4C_ACT_TANH) return tanh_f(x);
    return x;
}

}  // namespace g4c
