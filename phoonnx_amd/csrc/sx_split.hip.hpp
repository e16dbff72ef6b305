// sx_split.hip.hpp — the fp16 two-plane split of an fp32 value (the f16x3 arithmetic's operand format) and the
// range-guard bookkeeping that goes with it; shared by the split-operand conv engine (conv_sx_engine.hip.hpp), which
// consumes the planes, and the f32 engine (conv_engine.hip.hpp), whose epilogue can emit them for a following sx conv.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace vitsmi {

typedef float sxf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kSxPeakSlots = 64;
constexpr float kF16Max = 65504.f;

// Timing mode (vits_set_timing): every conv launcher notes which kernel instantiation it started, spelled as rocprofv3
// prints it, so that the per-launch records of vits_launch_records() can be joined with a kernel trace by name.
inline thread_local bool g_launch_name_on = false;
inline thread_local char g_launch_name[128];

// ---- f16 mode (NP = 2): an fp32 operand as TWO fp16 planes, round-to-nearest at each step:
//     v ~ h0 + h1,   h0 = f16(v), h1 = f16(v - h0):   |v - h0 - h1| <= 2^-24 |v|  (11 + 11 bits and the sign of h1)
// so a product needs only the three MFMAs h0g0 + h0g1 + h1g0 (the dropped h1g1 is <= 2^-24 |v||g|): the same
// per-product error bound as one fp32 rounding, at half the matrix work of the six bf16 products.  fp16 has 5
// exponent bits, so the planes are kept in range explicitly:
//   weights      g = w * 2^k, k per tensor so that max |g| is in [2^14, 2^15); planes g0, g1 (packed two per block
//                row) and g0' = g0 * 2^-11, made in registers; 2^-k is applied to the accumulators (SxArgs::wscale)
//   activations  h0 = f16(x) (clamped to +-65504), h1' = f16((x - h0) * 2^11): the low plane is stored 2^11 up, which
//                keeps it a normal number wherever h0 is one, and meets g0' instead of g0 in its product:
//                x*g ~ h0*g0 + h0*g1 + h1'*g0'
// Activations therefore carry ~2^-23 relative error down to |x| = 2^-14 and an absolute floor of 2^-36 below that.
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    const sxf32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void split2h_pair(float x, float y, unsigned &w0, unsigned &w1) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
    w0 = cvt_pk_f16(x, y);
    const f16x2 h = __builtin_bit_cast(f16x2, w0);
    w1 = cvt_pk_f16((x - (float)h[0]) * 2048.f, (y - (float)h[1]) * 2048.f);
}
// ... the same, recording the largest magnitude seen (one v_max3_f32 per pair)
__device__ __forceinline__ void split2h_pair_pk(float x, float y, unsigned &w0, unsigned &w1, float &pk) {
    pk = __builtin_fmaxf(pk, __builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)));
    split2h_pair(x, y, w0, w1);
}
// ---- f16 single-plane storage (NP = 1, the reduced-precision vocoder of BASELINE config 4): an activation IS its fp16
// rounding (clamped to +-65504, range-tracked like the two-plane split); one MFMA product per fp32 product.
__device__ __forceinline__ unsigned cvt1h_pair_pk(float x, float y, float &pk) {
    pk = __builtin_fmaxf(pk, __builtin_fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)));
    return cvt_pk_f16(__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f));
}
// four stored values (two packed words = the 8 bytes a lane owns of a cell) back to fp32, undoing the leaky-ReLU the
// plane was stored with: x = p >= 0 ? p : p * unslope (unslope = 1 / slope; 1 = the plane holds x itself).  The map is
// exact up to the fp32 multiply, so a tensor that is both a conv input (activated) and a residual (not) is stored once.
__device__ __forceinline__ void unact4h(u32x2 w, float unslope, float (&o)[4]) {
    // (hipcc 7.2: __builtin_bit_cast of a vector ELEMENT expression - bit_cast(f16x2, w.y) - folds to element 0; the
    // elements go through scalars first)
    const unsigned wx = w.x, wy = w.y;
    const f16x2 a = __builtin_bit_cast(f16x2, wx), b = __builtin_bit_cast(f16x2, wy);
    const float v[4] = {(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = v[e] < 0.f ? v[e] * unslope : v[e];
}

// Publish a workgroup's peak (all 256 threads must call, uniformly): wave-wide max, the four wave maxima through LDS,
// then ONE atomicMax on one of the launch's 64 slots - each slot on its own 128-byte line, and skipped when the slot
// already holds as much (nearly always, after the first workgroups of a launch).  A first version with one atomic
// per WAVE on 64 adjacent words made the 10 us split kernel take 200 us: atomics on one cache line serialise in the L2.
constexpr int kSxPeakStride = 32;  // uints between slots: one 128-byte line each
// (sx_publish_peak's four floats are static LDS: the dynamic part a kernel may ask for is the CU's 160 KiB less that)
constexpr int kSxMaxDynLds = 160 * 1024 - 256;
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (function, device) pair: a process that opens handles on several
// GPUs must set it once per device, and the engine's launchers run on PipelinedSession's worker threads.  `done` is the
// instantiation's own bit mask of devices (device id modulo 64; a collision only repeats the idempotent call).
inline hipError_t sx_allow_big_lds(const void *kern, std::atomic<uint64_t> &done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kSxMaxDynLds);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}
// ... with the four floats in caller-provided LDS (a kernel whose dynamic LDS must be the whole of the CU's share: no
// static allocation beside it).  `s_pk` may alias memory other waves are still reading: a barrier comes first.
__device__ __forceinline__ void sx_publish_peak_at(unsigned *slots, int slot_idx, float pk, float *s_pk) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pk = __builtin_fmaxf(pk, __shfl_xor(pk, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_pk[threadIdx.x >> 6] = pk;
    __syncthreads();
    if (threadIdx.x == 0) {
        pk = __builtin_fmaxf(__builtin_fmaxf(s_pk[0], s_pk[1]), __builtin_fmaxf(s_pk[2], s_pk[3]));
        unsigned *slot = slots + (slot_idx & (kSxPeakSlots - 1)) * kSxPeakStride;
        const unsigned bits = __float_as_uint(pk);
        if (bits > __builtin_nontemporal_load(slot)) atomicMax(slot, bits);
    }
}
__device__ __forceinline__ void sx_publish_peak(unsigned *slots, int slot_idx, float pk) {
    __shared__ float s_pk[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pk = __builtin_fmaxf(pk, __shfl_xor(pk, o, 64));
    if ((threadIdx.x & 63) == 0) s_pk[threadIdx.x >> 6] = pk;
    __syncthreads();
    if (threadIdx.x == 0) {
        pk = __builtin_fmaxf(__builtin_fmaxf(s_pk[0], s_pk[1]), __builtin_fmaxf(s_pk[2], s_pk[3]));
        unsigned *slot = slots + (slot_idx & (kSxPeakSlots - 1)) * kSxPeakStride;
        const unsigned bits = __float_as_uint(pk);  // (non-negative floats order like their bit patterns; inf on top)
        if (bits > __builtin_nontemporal_load(slot)) atomicMax(slot, bits);
    }
}

// tanh / sigmoid without branches or library range handling (|error| < 1e-7 absolute, 2.5e-7 relative for tanh): the WN
// gate's epilogue applies 32 of them per lane next to a matrix loop that is ~40 % busy; the library tanhf alone is ~50
// instructions with branches.  v_exp_f32 saturates the way the functions need (exp2(+big) = inf -> 1 - 2/inf = 1).
__device__ __forceinline__ float tanh_nb(float v) {
    const float t = __builtin_fabsf(v), s = v * v;
    float p = 62.0f / 2835.0f;
    p = __builtin_fmaf(p, s, -17.0f / 315.0f);
    p = __builtin_fmaf(p, s, 2.0f / 15.0f);
    p = __builtin_fmaf(p, s, -1.0f / 3.0f);
    p = __builtin_fmaf(p * s, v, v);
    const float e = __builtin_amdgcn_exp2f(t * 2.8853900817779268f);  // exp(2 |v|)
    float r = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
    r = __builtin_copysignf(r, v);
    return t < 0.3f ? p : r;
}
__device__ __forceinline__ float sigmoid_nb(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

}  // namespace vitsmi
