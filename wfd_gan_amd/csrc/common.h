// Shared helpers for the wfd-gan MI355X (gfx950) kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WFD_OK 0
#define WFD_ERR_ARG 1
#define WFD_ERR_LAUNCH 2
#define WFD_ERR_WORKSPACE 3

#define WFD_CHECK_LAUNCH()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return WFD_ERR_LAUNCH; \
    } while (0)

// Raise a kernel's dynamic-LDS limit once PER DEVICE (a `static bool` would serve only the first GPU a process uses; the
// bitmask is also safe against concurrent host threads).  `fn` is the kernel in parentheses: WFD_LDS_ATTR((k<A, B>), bytes).
#include <atomic>
#define WFD_LDS_ATTR(fn, bytes)                                                                                          \
    do {                                                                                                                  \
        static std::atomic<unsigned long long> done__{0};                                                                 \
        int dev__ = 0;                                                                                                    \
        (void)hipGetDevice(&dev__);                                                                                       \
        const unsigned long long bit__ = 1ull << (dev__ & 63);                                                            \
        if (!(done__.load(std::memory_order_acquire) & bit__)) {                                                          \
            (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));         \
            done__.fetch_or(bit__, std::memory_order_release);                                                            \
        }                                                                                                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// exp(v) - 1 for v <= 0 (ELU's negative branch) in ~12 instructions: a degree-6 Taylor polynomial above -0.25 (relative
// error < 2e-8), v_exp-based __expf(v) - 1 below it (the result is then <= -0.22, so the cancellation costs < 1e-6
// relative).  OCML's expm1f is ~4x the instructions, which showed as VALU time in the channels-last conv epilogues.
__device__ __forceinline__ float elu_expm1(float v) {
    const float p = v * (1.f + v * (0.5f + v * (1.f / 6 + v * (1.f / 24 + v * (1.f / 120 + v * (1.f / 720))))));
    return v > -0.25f ? p : __expf(v) - 1.f;
}

// torch.optim.RMSprop defaults (no momentum, not centered, weight decay 0; src/mlp_df_wgan_train.py:163-164): one element.
// Contraction is switched off inside (every operation rounded on its own, as torch's CPU kernels do), so that every kernel
// that applies the update -- the flat optimizer pass and the weight-gradient stores that fold it in -- produces the same bits
// from the same (p, sq, g) whatever the surrounding code looks like.
__device__ __forceinline__ void rmsprop_update(float& p, float& s, float g, float lr, float alpha, float eps) {
#pragma clang fp contract(off)
    const float t = (1.f - alpha) * g * g;
    s = s * alpha + t;
#ifdef WFD_RMSPROP_IEEE
    p = p - lr * g / (sqrtf(s) + eps);           // correctly rounded sqrt and division: ~40 VALU instructions per element
#else
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sequences: the update then differs from torch's
    // by a few ulp of the STEP (~1e-7 of lr * g / sqrt(s)), far inside the 1e-3 parity bar and the 1e-4 the optimizer tests
    // state; it takes the update from ~40 to ~12 VALU instructions per element, which is what the weight-gradient launch's
    // store phase was spending (knock-out: 4.6 of 43.8 us per critic launch).  Every
    // kernel that applies RMSprop shares this function, so the fused and the flat forms stay bit-identical to each other.
    p = p - (lr * g) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(s) + eps);
#endif
}

// 16-byte non-temporal accesses: data that is touched once per step (square averages, gradients on their way into the
// optimizer, the sampler's final bursts) streams past the caches instead of evicting what the next launches read
__device__ __forceinline__ float4 nt_load4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, float4 v) {
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x multiple of 64, <= 1024; scratch >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
