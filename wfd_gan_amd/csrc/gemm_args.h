// Argument block and fused-store math shared by the contraction kernels (gemm.hip: fp32-operand MFMA; gemm_mx.hip:
// fp32 data staged as bf16 / split-bf16 operands).
#pragma once
#include "common.h"

struct GemmArgs {
    const float* A;
    const float* B;
    float* P;
    int M, N, K;
    int lda, ldb, ldp;
    long slab_stride;   // floats between split slabs
    int ktiles_per_split;
    int convL, convCin, convPad;  // IM2COL only
    int narrow;                   // N <= 32: the two wave columns split each k-tile instead of the N range
    // fused epilogue (splits == 1 only): 0 raw; 1 out = act(acc*scale[n]+shift[n]); 2 out = acc*f(D[m][n])*scale[n];
    // 4: the product is a weight gradient and P the parameter matrix itself: RMSprop in the store (P, sq updated in place,
    //    the gradient never reaches memory)
    int epi, act, dmode, ldd;
    const float* scale;
    const float* shift;
    const float* D;
    float* sq;
    float lr, alpha, eps;
    // conv8_c32 forward only: != 0 -> the launch also writes the k = 8 / stride 8 max-pool (window j = rows 8j - 3 .. 8j + 4
    // of a sample, zero padding taking part: MyMaxPool1dPadSame, src/model.py:107-131) of its activated output,
    // pool_out[(b * L/8 + j) * N + n]; needs L % 8 == 0 (the same window phase in every sample)
    float* pool_out;
};

static __device__ __forceinline__ float epi_apply(const GemmArgs& g, float v, int row, int col) {
    if (g.epi == 1) {
        const float sh = g.shift ? g.shift[col] : 0.f;
        v = g.scale ? __fmaf_rn(v, g.scale[col], sh) : v + sh;
        if (g.act == 1) v = fmaxf(v, 0.f);
        else if (g.act == 2) v = v > 0.f ? v : elu_expm1(v);
        else if (g.act == 3) v = 1.0f / (1.0f + expf(-v));
    } else if (g.epi == 2) {
        if (g.dmode) {
            const float d = g.D[(long)row * g.ldd + col];
            if (g.dmode == 1) v *= d;
            else if (g.dmode == 2) v = d > 0.f ? v : 0.f;
            else v = d > 0.f ? v : v * (d + 1.f);
        }
        if (g.scale) v *= g.scale[col];
    }
    return v;
}

// the store of a fused launch: element (row, col) of the product
static __device__ __forceinline__ void epi_store(const GemmArgs& g, float* __restrict__ P, float v, int row, int col) {
    const long idx = (long)row * g.ldp + col;
    if (g.epi == 4) {
        float p = P[idx], s = g.sq[idx];
        rmsprop_update(p, s, v, g.lr, g.alpha, g.eps);
        g.sq[idx] = s;
        P[idx] = p;
    } else {
        P[idx] = epi_apply(g, v, row, col);
    }
}
