// bf16-MFMA contraction core over fp32 data (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for gfx950.
//
// Same products, layouts, split-K slabs and fused stores as gemm.hip (NT / NN / TN / implicit-GEMM conv over fp32
// matrices in HBM), but the operands are converted on their way into LDS and multiplied on the bf16 matrix pipe,
// whose rate is 16x that of the f32-input MFMA on this part:
//   TERMS = 1  "bf16":   x -> bf16(x); one MFMA per 16-deep k-step.  bf16-operand / fp32-accumulate numerics
//                        (~4e-3 per operand): the large-batch characterisation mode of BASELINE config 4.
//   TERMS = 2  "bf16x3": x -> hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in f32); the product is
//                        hi*hi + hi*lo + lo*hi, three MFMAs per k-step; the dropped lo*lo term and the residual of
//                        the split are each <= 2^-17 relative, so products carry ~1e-5 relative error (f32-input
//                        MFMA: 6e-8) at 16/3 of its rate.
// Nothing in HBM changes: activations, weights, slabs and every epilogue stay fp32, so a step plan runs on either
// core unmodified (wfd_set_gemm_mode).
//
// Tile 64x64x64 per 256-thread workgroup, 2x2 waves of one 32x32 accumulator.  Staging is global -> registers ->
// convert -> LDS, double buffered, one barrier per k-tile.  LDS images:
//   k-contiguous operand ([rows][K] in HBM):  [row][72] bf16 (144-B rows: nine 16-B slots, so the ds_read_b128
//       fragment reads of 32 consecutive rows are conflict-free); lane (i, h) of k-step s reads k = 16 s + 8 h .. + 7.
//   m-contiguous operand ([K][rows] in HBM):  [k][64] bf16 as loaded; the fragment (8 consecutive k of one row per
//       lane) comes from two ds_read_b64_tr_b16 (hardware transpose: a 16-lane group reads a 4 x 16 block and lane c
//       receives column c), so no transposing store and no second image is needed.
#include "gemm_args.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int XM = 64, XN = 64, XK = 64, XT = 256;
constexpr int RSK = XK + 8;          // row stride (elements) of a k-contiguous image
constexpr int RSM = 64;              // row stride of an m-contiguous image [k][64]
constexpr int IMG = 64 * RSK;        // elements reserved per image (>= 64 * RSM)

// ---- global -> registers: 4 float4 per thread per operand tile ---------------------------------------------------
template <bool KC, bool IM2COL>
__device__ __forceinline__ void mx_load(float4 (&r)[4], const float* __restrict__ G, int ld, int row0, int nrows, int k0,
                                        int kend, int tid, int convL, int convCin, int convPad) {
    const int q = tid & 15, w = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (KC) {
            const int gr = row0 + w + 16 * j, gk = k0 + 4 * q;
            bool ok = gr < nrows && gk < kend;
            long off = (long)gr * ld + gk;
            if constexpr (IM2COL) {
                const int l = gr % convL, t = gk / convCin;
                ok = ok && (unsigned)(l + t - convPad) < (unsigned)convL;
                off = (long)gr * convCin + gk - (long)convPad * convCin;
            }
            if (ok) {
                if (gk + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(G + off);
                } else {
                    v.x = G[off];
                    if (gk + 1 < kend) v.y = G[off + 1];
                    if (gk + 2 < kend) v.z = G[off + 2];
                }
            }
        } else {
            const int gk = k0 + w + 16 * j, gc = row0 + 4 * q;
            if (gk < kend && gc < nrows) {
                const long off = (long)gk * ld + gc;
                if (gc + 3 < nrows) {
                    v = *reinterpret_cast<const float4*>(G + off);
                } else {
                    v.x = G[off];
                    if (gc + 1 < nrows) v.y = G[off + 1];
                    if (gc + 2 < nrows) v.z = G[off + 2];
                }
            }
        }
        r[j] = v;
    }
}

// ---- registers -> bf16 (hi [, lo]) -> LDS --------------------------------------------------------------------------
template <bool KC, int TERMS>
__device__ __forceinline__ void mx_store(const float4 (&r)[4], __bf16* __restrict__ S, int tid) {
    const int q = tid & 15, w = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 x = {r[j].x, r[j].y, r[j].z, r[j].w};
        const bf16x4 hi = __builtin_convertvector(x, bf16x4);
        const int off = KC ? (w + 16 * j) * RSK + 4 * q : (w + 16 * j) * RSM + 4 * q;
        *reinterpret_cast<bf16x4*>(S + off) = hi;
        if constexpr (TERMS == 2) {
            const f32x4 res = x - __builtin_convertvector(hi, f32x4);
            *reinterpret_cast<bf16x4*>(S + IMG + off) = __builtin_convertvector(res, bf16x4);
        }
    }
}

// fragment of k-step s (16 deep) for the 32 rows starting at r0: lane (i = lane & 31, h = lane >> 5) gets
// k = 16 s + 8 h + 0..7 of row r0 + i
template <bool KC>
__device__ __forceinline__ bf16x8 mx_frag(const __bf16* __restrict__ S, int r0, int s, int lane) {
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(S + (r0 + (lane & 31)) * RSK + 16 * s + 8 * (lane >> 5));
    } else {
        // 16-lane group g2 covers rows r0 + 16 g2 .. + 15; lane 4 qq + p of the group addresses k-row qq, columns 4 p .. 4 p + 3
        const int li = lane & 15, qq = li >> 2, p = li & 3, g2 = (lane >> 4) & 1, h = lane >> 5;
        const __bf16* base = S + (16 * s + 8 * h + qq) * RSM + r0 + 16 * g2 + 4 * p;
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * RSM));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        return u.v;
    }
}

template <bool AKC, bool BKC, bool IM2COL, int TERMS, bool EPI>
__global__ __launch_bounds__(XT) void gemm_mx_kernel(GemmArgs g, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    // buffer c: [A hi | A lo | B hi | B lo] images (lo only with TERMS = 2)
    constexpr int BUF = 2 * TERMS * IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * XM, n0 = blockIdx.x * XN, z = blockIdx.z;
    const int kbeg = z * kchunk;
    const int kend = min(g.K, kbeg + kchunk);
    const int nk = (kend > kbeg) ? (kend - kbeg + XK - 1) / XK : 0;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float4 ra[4], rb[4];
    if (nk > 0) {
        mx_load<AKC, IM2COL>(ra, g.A, g.lda, m0, g.M, kbeg, kend, tid, g.convL, g.convCin, g.convPad);
        mx_load<BKC, false>(rb, g.B, g.ldb, n0, g.N, kbeg, kend, tid, 0, 0, 0);
        mx_store<AKC, TERMS>(ra, smem, tid);
        mx_store<BKC, TERMS>(rb, smem + TERMS * IMG, tid);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            const int k0 = kbeg + (kt + 1) * XK;
            mx_load<AKC, IM2COL>(ra, g.A, g.lda, m0, g.M, k0, kend, tid, g.convL, g.convCin, g.convPad);
            mx_load<BKC, false>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid, 0, 0, 0);
        }
        const __bf16* a_s = smem + cur * BUF;
        const __bf16* b_s = a_s + TERMS * IMG;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ah = mx_frag<AKC>(a_s, wm * 32, s, lane);
            const bf16x8 bh = mx_frag<BKC>(b_s, wn * 32, s, lane);
            if constexpr (TERMS == 2) {
                const bf16x8 al = mx_frag<AKC>(a_s + IMG, wm * 32, s, lane);
                const bf16x8 bl = mx_frag<BKC>(b_s + IMG, wn * 32, s, lane);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        if (more) {
            mx_store<AKC, TERMS>(ra, smem + (cur ^ 1) * BUF, tid);
            mx_store<BKC, TERMS>(rb, smem + (cur ^ 1) * BUF + TERMS * IMG, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // C/D map of the 32x32 accumulator: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* P = g.P + (long)z * g.slab_stride;
    const int col = n0 + wn * 32 + i;
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < g.M) {
                if constexpr (EPI) P[(long)row * g.ldp + col] = epi_apply(g, acc[r], row, col);
                else P[(long)row * g.ldp + col] = acc[r];
            }
        }
    }
}

template <bool AKC, bool BKC, bool IM2COL, int TERMS, bool EPI>
int mx_launch3(const GemmArgs& g, int splits, hipStream_t st) {
    static bool attr = false;
    const size_t lds = (size_t)2 * 2 * TERMS * IMG * sizeof(__bf16);
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_mx_kernel<AKC, BKC, IM2COL, TERMS, EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int kchunk = g.ktiles_per_split * 32;       // ktiles_per_split counts 32-deep tiles (gemm.hip's BK)
    dim3 grid(cdiv(g.N, XN), cdiv(g.M, XM), splits), block(XT);
    hipLaunchKernelGGL((gemm_mx_kernel<AKC, BKC, IM2COL, TERMS, EPI>), grid, block, lds, st, g, kchunk);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

template <bool AKC, bool BKC, bool IM2COL>
int mx_launch2(const GemmArgs& g, int splits, int terms, hipStream_t st) {
    if (terms == 1) {
        if (g.epi) return mx_launch3<AKC, BKC, IM2COL, 1, true>(g, splits, st);
        return mx_launch3<AKC, BKC, IM2COL, 1, false>(g, splits, st);
    }
    if (g.epi) return mx_launch3<AKC, BKC, IM2COL, 2, true>(g, splits, st);
    return mx_launch3<AKC, BKC, IM2COL, 2, false>(g, splits, st);
}

}  // namespace

// layout: 0 NT, 1 NN, 2 TN, 3 NT with the A operand read as the im2col matrix of a k=8 conv.  terms: 1 bf16, 2 bf16x3.
// Requires 16-byte aligned operands with leading dimensions that are multiples of 4 floats (the caller falls back to
// the f32-operand core otherwise).
int wfd_mx_launch(const GemmArgs& g, int layout, int splits, int terms, hipStream_t st) {
    switch (layout) {
        case 0: return mx_launch2<true, true, false>(g, splits, terms, st);
        case 1: return mx_launch2<true, false, false>(g, splits, terms, st);
        case 2: return mx_launch2<false, false, false>(g, splits, terms, st);
        case 3: return mx_launch2<true, true, true>(g, splits, terms, st);
        default: return WFD_ERR_ARG;
    }
}
