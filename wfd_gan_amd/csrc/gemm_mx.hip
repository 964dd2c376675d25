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
// Tiles: 64x64x64 per 256-thread workgroup (2x2 waves of one 32x32 accumulator; the batch-64 step, with split-K), or
// 128x128 (2x2 waves of 2x2 accumulators) once such tiles alone fill the chip.  Staging is global -> registers ->
// convert -> LDS, double buffered, one barrier per k-tile.  LDS images:
//   k-contiguous operand ([rows][K] in HBM):  [row][KT + 8] bf16 (an odd number of 16-B slots per row, so the
//       ds_read_b128 fragment reads of 32 consecutive rows are conflict-free); lane (i, h) of k-step s reads
//       k = 16 s + 8 h .. + 7.
//   m-contiguous operand ([K][rows] in HBM):  [k][T (+16)] bf16 as loaded; the fragment (8 consecutive k of one row per
//       lane) comes from two ds_read_b64_tr_b16 (hardware transpose: a 16-lane group reads a 4 x 16 block and lane c
//       receives column c), so no transposing store and no second image is needed.
#include "gemm_args.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int XT = 256;

// Tile geometry: T x T outputs per 256-thread workgroup (T = 64: 2x2 waves of one 32x32 accumulator; T = 128: 2x2
// waves of 2x2 accumulators -- every fragment read feeds two MFMAs per term), KT-deep k-tiles.
template <int T, int KT>
struct Geo {
    static constexpr int RSK = KT + 8;        // row stride (elements) of a k-contiguous image [T][KT + 8]
    static constexpr int RSM = T == 64 ? 64 : T + 16;   // row stride of an m-contiguous image [KT][RSM]; the 4 x 16 blocks of
                                              // a transposed read are 2-way conflicted at these strides (a perfect tiling
                                              // needs T + 32, which costs a workgroup per CU in LDS: measured slower)
    static constexpr int IMG_K = T * RSK, IMG_M = KT * RSM;
    static constexpr int NLD = T * KT / 4 / XT;   // float4 per thread per operand tile
};

// ---- global -> registers ---------------------------------------------------------------------------------------
template <int T, int KT, bool KC, bool IM2COL, int NLD>
__device__ __forceinline__ void mx_load(float4 (&r)[NLD], const float* __restrict__ G, int ld, int row0,
                                        int nrows, int k0, int kend, int tid, int convL, int convCin, int convPad) {
    constexpr int QK = KT / 4, QM = T / 4;
    static_assert(NLD == Geo<T, KT>::NLD, "staging registers per thread");
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (KC) {
            const int q = tid % QK, w = tid / QK;
            const int gr = row0 + w + (XT / QK) * j, gk = k0 + 4 * q;
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
            const int q = tid % QM, w = tid / QM;
            const int gk = k0 + w + (XT / QM) * j, gc = row0 + 4 * q;
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

// Interior k-tiles (the whole tile inside the split's k-range) take a branch-free form: per-thread element offsets are
// computed once before the k-loop with out-of-range rows / columns CLAMPED to valid ones -- what they load only feeds
// output rows / columns that are never stored -- so a tile is NLD unconditional 16-byte loads at offset + constant.
// (Leading dimensions are multiples of 4 floats, so a 16-byte load that starts inside a row stays inside its
// allocation; offsets are 32-bit element counts -- the entry points reject operands of 2^31 elements or more.)  The implicit-GEMM operand additionally zeroes taps that fall outside the sample.
template <int T, int KT, bool KC, bool IM2COL, int NLD>
__device__ __forceinline__ void mx_offsets(int (&off)[NLD], int (&lpos)[NLD], int ld, int row0, int nrows, int kbeg, int tid,
                                           int convL, int convCin, int convPad) {
    constexpr int QK = KT / 4, QM = T / 4;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        if constexpr (KC) {
            const int gr = min(row0 + tid / QK + (XT / QK) * j, nrows - 1), gk = kbeg + 4 * (tid % QK);
            if constexpr (IM2COL) {
                off[j] = gr * convCin + gk - convPad * convCin;
                lpos[j] = gr % convL - convPad;
            } else {
                off[j] = gr * ld + gk;
                lpos[j] = 0;
            }
        } else {
            int gc = row0 + 4 * (tid % QM);
            if (gc >= nrows) gc = 0;
            off[j] = (kbeg + tid / QM + (XT / QM) * j) * ld + gc;
            lpos[j] = 0;
        }
    }
}

template <int T, int KT, bool KC, bool IM2COL, int NLD>
__device__ __forceinline__ void mx_load_fast(float4 (&r)[NLD], const float* __restrict__ G, const int (&off)[NLD],
                                             const int (&lpos)[NLD], int step, int k0, int tid, int convL, int convCin) {
    if constexpr (IM2COL) {
        const int t = (k0 + 4 * (tid % (KT / 4))) / convCin;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const bool ok = (unsigned)(lpos[j] + t) < (unsigned)convL;
            const float4 v = *reinterpret_cast<const float4*>(G + (ok ? off[j] + step : 0));
            r[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NLD; ++j) r[j] = *reinterpret_cast<const float4*>(G + off[j] + step);
    }
}

// ---- registers -> bf16 (hi [, lo]) -> LDS --------------------------------------------------------------------------
template <int T, int KT, bool KC, int TERMS, int NLD>
__device__ __forceinline__ void mx_store(const float4 (&r)[NLD], __bf16* __restrict__ S, int tid) {
    using G_ = Geo<T, KT>;
    constexpr int QK = KT / 4, QM = T / 4;
    constexpr int IMG = KC ? G_::IMG_K : G_::IMG_M;
#pragma unroll
    for (int j = 0; j < G_::NLD; ++j) {
        const f32x4 x = {r[j].x, r[j].y, r[j].z, r[j].w};
        const bf16x4 hi = __builtin_convertvector(x, bf16x4);
        const int off = KC ? (tid / QK + (XT / QK) * j) * G_::RSK + 4 * (tid % QK)
                           : (tid / QM + (XT / QM) * j) * G_::RSM + 4 * (tid % QM);
        *reinterpret_cast<bf16x4*>(S + off) = hi;
        if constexpr (TERMS == 2) {
            const f32x4 res = x - __builtin_convertvector(hi, f32x4);
            *reinterpret_cast<bf16x4*>(S + IMG + off) = __builtin_convertvector(res, bf16x4);
        }
    }
}

// fragment of k-step s (16 deep) for the 32 rows starting at r0: lane (i = lane & 31, h = lane >> 5) gets
// k = 16 s + 8 h + 0..7 of row r0 + i
template <int T, int KT, bool KC>
__device__ __forceinline__ bf16x8 mx_frag(const __bf16* __restrict__ S, int r0, int s, int lane) {
    using G_ = Geo<T, KT>;
    if constexpr (KC) {
        return *reinterpret_cast<const bf16x8*>(S + (r0 + (lane & 31)) * G_::RSK + 16 * s + 8 * (lane >> 5));
    } else {
        // 16-lane group g2 covers rows r0 + 16 g2 .. + 15; lane 4 qq + p of the group addresses k-row qq, columns 4 p .. 4 p + 3
        const int li = lane & 15, qq = li >> 2, p = li & 3, g2 = (lane >> 4) & 1, h = lane >> 5;
        const __bf16* base = S + (16 * s + 8 * h + qq) * G_::RSM + r0 + 16 * g2 + 4 * p;
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * G_::RSM));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = lo; u.s.b = hi;
        return u.v;
    }
}

template <int T, int KT, bool AKC, bool BKC, bool IM2COL, int TERMS, bool EPI>
__global__ __launch_bounds__(XT) void gemm_mx_kernel(GemmArgs g, int kchunk, int gx, int gy) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    using G_ = Geo<T, KT>;
    constexpr int IMG_A = AKC ? G_::IMG_K : G_::IMG_M, IMG_B = BKC ? G_::IMG_K : G_::IMG_M;
    // buffer c: [A hi | A lo | B hi | B lo] images (lo only with TERMS = 2)
    constexpr int BUF = TERMS * (IMG_A + IMG_B);
    constexpr int FR = T / 64;                    // 32x32 accumulators per wave along each dimension
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    // tile order: ids b and b + 8 share an XCD (round-robin dispatch); give each XCD a contiguous run of tiles so that
    // the operand panels its workgroups share stay in its L2 (speed only)
    int bid = blockIdx.x;
    if (gx * gy >= 256) {         // small grids keep the dispatch order: the two row tiles of a weight panel share an XCD
        const int nb = gx * gy, xcd = bid & 7, qd = nb >> 3, rm = nb & 7;
        bid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    }
    const int m0 = (bid / gx) * T, n0 = (bid % gx) * T, z = blockIdx.z;
    const int kbeg = z * kchunk;
    const int kend = min(g.K, kbeg + kchunk);
    const int nk = (kend > kbeg) ? (kend - kbeg + KT - 1) / KT : 0;

    f32x16 acc[FR][FR];
#pragma unroll
    for (int a = 0; a < FR; ++a)
#pragma unroll
        for (int b = 0; b < FR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // One register set, one tile ahead: the global loads of tile kt + 1 are issued before tile kt is multiplied out of
    // LDS buffer kt & 1 and go to the other buffer at the end of the step.  (A second set with the loads two tiles ahead
    // costs a workgroup per CU in registers: measured 530 -> 323 TF at M = 8192.)
    float4 ra0[G_::NLD], rb0[G_::NLD];
    int offa[G_::NLD], offb[G_::NLD];
    int lpa[G_::NLD], lpb[G_::NLD];
    mx_offsets<T, KT, AKC, IM2COL>(offa, lpa, g.lda, m0, g.M, kbeg, tid, g.convL, g.convCin, g.convPad);
    mx_offsets<T, KT, BKC, false>(offb, lpb, g.ldb, n0, g.N, kbeg, tid, 0, 0, 0);
    const int stepa = AKC ? KT : KT * g.lda, stepb = BKC ? KT : KT * g.ldb;
    auto load_tile = [&](int kt, float4 (&ra)[G_::NLD], float4 (&rb)[G_::NLD]) {
        const int k0 = kbeg + kt * KT;
        if (k0 + KT <= kend) {          // wave-uniform: interior tile
            mx_load_fast<T, KT, AKC, IM2COL>(ra, g.A, offa, lpa, kt * stepa, k0, tid, g.convL, g.convCin);
            mx_load_fast<T, KT, BKC, false>(rb, g.B, offb, lpb, kt * stepb, k0, tid, 0, 0);
        } else {                        // ragged last tile of the split
            mx_load<T, KT, AKC, IM2COL>(ra, g.A, g.lda, m0, g.M, k0, kend, tid, g.convL, g.convCin, g.convPad);
            mx_load<T, KT, BKC, false>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf, const float4 (&ra)[G_::NLD], const float4 (&rb)[G_::NLD]) {
        mx_store<T, KT, AKC, TERMS>(ra, smem + buf * BUF, tid);
        mx_store<T, KT, BKC, TERMS>(rb, smem + buf * BUF + TERMS * IMG_A, tid);
    };
    auto compute = [&](int buf) {
        const __bf16* a_s = smem + buf * BUF;
        const __bf16* b_s = a_s + TERMS * IMG_A;
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {
            bf16x8 ah[FR], bh[FR], al[FR], bl[FR];
#pragma unroll
            for (int t = 0; t < FR; ++t) {
                ah[t] = mx_frag<T, KT, AKC>(a_s, wm * (T / 2) + 32 * t, s, lane);
                bh[t] = mx_frag<T, KT, BKC>(b_s, wn * (T / 2) + 32 * t, s, lane);
                if constexpr (TERMS == 2) {
                    al[t] = mx_frag<T, KT, AKC>(a_s + IMG_A, wm * (T / 2) + 32 * t, s, lane);
                    bl[t] = mx_frag<T, KT, BKC>(b_s + IMG_B, wn * (T / 2) + 32 * t, s, lane);
                }
            }
#pragma unroll
            for (int a = 0; a < FR; ++a)
#pragma unroll
                for (int b = 0; b < FR; ++b) {
                    if constexpr (TERMS == 2) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
    };
    if (nk > 0) {
        load_tile(0, ra0, rb0);
        store_tile(0, ra0, rb0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1, ra0, rb0);
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1, ra0, rb0);
        __syncthreads();
    }

    // C/D map of a 32x32 accumulator: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* P = g.P + (long)z * g.slab_stride;
#pragma unroll
    for (int a = 0; a < FR; ++a)
#pragma unroll
        for (int b = 0; b < FR; ++b) {
            const int col = n0 + wn * (T / 2) + 32 * b + i;
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (T / 2) + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.M) continue;
                if constexpr (EPI) epi_store(g, P, acc[a][b][r], row, col);
                else P[(long)row * g.ldp + col] = acc[a][b][r];
            }
        }
}

template <int T, int KT, bool AKC, bool BKC, bool IM2COL, int TERMS, bool EPI>
int mx_launch4(const GemmArgs& g, int splits, hipStream_t st) {
    using G_ = Geo<T, KT>;
    const size_t lds = (size_t)2 * TERMS * ((AKC ? G_::IMG_K : G_::IMG_M) + (BKC ? G_::IMG_K : G_::IMG_M)) * sizeof(__bf16);
    WFD_LDS_ATTR((gemm_mx_kernel<T, KT, AKC, BKC, IM2COL, TERMS, EPI>), lds);
    const int kchunk = g.ktiles_per_split * 32;       // ktiles_per_split counts 32-deep tiles (gemm.hip's BK)
    const int gx = cdiv(g.N, T), gy = cdiv(g.M, T);
    dim3 grid(gx * gy, 1, splits), block(XT);
    hipLaunchKernelGGL((gemm_mx_kernel<T, KT, AKC, BKC, IM2COL, TERMS, EPI>), grid, block, lds, st, g, kchunk, gx, gy);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// k=8 "SAME" conv over a long channels-last sequence with few channels (the DF observer's first two blocks: Cin, Cout in
// {32, 64}, M = B*L rows; a workgroup owns COUT of the output columns): on the bf16 pipe the arithmetic is negligible and the implicit-GEMM form above is bound by
// re-reading every activation row eight times (once per tap).  Here a workgroup keeps the whole weight matrix
// [Cout][8*Cin] in LDS as bf16 for its lifetime and walks `cpw` consecutive chunks of 128 output rows: a chunk's
// 128 + 7 input rows are loaded once (contiguous, register-prefetched one chunk ahead), converted, and the A fragment of
// tap t is the same LDS tile read t rows further down.  HBM traffic = input + output once: an HBM-bound kernel.
// Wave w owns rows 32 w .. 32 w + 31 of the chunk and all Cout columns.  Sample edges are masked per lane.
constexpr int CV_ROWS = 128;

template <int CIN, int COUT, int TERMS>
struct ConvGeo {
    static constexpr int WS = 8 * CIN + 8, XS = CIN + 8;          // bf16 row strides: odd numbers of 16-B slots
    static constexpr int W_IMG = COUT * WS, X_IMG = (CV_ROWS + 7) * XS;
    static constexpr int XLD = ((CV_ROWS + 7) * CIN / 4 + XT - 1) / XT;   // float4 per thread per chunk
    static constexpr int IMG_BYTES = TERMS * (W_IMG + X_IMG) * (int)sizeof(__bf16);   // multiple of 16
    static constexpr size_t LDS = (size_t)IMG_BYTES + (size_t)CV_ROWS * COUT * sizeof(float);   // + output staging tile
};

template <int TERMS>
__device__ __forceinline__ void cv_put(__bf16* __restrict__ S, int img, int off, const float4& v) {
    const f32x4 x = {v.x, v.y, v.z, v.w};
    const bf16x4 hi = __builtin_convertvector(x, bf16x4);
    *reinterpret_cast<bf16x4*>(S + off) = hi;
    if constexpr (TERMS == 2) {
        const f32x4 res = x - __builtin_convertvector(hi, f32x4);
        *reinterpret_cast<bf16x4*>(S + img + off) = __builtin_convertvector(res, bf16x4);
    }
}

template <int CIN, int COUT, int TERMS, bool EPI>
__global__ __launch_bounds__(XT) void conv8_lds_mx_kernel(GemmArgs g, int cpw, int nchunks, int wide) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    using C_ = ConvGeo<CIN, COUT, TERMS>;
    __bf16* ws = smem;                            // [TERMS][COUT][WS]
    __bf16* xs = smem + TERMS * C_::W_IMG;        // [TERMS][135][XS]
    constexpr int Q = CIN / 4, NA = COUT / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    const int pad = g.convPad, L = g.convL;
    const int c0 = blockIdx.x * cpw, c1 = min(nchunks, c0 + cpw);
    const int n0 = blockIdx.y * COUT;             // COUT = output columns of this workgroup (a slice of g.N)

    for (int c = tid; c < COUT * 2 * CIN; c += XT) {           // weights: COUT rows x 8*CIN floats, float4 granules
        const int r = c / (2 * CIN), q = c % (2 * CIN);
        cv_put<TERMS>(ws, C_::W_IMG, r * C_::WS + 4 * q,
                      *reinterpret_cast<const float4*>(g.B + (long)(n0 + r) * (8 * CIN) + 4 * q));
    }
    float4 rx[C_::XLD];
    auto load_x = [&](int chunk) {
        const long base = (long)chunk * CV_ROWS - pad;
#pragma unroll
        for (int j = 0; j < C_::XLD; ++j) {
            const int c = tid + XT * j, r = c / Q, q = c % Q;
            const long src = base + r;
            rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < (CV_ROWS + 7) * Q && src >= 0 && src < g.M) rx[j] = *reinterpret_cast<const float4*>(g.A + src * CIN + 4 * q);
        }
    };
    if (c0 < c1) load_x(c0);
    for (int chunk = c0; chunk < c1; ++chunk) {
        __syncthreads();                           // the previous chunk's fragment reads are done (and the weights are in)
#pragma unroll
        for (int j = 0; j < C_::XLD; ++j) {
            const int c = tid + XT * j, r = c / Q, q = c % Q;
            if (c < (CV_ROWS + 7) * Q) cv_put<TERMS>(xs, C_::X_IMG, r * C_::XS + 4 * q, rx[j]);
        }
        __syncthreads();
        if (chunk + 1 < c1) load_x(chunk + 1);

        const int m0 = chunk * CV_ROWS;
        const int mrow = m0 + wave * 32 + i;       // the output row whose A values this lane feeds
        const int l = mrow < g.M ? mrow % L : -(1 << 20);
        f32x16 acc[NA];
#pragma unroll
        for (int b = 0; b < NA; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool ok = (unsigned)(l + t - pad) < (unsigned)L;
            const __bf16* ap = xs + (wave * 32 + i + t) * C_::XS + 8 * h;
            const __bf16* bp = ws + i * C_::WS + t * CIN + 8 * h;
#pragma unroll
            for (int s = 0; s < CIN / 16; ++s) {
                bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 16 * s), al;
                if constexpr (TERMS == 2) al = *reinterpret_cast<const bf16x8*>(ap + C_::X_IMG + 16 * s);
                if (!ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)0.f; if constexpr (TERMS == 2) al[e] = (__bf16)0.f; }
                }
#pragma unroll
                for (int b = 0; b < NA; ++b) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp + 32 * b * C_::WS + 16 * s);
                    if constexpr (TERMS == 2) {
                        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + C_::W_IMG + 32 * b * C_::WS + 16 * s);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[b], 0, 0, 0);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[b], 0, 0, 0);
                    }
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[b], 0, 0, 0);
                }
            }
        }
        if (wide) {
            // The accumulator layout gives a lane 16 rows of one column: stored directly that is 16 dword stores per
            // lane, two 128-byte row pieces per instruction (and the same for the D operand of the backward form).  The
            // tile goes through LDS instead and leaves as 16-byte accesses along the rows: this kernel is HBM-bound.
            float* os = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + C_::IMG_BYTES);     // [128][COUT]
#pragma unroll
            for (int b = 0; b < NA; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    os[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * COUT + 32 * b + i] = acc[b][r];
            __syncthreads();
            constexpr int Q4 = COUT / 4;
            for (int e = tid; e < CV_ROWS * Q4; e += XT) {
                const int rl = e / Q4, c4 = 4 * (e % Q4), row = m0 + rl, col = n0 + c4;
                if (row >= g.M) break;
                float4 v = *reinterpret_cast<const float4*>(os + rl * COUT + c4);
                if constexpr (EPI) {
                    float o[4] = {v.x, v.y, v.z, v.w};
                    if (g.epi == 1) {
                        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (g.scale) sc = *reinterpret_cast<const float4*>(g.scale + col);
                        if (g.shift) sh = *reinterpret_cast<const float4*>(g.shift + col);
                        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float x = g.scale ? __fmaf_rn(o[k], scv[k], shv[k]) : o[k] + shv[k];
                            if (g.act == 1) x = fmaxf(x, 0.f);
                            else if (g.act == 2) x = x > 0.f ? x : elu_expm1(x);
                            else if (g.act == 3) x = 1.0f / (1.0f + expf(-x));
                            o[k] = x;
                        }
                    } else if (g.epi == 2) {
                        if (g.dmode) {
                            const float4 d4 = *reinterpret_cast<const float4*>(g.D + (long)row * g.ldd + col);
                            const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if (g.dmode == 1) o[k] *= dv[k];
                                else if (g.dmode == 2) o[k] = dv[k] > 0.f ? o[k] : 0.f;
                                else o[k] = dv[k] > 0.f ? o[k] : o[k] * (dv[k] + 1.f);
                            }
                        }
                        if (g.scale) {
                            const float4 sc = *reinterpret_cast<const float4*>(g.scale + col);
                            o[0] *= sc.x; o[1] *= sc.y; o[2] *= sc.z; o[3] *= sc.w;
                        }
                    }
                    v = make_float4(o[0], o[1], o[2], o[3]);
                }
                *reinterpret_cast<float4*>(g.P + (long)row * g.ldp + col) = v;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NA; ++b) {
                const int col = n0 + 32 * b + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < g.M) {
                        if constexpr (EPI) epi_store(g, g.P, acc[b][r], row, col);
                        else g.P[(long)row * g.ldp + col] = acc[b][r];
                    }
                }
            }
        }
    }
}

template <int CIN, int COUT, int TERMS, bool EPI>
int cv_launch2(const GemmArgs& g, hipStream_t st) {
    using C_ = ConvGeo<CIN, COUT, TERMS>;
    WFD_LDS_ATTR((conv8_lds_mx_kernel<CIN, COUT, TERMS, EPI>), C_::LDS);
    const int nchunks = cdiv(g.M, CV_ROWS);
    int cpw = nchunks / 1024;                     // amortise the weight staging once the chip is covered four times over
    if (cpw < 1) cpw = 1;
    if (cpw > 8) cpw = 8;
    // the 16-byte epilogue needs 16-byte aligned rows in everything it touches
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int wide = a16(g.P) && g.ldp % 4 == 0 && (!g.scale || a16(g.scale)) && (!g.shift || a16(g.shift)) &&
                     (!(g.epi == 2 && g.dmode) || (a16(g.D) && g.ldd % 4 == 0));
    hipLaunchKernelGGL((conv8_lds_mx_kernel<CIN, COUT, TERMS, EPI>), dim3(cdiv(nchunks, cpw), g.N / COUT), dim3(XT), C_::LDS,
                       st, g, cpw, nchunks, wide);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

template <int CIN, int COUT>
int cv_launch(const GemmArgs& g, int terms, hipStream_t st) {
    if (terms == 1) return g.epi ? cv_launch2<CIN, COUT, 1, true>(g, st) : cv_launch2<CIN, COUT, 1, false>(g, st);
    return g.epi ? cv_launch2<CIN, COUT, 2, true>(g, st) : cv_launch2<CIN, COUT, 2, false>(g, st);
}

template <bool AKC, bool BKC, bool IM2COL, int TERMS, bool EPI>
int mx_launch3(const GemmArgs& g, int splits, hipStream_t st) {
    // 128x128 tiles once they alone fill the chip (large-batch steps, bulk sampling); 64x64 with split-K below that
    if ((long)cdiv(g.M, 128) * cdiv(g.N, 128) * splits >= 256)
        return mx_launch4<128, TERMS == 2 ? 32 : 64, AKC, BKC, IM2COL, TERMS, EPI>(g, splits, st);
    return mx_launch4<64, 64, AKC, BKC, IM2COL, TERMS, EPI>(g, splits, st);
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
bool wfd_mx_conv_direct(int M, int Cin, int Cout, int terms) {
    if (!((Cin == 32 || Cin == 64) && (Cout == 32 || Cout == 64))) return false;
    return cdiv(M, CV_ROWS) >= 256;               // enough 128-row chunks to give every CU one
}

int wfd_mx_launch(const GemmArgs& g, int layout, int splits, int terms, hipStream_t st) {
    if (layout == 3 && splits == 1 && wfd_mx_conv_direct(g.M, g.convCin, g.N, terms) && g.ldb == 8 * g.convCin) {
        if (g.convCin == 32) return g.N == 32 ? cv_launch<32, 32>(g, terms, st) : cv_launch<32, 64>(g, terms, st);
        // 64 -> 64: the whole weight matrix (66 KB as bf16) would leave one workgroup per CU; two 32-column slices keep three
        return cv_launch<64, 32>(g, terms, st);
    }
    switch (layout) {
        case 0: return mx_launch2<true, true, false>(g, splits, terms, st);
        case 1: return mx_launch2<true, false, false>(g, splits, terms, st);
        case 2: return mx_launch2<false, false, false>(g, splits, terms, st);
        case 3: return mx_launch2<true, true, true>(g, splits, terms, st);
        default: return WFD_ERR_ARG;
    }
}
