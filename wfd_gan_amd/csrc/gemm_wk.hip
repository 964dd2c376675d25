// Small-batch contraction with the K split INSIDE the workgroup and the consumer's math fused into the store (gfx950).
//
// At batch 64 a Linear layer of the hot path is a 64..192-row product: too few 64x64 tiles to fill 256 CUs, so gemm.hip
// splits K over workgroups into slabs and a second launch (epilogues.hip) sums them while applying bias / activation /
// dropout / chain factors.  That costs a dependent launch and a slab round trip per product.  Here one workgroup owns a
// small output tile (16*RT rows x 16*CT columns) and ALL of K; its NW waves each contract their own k-range into a private
// accumulator (v_mfma_f32_16x16x4_f32: exact f32 products, the parity arithmetic of gemm.hip), the accumulators are folded
// through LDS in wave order (deterministic), and the epilogue runs on the folded tile:
//   epi 0  raw store                               (consumers that reduce over rows / columns read it as one slab)
//   epi 1  out = act(acc * scale[n] + shift[n])    (wfd_ep_affine_act: bias + ReLU, eval BatchNorm + ELU/ReLU, src/model.py:20-34,140-196)
//   epi 2  out = acc * f(D[m][n]) * scale[n]; out2 = out * rowscale[m]      (wfd_ep_mul: closed-form autograd.grad chain,
//                                                                            src/mlp_df_wgan_train.py:204-213,229,273)
//   epi 3  z = acc + shift[n]; out = LeakyReLU(z) * mask; out2 = mask * (z > 0 ? 1 : slope)   (wfd_ep_lrelu_drop, src/model.py:58-65)
// Waves share no operand data, so each wave stages its own k-chunks global -> registers -> its own LDS region -> MFMA
// fragments with no workgroup barrier inside the k-loop (coalesced 128-byte row segments on the way in, conflict-free
// ds_read_b128 fragments on the way out); the only barrier precedes the fold.
// Measured on the critic's first layer (128 x 2048 x 1501, 256 workgroups of 4 waves, graph replay): 12.5 us, of which
// 4.7 us remain with loads, MFMAs and LDS writes all disabled (launch, first chunk, fold + store), 11.4 us with only the
// global loads disabled and 7.4 us with only the MFMAs disabled -- the loop runs at the f32 matrix pipe's rate
// (188 x 16 MFMA steps of 32 cycles per SIMD = 5.9 us at 2.1 GHz); deeper prefetch (a second register set) bought
// nothing and made hipcc spill the set to scratch.
// TERMS (the operand mode of wfd_set_gemm_mode): 0 = f32-input MFMA as above; 1 / 2 = the bf16 matrix pipe over the same fp32
// data (gemm_mx.hip's arithmetic): the fragments come out of LDS as fp32 exactly as before and are converted in registers --
// x -> hi = bf16(x) [, lo = bf16(x - hi)] -- the four consecutive k a lane holds being precisely the lane's share of one
// v_mfma_f32_16x16x16_bf16 (k = 4 * (lane >> 4) + 0..3), so one bf16 MFMA (TERMS = 1) or three (lo*hi + hi*lo + hi*hi,
// TERMS = 2, products ~1e-5 relative) replace four f32 ones and nothing else in the kernel changes.
// Layouts: NT (A[m][k], B[n][k]) and NN (A[m][k], B[k][n]); NT + IM2COL: A = channels-last activation of a k=8 "SAME" conv.
// Operands must be 16-byte aligned with leading dimensions that are multiples of 4 floats (every buffer of the step is).
#include "gemm_args.h"

typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
typedef short s16x4_ __attribute__((ext_vector_type(4)));

extern "C" int wfd_get_gemm_mode(void);

namespace {

#ifndef WFD_WK_PREFETCH_EPI
#define WFD_WK_PREFETCH_EPI 1
#endif
constexpr bool PREFETCH_EPI = WFD_WK_PREFETCH_EPI != 0;
constexpr int KC = 32;         // k per chunk (one 128-byte row segment)
constexpr int LDK = KC + 4;    // row stride of a k-contiguous tile: 36 floats, conflict-free ds_read_b128 / ds_write_b128

struct WkArgs {
    const float* A;
    const float* B;
    int M, N, K, lda, ldb;
    int epi, act, dmode;
    float* out;
    int ldo;
    float* out2;
    int ldo2;
    const float* scale;
    const float* shift;
    const float* D;
    int ldd;
    const float* rowscale;
    const float* mask;
    int ldmask;
    float slope;
    int mt, nt;      // tiles along M / N
    int cpw;         // k-chunks per wave
    int convL, convCin, convPad;     // IM2COL
};

__device__ __forceinline__ void wave_lds_fence() {
    // orders this wave's LDS writes before its later LDS reads (and the reverse) across lanes: LDS serves one wave's
    // instructions in issue order; the fence keeps the compiler from reordering them and waits for the counters
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// bid: the workgroup's index within ITS product (a launch may carry several products: gemm_wk_multi_kernel)
// NBUF: LDS buffers per wave.  2: the next chunk is parked beside the one being multiplied.  1: it overwrites it -- legal
// because the region is private to the wave and LDS serves a wave's instructions in issue order (the chunk's fragment reads
// precede the next chunk's writes in program order, the wave fence keeps the compiler from swapping them); half the LDS,
// which is what lets the 48 x 32 tile run with 8 waves (184 KB double-buffered, 92 KB single).
template <bool BKC, bool IM2COL, int RT, int CT, int NW, int TERMS, int NBUF = 2>
__device__ __forceinline__ void gemm_wk_body(const WkArgs& g, float* __restrict__ smem, const int bid) {
    constexpr int TM = 16 * RT, TN = 16 * CT;
    constexpr int NA = TM / 8;                       // float4 loads per lane for an A chunk (8 lanes cover a 128-byte row segment)
    constexpr int NB = TN / 8;                       // ... for a B chunk (either layout: 32 x TN floats)
    constexpr int A_F = TM * LDK;
    constexpr int LDB = BKC ? LDK : TN + 4;          // NN: [k][TN + 4]
    constexpr int B_F = BKC ? TN * LDK : KC * LDB;
    constexpr int BUF_F = A_F + B_F;
    constexpr int WAVE_F = NBUF * BUF_F;             // private to the wave
    constexpr int TNP = TN + 4;                      // row stride of the fold image
    static_assert(TM * TNP <= WAVE_F, "fold image must fit the wave's region");

    // the wave index as a scalar: everything derived from it (k-range, chunk loop, interior test) then stays wave-uniform
    // for the compiler too -- scalar branches instead of exec-masked code around every load
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g4 = lane >> 4, c16 = lane & 15;
    float* wsm = smem + wave * WAVE_F;

    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, so XCD x takes tiles [x*T8, (x+1)*T8) and the
    // row tiles that stream the same weight rows (consecutive v) share one L2
    const int T = g.mt * g.nt, T8 = (T + 7) >> 3;
    const int v = (bid & 7) * T8 + (bid >> 3);
    if (v >= T) return;                              // uniform over the workgroup, before any barrier
    const int m0 = (v % g.mt) * TM, n0 = (v / g.mt) * TN;

    const int nchunk = (g.K + KC - 1) / KC;
    const int c0 = wave * g.cpw, c1 = min(nchunk, c0 + g.cpw);

    f32x4_ acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = (f32x4_){0.f, 0.f, 0.f, 0.f};

    // ---- per-lane source offsets (rows / columns beyond the matrix are clamped: what they load only feeds outputs that
    //      are never stored)
    int offa[NA], lpa[NA], offb[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int gr = min(m0 + (lane >> 3) + 8 * j, g.M - 1);
        if constexpr (IM2COL) {
            offa[j] = gr * g.convCin - g.convPad * g.convCin + 4 * (lane & 7);
            lpa[j] = gr % g.convL - g.convPad;
        } else {
            offa[j] = gr * g.lda + 4 * (lane & 7);
            lpa[j] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if constexpr (BKC) {
            const int gn = min(n0 + (lane >> 3) + 8 * j, g.N - 1);
            offb[j] = gn * g.ldb + 4 * (lane & 7);
        } else {
            // 32 k-rows x TN columns: TN / 4 float4 per row
            constexpr int QR = TN / 4;               // float4 per k-row (8 or 4)
            int gc = n0 + 4 * (lane % QR);
            if (gc >= g.N) gc = 0;
            offb[j] = ((lane / QR) + (64 / QR) * j) * g.ldb + gc;
        }
    }

    float4 ra0[NA], rb0[NB];
    // interior chunks (the whole 32-k chunk inside K): unconditional 16-byte loads at offset + k0, nothing else
    auto load_fast = [&](int k0, float4 (&ra)[NA], float4 (&rb)[NB]) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if constexpr (IM2COL) {
                const bool ok = (unsigned)(lpa[j] + k0 / g.convCin) < (unsigned)g.convL;     // Cin % 32 == 0: one tap per chunk
                float4 v4 = *reinterpret_cast<const float4*>(g.A + (ok ? offa[j] + k0 : 0));
                if (!ok) v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                ra[j] = v4;
            } else {
                ra[j] = *reinterpret_cast<const float4*>(g.A + offa[j] + k0);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (BKC) rb[j] = *reinterpret_cast<const float4*>(g.B + offb[j] + k0);
            else rb[j] = *reinterpret_cast<const float4*>(g.B + offb[j] + (long)k0 * g.ldb);
        }
    };
    // the ragged last chunk: element-wise, zeros beyond K
    auto load_tail = [&](int k0, float4 (&ra)[NA], float4 (&rb)[NB]) {
        const int gk = k0 + 4 * (lane & 7);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            bool ok = true;
            if constexpr (IM2COL) ok = (unsigned)(lpa[j] + k0 / g.convCin) < (unsigned)g.convL;
            const float* p = g.A + (ok ? offa[j] + k0 : 0);
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (ok && gk + e < g.K) ? p[e] : 0.f;
            ra[j] = make_float4(t[0], t[1], t[2], t[3]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (BKC) {
                const float* p = g.B + offb[j] + k0;
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (gk + e < g.K) ? p[e] : 0.f;
                rb[j] = make_float4(t[0], t[1], t[2], t[3]);
            } else {
                constexpr int QR = TN / 4;
                const int kr = k0 + (lane / QR) + (64 / QR) * j;
                rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kr < g.K) rb[j] = *reinterpret_cast<const float4*>(g.B + offb[j] + (long)k0 * g.ldb);
            }
        }
    };
    auto store_chunk = [&](float* buf, const float4 (&ra)[NA], const float4 (&rb)[NB]) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
            *reinterpret_cast<float4*>(buf + ((lane >> 3) + 8 * j) * LDK + 4 * (lane & 7)) = ra[j];
        float* bs = buf + A_F;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (BKC) {
                *reinterpret_cast<float4*>(bs + ((lane >> 3) + 8 * j) * LDK + 4 * (lane & 7)) = rb[j];
            } else {
                constexpr int QR = TN / 4;
                *reinterpret_cast<float4*>(bs + ((lane / QR) + (64 / QR) * j) * LDB + 4 * (lane % QR)) = rb[j];
            }
        }
    };
    // MFMA step (j, s) contracts k = 16 j + 4 g + s over the four lane groups g, for A and B alike
    auto compute = [&](const float* buf) {
        const float* as = buf;
        const float* bs = buf + A_F;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4_ af[RT], bf[CT];
#pragma unroll
            for (int a = 0; a < RT; ++a) af[a] = *reinterpret_cast<const f32x4_*>(as + (16 * a + c16) * LDK + 4 * g4 + 16 * j);
#pragma unroll
            for (int b = 0; b < CT; ++b) {
                if constexpr (BKC) {
                    bf[b] = *reinterpret_cast<const f32x4_*>(bs + (16 * b + c16) * LDK + 4 * g4 + 16 * j);
                } else {
                    const float* p = bs + (4 * g4 + 16 * j) * LDB + 16 * b + c16;
                    bf[b] = (f32x4_){p[0], p[LDB], p[2 * LDB], p[3 * LDB]};
                }
            }
            if constexpr (TERMS == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int a = 0; a < RT; ++a)
#pragma unroll
                        for (int b = 0; b < CT; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
            } else {
                s16x4_ ah[RT], al[RT], bh[CT], bl[CT];
#pragma unroll
                for (int a = 0; a < RT; ++a) {
                    const bf16x4_ h = __builtin_convertvector(af[a], bf16x4_);
                    ah[a] = __builtin_bit_cast(s16x4_, h);
                    if constexpr (TERMS == 2)
                        al[a] = __builtin_bit_cast(s16x4_, __builtin_convertvector(af[a] - __builtin_convertvector(h, f32x4_), bf16x4_));
                }
#pragma unroll
                for (int b = 0; b < CT; ++b) {
                    const bf16x4_ h = __builtin_convertvector(bf[b], bf16x4_);
                    bh[b] = __builtin_bit_cast(s16x4_, h);
                    if constexpr (TERMS == 2)
                        bl[b] = __builtin_bit_cast(s16x4_, __builtin_convertvector(bf[b] - __builtin_convertvector(h, f32x4_), bf16x4_));
                }
#pragma unroll
                for (int a = 0; a < RT; ++a)
#pragma unroll
                    for (int b = 0; b < CT; ++b) {
                        if constexpr (TERMS == 2) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[a], bh[b], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        }
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    }
            }
        }
    };

    // ---- what the epilogue will read for this thread's float4 group of the tile (chain factor D, dropout mask, shift /
    //      scale, row scale) is requested HERE, ahead of the k-loop: behind the fold barrier each of these loads is a
    //      dependent round trip in front of the store, on the critical path of every launch of the batch-64 step
    constexpr int Q = TN / 4, NQ = TM * Q;           // float4 groups of the tile
    float pd[4] = {0.f, 0.f, 0.f, 0.f}, pm[4] = {0.f, 0.f, 0.f, 0.f}, psh[4] = {0.f, 0.f, 0.f, 0.f}, psc[4] = {1.f, 1.f, 1.f, 1.f};
    float prs = 0.f;
    if (PREFETCH_EPI && (int)threadIdx.x < NQ) {
        const int prow = m0 + (int)threadIdx.x / Q, pcol = n0 + 4 * ((int)threadIdx.x % Q);
        if (prow < g.M && pcol < g.N) {
            const int pnv = min(4, g.N - pcol);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < pnv) {
                    if (g.epi == 1) {
                        if (g.shift) psh[k] = g.shift[pcol + k];
                        if (g.scale) psc[k] = g.scale[pcol + k];
                    } else if (g.epi == 2) {
                        if (g.dmode) pd[k] = g.D[(long)prow * g.ldd + pcol + k];
                        if (g.scale) psc[k] = g.scale[pcol + k];
                    } else if (g.epi == 3) {
                        psh[k] = g.shift[pcol + k];
                        pm[k] = g.mask[(long)prow * g.ldmask + pcol + k];
                    }
                }
            }
            if (g.epi == 2 && g.out2) prs = g.rowscale[prow];
        }
    }

    // main loop over the wave's interior chunks only (the ragged last chunk of K, if this wave owns it, runs after the
    // loop: with its element-wise loads inside the loop body the compiler waits for every load right after issuing it)
    const int c1f = min(c1, g.K / KC);
    if (c0 < c1f) {
        load_fast(c0 * KC, ra0, rb0);
        store_chunk(wsm, ra0, rb0);
    }
    wave_lds_fence();
    int cur = 0;
    for (int c = c0; c < c1f; ++c) {
        const bool more = c + 1 < c1f;
        if (more) load_fast((c + 1) * KC, ra0, rb0);
        compute(wsm + (NBUF == 2 ? cur * BUF_F : 0));
        if constexpr (NBUF == 1) wave_lds_fence();
        if (more) store_chunk(wsm + (NBUF == 2 ? (cur ^ 1) * BUF_F : 0), ra0, rb0);
        wave_lds_fence();
        cur ^= 1;
    }
    if (c1 > c1f && c1f >= c0) {                      // this wave owns the ragged chunk
        load_tail(c1f * KC, ra0, rb0);
        store_chunk(wsm, ra0, rb0);
        wave_lds_fence();
        compute(wsm);
        wave_lds_fence();
    }

    // ---- fold: every wave parks its accumulators as a [TM][TNP] image in its own region; after the barrier the
    //      workgroup's threads each sum one float4 of the tile over the waves in wave order and apply the epilogue.
    //      C/D map of the 16x16 accumulator: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) wsm[(16 * a + 4 * g4 + r) * TNP + 16 * b + c16] = acc[a][b][r];
    __syncthreads();
    for (int q = threadIdx.x; q < NQ; q += NW * 64) {
        const bool first = PREFETCH_EPI && q == (int)threadIdx.x;       // this group's epilogue operands are in registers
        const int r = q / Q, cq = q % Q;
        const int row = m0 + r, col = n0 + 4 * cq;
        if (row >= g.M || col >= g.N) continue;
        float4 s = *reinterpret_cast<const float4*>(smem + r * TNP + 4 * cq);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(smem + w * WAVE_F + r * TNP + 4 * cq);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int nv = min(4, g.N - col);
        float vv[4] = {s.x, s.y, s.z, s.w}, v2[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.epi == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nv) {
                    const float sh = first ? psh[k] : (g.shift ? g.shift[col + k] : 0.f);
                    float x = g.scale ? __fmaf_rn(vv[k], first ? psc[k] : g.scale[col + k], sh) : vv[k] + sh;
                    if (g.act == 1) x = fmaxf(x, 0.f);
                    else if (g.act == 2) x = x > 0.f ? x : elu_expm1(x);
                    else if (g.act == 3) x = 1.0f / (1.0f + expf(-x));
                    vv[k] = x;
                }
            }
        } else if (g.epi == 2) {
            const float rs = first ? prs : (g.out2 ? g.rowscale[row] : 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nv) {
                    float x = vv[k];
                    if (g.dmode) {
                        const float d = first ? pd[k] : g.D[(long)row * g.ldd + col + k];
                        if (g.dmode == 1) x *= d;
                        else if (g.dmode == 2) x = d > 0.f ? x : 0.f;
                        else x = d > 0.f ? x : x * (d + 1.f);
                    }
                    if (g.scale) x *= first ? psc[k] : g.scale[col + k];
                    vv[k] = x;
                    v2[k] = x * rs;
                }
            }
        } else if (g.epi == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nv) {
                    const float zz = vv[k] + (first ? psh[k] : g.shift[col + k]);
                    const float mk = first ? pm[k] : g.mask[(long)row * g.ldmask + col + k];
                    const float lr = zz > 0.f ? zz : zz * g.slope;
                    vv[k] = lr * mk;
                    v2[k] = zz > 0.f ? mk : mk * g.slope;
                }
            }
        }
        float* o = g.out + (long)row * g.ldo + col;
        if (nv == 4 && ((g.ldo & 3) == 0)) *reinterpret_cast<float4*>(o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        else
            for (int k = 0; k < nv; ++k) o[k] = vv[k];
        if (g.out2) {
            float* o2 = g.out2 + (long)row * g.ldo2 + col;
            if (nv == 4 && ((g.ldo2 & 3) == 0)) *reinterpret_cast<float4*>(o2) = make_float4(v2[0], v2[1], v2[2], v2[3]);
            else
                for (int k = 0; k < nv; ++k) o2[k] = v2[k];
        }
    }
}

template <bool BKC, bool IM2COL, int RT, int CT, int NW, int TERMS, int NBUF = 2>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_wk_kernel(WkArgs g) {
    // at most two waves per SIMD are ever resident (LDS: 9-18 KB per wave), so the register allocator may use 256 VGPRs:
    // without the attribute it aims at four waves and spills the prefetched chunk to scratch
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_wk_body<BKC, IM2COL, RT, CT, NW, TERMS, NBUF>(g, smem, blockIdx.x);
}

// Several INDEPENDENT products of one layout and configuration in one launch (the observer's FC layer beside the critic's
// layer of the same depth in the G-step): each product's workgroups keep their own XCD-aware order (every product takes a
// multiple of 8 workgroups), the products' fill and drain phases overlap, and the step pays one launch for both.
constexpr int WK_MULTI_MAX = 3;
struct WkMulti { WkArgs g[WK_MULTI_MAX]; int nb[WK_MULTI_MAX]; int n; };

template <bool BKC, int RT, int CT, int NW, int TERMS>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_wk_multi_kernel(WkMulti a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int id = blockIdx.x, k = 0;
    while (k < a.n - 1 && id >= a.nb[k]) { id -= a.nb[k]; ++k; }       // uniform over the workgroup
    gemm_wk_body<BKC, false, RT, CT, NW, TERMS>(a.g[k], smem, id);
}

template <bool BKC, int RT, int CT, int NW, int TERMS>
int launch_wk_multi(WkMulti a, hipStream_t st) {
    constexpr int TM = 16 * RT, TN = 16 * CT;
    constexpr int LDB = BKC ? LDK : TN + 4;
    constexpr int WAVE_F = 2 * (TM * LDK + (BKC ? TN * LDK : KC * LDB));
    const size_t lds = (size_t)NW * WAVE_F * sizeof(float);
    WFD_LDS_ATTR((gemm_wk_multi_kernel<BKC, RT, CT, NW, TERMS>), lds);
    long total = 0;
    for (int k = 0; k < a.n; ++k) {
        WkArgs& g = a.g[k];
        g.mt = cdiv(g.M, TM);
        g.nt = cdiv(g.N, TN);
        g.cpw = cdiv(cdiv(g.K, KC), NW);
        a.nb[k] = 8 * cdiv((long)g.mt * g.nt, 8);
        total += a.nb[k];
    }
    hipLaunchKernelGGL((gemm_wk_multi_kernel<BKC, RT, CT, NW, TERMS>), dim3((unsigned)total), dim3(NW * 64), lds, st, a);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

template <bool BKC, int TERMS>
int launch_wk_multi_cfg(const WkMulti& a, int cfg, hipStream_t st) {
    switch (cfg) {          // the two configurations the batch-64 step's paired products use
        case 3: return launch_wk_multi<BKC, 2, 1, 4, TERMS>(a, st);
        case 4: return launch_wk_multi<BKC, 2, 1, 8, TERMS>(a, st);
        default: return WFD_ERR_ARG;
    }
}

template <bool BKC, bool IM2COL, int RT, int CT, int NW, int TERMS, int NBUF = 2>
int launch_wk(WkArgs g, hipStream_t st) {
    constexpr int TM = 16 * RT, TN = 16 * CT;
    constexpr int LDB = BKC ? LDK : TN + 4;
    constexpr int WAVE_F = NBUF * (TM * LDK + (BKC ? TN * LDK : KC * LDB));
    const size_t lds = (size_t)NW * WAVE_F * sizeof(float);
    WFD_LDS_ATTR((gemm_wk_kernel<BKC, IM2COL, RT, CT, NW, TERMS, NBUF>), lds);
    g.mt = cdiv(g.M, TM);
    g.nt = cdiv(g.N, TN);
    g.cpw = cdiv(cdiv(g.K, KC), NW);
    const int T8 = cdiv((long)g.mt * g.nt, 8);
    hipLaunchKernelGGL((gemm_wk_kernel<BKC, IM2COL, RT, CT, NW, TERMS, NBUF>), dim3(8 * T8), dim3(NW * 64), lds, st, g);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

template <bool BKC, bool IM2COL, int TERMS>
int launch_wk_cfg2(const WkArgs& g, int cfg, hipStream_t st) {
    switch (cfg) {
        case 1: return launch_wk<BKC, IM2COL, 2, 2, 4, TERMS>(g, st);      // 32 x 32 tile, 4 waves
        case 2: return launch_wk<BKC, IM2COL, 2, 2, 8, TERMS>(g, st);      // 32 x 32 tile, 8 waves
        case 3: return launch_wk<BKC, IM2COL, 2, 1, 4, TERMS>(g, st);      // 32 x 16 tile, 4 waves
        case 4: return launch_wk<BKC, IM2COL, 2, 1, 8, TERMS>(g, st);      // 32 x 16 tile, 8 waves
        case 5: return launch_wk<BKC, IM2COL, 2, 2, 2, TERMS>(g, st);      // 32 x 32 tile, 2 waves (many tiles, short K)
        case 6:                                                            // 48 x 32 tile, 8 waves, one LDS buffer per wave
            if constexpr (!IM2COL) return launch_wk<BKC, false, 3, 2, 8, TERMS, 1>(g, st);
            return WFD_ERR_ARG;
        default: return WFD_ERR_ARG;
    }
}

// operand mode (wfd_set_gemm_mode) read when the launch is issued (or captured), as in gemm.hip
template <bool BKC, bool IM2COL>
int launch_wk_cfg(const WkArgs& g, int cfg, hipStream_t st) {
    switch (wfd_get_gemm_mode()) {
        case 1: return launch_wk_cfg2<BKC, IM2COL, 1>(g, cfg, st);
        case 2: return launch_wk_cfg2<BKC, IM2COL, 2>(g, cfg, st);
        default: return launch_wk_cfg2<BKC, IM2COL, 0>(g, cfg, st);
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

// Configuration the library would pick for an (M, N, K) product of the given layout (0 NT, 1 NN, 3 conv), or 0 when the
// split-inside-the-workgroup form does not pay (tall products: the 64x64 / 128x128 tiles of gemm.hip stream less per FLOP).
int wfd_gemm_wk_pick(int layout, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < KC) return 0;
    const long t32 = (long)cdiv(M, 32) * cdiv(N, 32), t16 = (long)cdiv(M, 32) * cdiv(N, 16);
    const int nchunk = cdiv(K, KC);
    if (layout == 3) {
        // conv rows are plentiful: one wave pair per 32 x 32 tile while K is short
        // (tried in round 3, not kept: 513 .. 768 tiles -- the 64-channel convs at L = 175 -- with ONE LDS buffer per wave so
        // that three workgroups fit a CU and the launch is a single round: 3551 vs 3565 it/s, same session)
        if (t32 >= 96) return nchunk >= 16 ? 1 : 5;
        // the short, wide convs at the bottom of the observer (few rows, K = 8 Cin >= 1024): 32 x 16 tiles with K over 8 waves
        // instead of split-K slabs + a consumer launch
        return (t16 >= 32 && nchunk >= 16) ? 4 : 0;
    }
    if (M > 256) return 0;
    // 129 .. 256 rows (the stacked generator forwards of a group: 192; the critic under --penalty gp: 192): with 32-row tiles
    // the tile count lands just above a multiple of the 256 CUs (192 x 1401: 264 tiles, 192 x 2048: 384) and the launch runs
    // two rounds of workgroups; 48-row tiles (cfg 6) make it one round of 1.5x the work each
    // (NN too: under --penalty gp the critic's chain products have 192 rows as well; in-step tuner, profiles/r03_wk_tune_gp.json:
    // 192 x 1401 x 2048 -2.4 %, 192 x 2048 x 1024 -1.0 % per group)
    if ((layout == 0 || layout == 1) && M > 128 && nchunk >= 16) {
        const long t48 = (long)cdiv(M, 48) * cdiv(N, 32);
        static const int use48 = getenv("WFD_WK_48") ? atoi(getenv("WFD_WK_48")) : 1;       // (the switch: same-session A/B)
        if (use48 && cdiv(t48, 256) * 48 < cdiv(t32, 256) * 32) return 6;
    }
    // the critic's input gradient (NN, 128 x 1401 x 2048: 176 tiles, a long K and a consumer -- the penalty row kernel -- that
    // sums slabs anyway): split-K slabs beat every configuration here in the step (tools/wk_tune.py: -0.9 % per group)
    if (layout == 1 && M > 64 && t32 >= 128 && t32 < 192 && nchunk >= 48) return 0;
    if (t32 >= 192) return nchunk >= 32 ? 2 : 1;
    if (t16 >= 96) return nchunk >= 32 ? 4 : 3;
    return nchunk >= 8 ? 4 : 3;
}

int wfd_gemm_f32_wk(int layout, const float* A, int lda, const float* B, int ldb, int M, int N, int K, int epi, float* out,
                    int ldo, float* out2, int ldo2, const float* scale, const float* shift, int act, const float* D, int ldd,
                    int dmode, const float* rowscale, const float* mask, int ldmask, float slope, int cfg, void* stream) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || ldo < N || epi < 0 || epi > 3) return WFD_ERR_ARG;
    if (layout != 0 && layout != 1) return WFD_ERR_ARG;
    if (!al16(A) || !al16(B) || (lda & 3) || (ldb & 3) || lda < K || ldb < (layout == 0 ? K : N)) return WFD_ERR_ARG;
    if ((long)M * lda >= (1L << 31) || (long)(layout == 0 ? N : K) * ldb >= (1L << 31)) return WFD_ERR_ARG;
    if (out2 && ldo2 < N) return WFD_ERR_ARG;
    if (epi == 2 && ((dmode && !D) || (out2 && !rowscale))) return WFD_ERR_ARG;
    if (epi == 3 && (!shift || !mask || !out2)) return WFD_ERR_ARG;
    if (epi < 2 && out2) return WFD_ERR_ARG;
    if (!cfg) cfg = wfd_gemm_wk_pick(layout, M, N, K);
    if (!cfg) cfg = 3;
    WkArgs g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.epi = epi; g.act = act; g.dmode = dmode; g.out = out; g.ldo = ldo; g.out2 = out2; g.ldo2 = ldo2;
    g.scale = scale; g.shift = shift; g.D = D; g.ldd = ldd; g.rowscale = rowscale; g.mask = mask; g.ldmask = ldmask;
    g.slope = slope;
    hipStream_t st = (hipStream_t)stream;
    if (layout == 0) return launch_wk_cfg<true, false>(g, cfg, st);
    return launch_wk_cfg<false, false>(g, cfg, st);
}

// desc (HOST memory): n <= 3 products x 24 64-bit words, the arguments of wfd_gemm_f32_wk in order
// [layout, A, lda, B, ldb, M, N, K, epi, out, ldo, out2, ldo2, scale, shift, act, D, ldd, dmode, rowscale, mask, ldmask,
//  slope (the float's bits), cfg].  All products must share layout and configuration (3 or 4); returns WFD_ERR_ARG otherwise
// (the caller launches them one by one).
int wfd_gemm_f32_wk_multi(const long long* desc, int n, void* stream) {
    if (!desc || n < 1 || n > WK_MULTI_MAX) return WFD_ERR_ARG;
    WkMulti a{};
    a.n = n;
    const int layout = (int)desc[0], cfg = (int)desc[23];
    for (int k = 0; k < n; ++k) {
        const long long* d = desc + 24 * k;
        WkArgs& g = a.g[k];
        g.A = reinterpret_cast<const float*>(d[1]); g.lda = (int)d[2];
        g.B = reinterpret_cast<const float*>(d[3]); g.ldb = (int)d[4];
        g.M = (int)d[5]; g.N = (int)d[6]; g.K = (int)d[7]; g.epi = (int)d[8];
        g.out = reinterpret_cast<float*>(d[9]); g.ldo = (int)d[10];
        g.out2 = reinterpret_cast<float*>(d[11]); g.ldo2 = (int)d[12];
        g.scale = reinterpret_cast<const float*>(d[13]); g.shift = reinterpret_cast<const float*>(d[14]); g.act = (int)d[15];
        g.D = reinterpret_cast<const float*>(d[16]); g.ldd = (int)d[17]; g.dmode = (int)d[18];
        g.rowscale = reinterpret_cast<const float*>(d[19]); g.mask = reinterpret_cast<const float*>(d[20]); g.ldmask = (int)d[21];
        const unsigned sb = (unsigned)d[22];
        __builtin_memcpy(&g.slope, &sb, sizeof(float));
        if ((int)d[0] != layout || (int)d[23] != cfg || (layout != 0 && layout != 1)) return WFD_ERR_ARG;
        if (!g.A || !g.B || !g.out || g.M <= 0 || g.N <= 0 || g.K <= 0 || g.ldo < g.N || g.epi < 0 || g.epi > 3) return WFD_ERR_ARG;
        if (!al16(g.A) || !al16(g.B) || (g.lda & 3) || (g.ldb & 3) || g.lda < g.K || g.ldb < (layout == 0 ? g.K : g.N)) return WFD_ERR_ARG;
        if ((long)g.M * g.lda >= (1L << 31) || (long)(layout == 0 ? g.N : g.K) * g.ldb >= (1L << 31)) return WFD_ERR_ARG;
        if (g.out2 && g.ldo2 < g.N) return WFD_ERR_ARG;
        if (g.epi == 2 && ((g.dmode && !g.D) || (g.out2 && !g.rowscale))) return WFD_ERR_ARG;
        if (g.epi == 3 && (!g.shift || !g.mask || !g.out2)) return WFD_ERR_ARG;
        if (g.epi < 2 && g.out2) return WFD_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int mode = wfd_get_gemm_mode();
    if (layout == 0) {
        if (mode == 1) return launch_wk_multi_cfg<true, 1>(a, cfg, st);
        if (mode == 2) return launch_wk_multi_cfg<true, 2>(a, cfg, st);
        return launch_wk_multi_cfg<true, 0>(a, cfg, st);
    }
    if (mode == 1) return launch_wk_multi_cfg<false, 1>(a, cfg, st);
    if (mode == 2) return launch_wk_multi_cfg<false, 2>(a, cfg, st);
    return launch_wk_multi_cfg<false, 0>(a, cfg, st);
}

// k=8 "SAME" conv over a channels-last activation (wfd_conv8_gemm_f32's contraction) in the same form; Cin % 32 == 0
int wfd_conv8_f32_wk(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, int epi, float* out, int ldo,
                     const float* scale, const float* shift, int act, const float* D, int ldd, int dmode, int cfg,
                     void* stream) {
    if (!X || !Wp || !out || Bn <= 0 || L <= 0 || Cin <= 0 || (Cin & 31) || Cout <= 0 || ldo < Cout) return WFD_ERR_ARG;
    if (!al16(X) || !al16(Wp) || epi < 0 || epi > 2 || (epi == 2 && dmode && !D)) return WFD_ERR_ARG;
    if ((long)Bn * L * Cin >= (1L << 31)) return WFD_ERR_ARG;
    WkArgs g{};
    g.A = X; g.B = Wp; g.M = Bn * L; g.N = Cout; g.K = 8 * Cin; g.lda = Cin; g.ldb = 8 * Cin;
    g.epi = epi; g.act = act; g.dmode = dmode; g.out = out; g.ldo = ldo;
    g.scale = scale; g.shift = shift; g.D = D; g.ldd = ldd;
    g.convL = L; g.convCin = Cin; g.convPad = pad;
    if (!cfg) cfg = wfd_gemm_wk_pick(3, g.M, g.N, g.K);
    if (!cfg) cfg = 5;
    return launch_wk_cfg<true, true>(g, cfg, (hipStream_t)stream);
}

}  // extern "C"
