// fp32-operand MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate).
//
// One kernel template covers every dense contraction on the wfd-gan hot path:
//   NT  P[m][n] = sum_k A[m][k] * B[n][k]     forward Linear (weights are (out,in) row-major), penalty t-chain
//   NN  P[m][n] = sum_k A[m][k] * B[k][n]     input-gradient chains  (delta * W)
//   TN  P[m][n] = sum_k A[k][m] * B[k][n]     weight gradients (reduction over batch rows)
//   NT + IM2COL: A is a channels-last activation [B][L][Cin] read as the im2col matrix of a k=8 "SAME" conv,
//                A[m][k] = X[m*Cin + k - pad*Cin] masked by 0 <= (m % L) + k/Cin - pad < L.
// Raw products only: the batch dimension is tiny (64..128 rows) so the K dimension is split over gridDim.z to
// fill 256 CUs; each split writes its own slab P[z][M][ldp] with plain stores (deterministic, no atomics) and
// the consumer epilogue kernel (epilogues.hip / df.hip) sums the slabs in fixed order while applying the fused
// bias / activation / dropout / BN / penalty math.
//
// Tile: 64x64x32 per 256-thread workgroup, 2x2 waves of one 32x32 accumulator each.  Operands are staged
// global -> registers -> LDS (double buffered, one barrier per k-tile).  A k-contiguous operand tile is stored
// [row][36] and read with one ds_read_b128 per 4 MFMAs (conflict-free: 36 mod 64 walks all 16 slots); an
// m-contiguous one is stored [k][68] and read with ds_read_b32.  Lane half h of MFMA step s consumes
// k = 8*kk + 4*h + s for BOTH operands, so the k-permutation inside a tile is consistent.
#include "gemm_args.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, NTHREADS = 256;

// Diagnostic build only (make stamps -> libwfdgan_stamps.so, tools/gemm_stamps.py): per-workgroup s_memrealtime stamps
// (100 MHz) at entry, after the first tile is in LDS, after the k-loop and after the stores, written to the otherwise
// unused GemmArgs::D of a raw (EPI = false) launch.  The product library is built without WFD_STAMPS: no stamp executes.
#ifdef WFD_STAMPS
#define WFD_STAMP(slot)                                                                                         \
    if ((EPI == 0 || EPI == 2) && g.D && threadIdx.x == 0) {                                                                       \
        const long wg_ = (long)bz * 65536 + (long)by * 256 + bx;                                                 \
        reinterpret_cast<unsigned long long*>(const_cast<float*>(g.D))[(wg_ % g.ldd) * 4 + slot] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define WFD_STAMP(slot)
#endif
constexpr int LDK = BK + 4;  // row stride of a k-contiguous tile  [64][36]
constexpr int LDM = BM + 4;  // row stride of an m-contiguous tile [32][68]
constexpr int TILE_F = 64 * LDK;  // 2304 floats >= 32*68 = 2176

// ---- global -> register staging -------------------------------------------------------------------------
// KC = operand is k-contiguous in global memory ([rows][K], ld); else row(k)-major with the m/n index contiguous.
template <bool KC, int VEC, bool IM2COL>
__device__ __forceinline__ void load_tile(float (&r)[8], const float* __restrict__ G, int ld, int row0, int nrows,
                                          int k0, int kend, int tid, int convL, int convCin, int convPad) {
    if constexpr (KC) {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rr = (tid >> 3) + 32 * j, gk = k0 + 4 * (tid & 7), gr = row0 + rr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                bool ok = gr < nrows && gk < kend;
                long off;
                if constexpr (IM2COL) {
                    const int l = gr % convL, t = gk / convCin;
                    ok = ok && (unsigned)(l + t - convPad) < (unsigned)convL;
                    off = (long)gr * convCin + gk - (long)convPad * convCin;
                } else {
                    off = (long)gr * ld + gk;
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
                r[4 * j + 0] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rr = (tid >> 5) + 8 * j, gk = k0 + (tid & 31), gr = row0 + rr;
                r[j] = (gr < nrows && gk < kend) ? G[(long)gr * ld + gk] : 0.f;
            }
        }
    } else {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kr = (tid >> 4) + 16 * j, gk = k0 + kr, gc = row0 + 4 * (tid & 15);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                bool ok = gk < kend && gc < nrows;
                long off = (long)gk * ld + gc;
                if constexpr (IM2COL) {
                    // conv weight gradient: k runs over (sample, position) rows of the channels-last activation,
                    // the column is (tap, cin); 4 consecutive columns share a tap (Cin % 4 == 0)
                    const int l = gk % convL, t = gc / convCin;
                    ok = ok && (unsigned)(l + t - convPad) < (unsigned)convL;
                    off = (long)gk * convCin + gc - (long)convPad * convCin;
                }
                if (ok) {
                    if (gc + 3 < nrows) {
                        v = *reinterpret_cast<const float4*>(G + off);
                    } else {
                        v.x = G[off];
                        if (gc + 1 < nrows) v.y = G[off + 1];
                        if (gc + 2 < nrows) v.z = G[off + 2];
                    }
                }
                r[4 * j + 0] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kr = (tid >> 6) + 4 * j, gk = k0 + kr, gc = row0 + (tid & 63);
                r[j] = (gk < kend && gc < nrows) ? G[(long)gk * ld + gc] : 0.f;
            }
        }
    }
}

template <bool KC, int VEC>
__device__ __forceinline__ void store_tile(const float (&r)[8], float* __restrict__ S, int tid) {
    if constexpr (KC) {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rr = (tid >> 3) + 32 * j, kq = tid & 7;
                *reinterpret_cast<float4*>(S + rr * LDK + 4 * kq) =
                    make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) S[((tid >> 5) + 8 * j) * LDK + (tid & 31)] = r[j];
        }
    } else {
        if constexpr (VEC == 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kr = (tid >> 4) + 16 * j, mq = tid & 15;
                *reinterpret_cast<float4*>(S + kr * LDM + 4 * mq) =
                    make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) S[((tid >> 6) + 4 * j) * LDM + (tid & 63)] = r[j];
        }
    }
}

// Interior k-tiles (whole tile inside the split's k-range) of 16-byte-aligned operands take a branch-free form: the
// per-thread element offsets are computed once before the k-loop with out-of-range rows / columns CLAMPED to valid
// ones (what they load only feeds output rows / columns that are never stored), so a tile is two unconditional
// 16-byte loads at offset + constant.  Leading dimensions are multiples of 4 floats here, so a 16-byte load that
// starts inside a row stays inside its allocation.  The implicit-GEMM operand also zeroes taps outside the sample.
template <bool KC, bool IM2COL>
__device__ __forceinline__ void tile_offsets(int (&off)[2], int (&lpos)[2], int ld, int row0, int nrows, int kbeg, int tid,
                                             int convL, int convCin, int convPad) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        lpos[j] = 0;
        if constexpr (KC) {
            const int gr = min(row0 + (tid >> 3) + 32 * j, nrows - 1), gk = kbeg + 4 * (tid & 7);
            if constexpr (IM2COL) {
                off[j] = gr * convCin + gk - convPad * convCin;
                lpos[j] = gr % convL - convPad;
            } else {
                off[j] = gr * ld + gk;
            }
        } else {
            int gc = row0 + 4 * (tid & 15);
            if (gc >= nrows) gc = 0;
            if constexpr (IM2COL) {       // conv weight gradient: rows are (sample, position), columns (tap, cin)
                off[j] = (kbeg + (tid >> 4) + 16 * j) * convCin + gc - convPad * convCin;
                lpos[j] = gc / convCin - convPad;
            } else {
                off[j] = (kbeg + (tid >> 4) + 16 * j) * ld + gc;
            }
        }
    }
}

template <bool KC, bool IM2COL>
__device__ __forceinline__ void load_tile_fast(float (&r)[8], const float* __restrict__ G, const int (&off)[2],
                                               const int (&lpos)[2], int step, int k0, int tid, int convL, int convCin) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float4 v;
        if constexpr (IM2COL) {
            bool ok;
            if constexpr (KC) ok = (unsigned)(lpos[j] + (k0 + 4 * (tid & 7)) / convCin) < (unsigned)convL;
            else ok = (unsigned)((k0 + (tid >> 4) + 16 * j) % convL + lpos[j]) < (unsigned)convL;
            v = *reinterpret_cast<const float4*>(G + (ok ? off[j] + step : 0));
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v = *reinterpret_cast<const float4*>(G + off[j] + step);
        }
        r[4 * j + 0] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
    }
}

// fragment of MFMA steps s = 0..3 of k-group kk for the 32-row block starting at row0 (i = lane & 31, h = lane >> 5)
template <bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ S, int row0, int kk, int i, int h) {
    if constexpr (KC) {
        return *reinterpret_cast<const f32x4*>(S + (row0 + i) * LDK + kk * 8 + 4 * h);
    } else {
        f32x4 v;
        const float* p = S + (kk * 8 + 4 * h) * LDM + row0 + i;
        v[0] = p[0]; v[1] = p[LDM]; v[2] = p[2 * LDM]; v[3] = p[3 * LDM];
        return v;
    }
}

// NARROW / EPI are compile-time so that the plain path keeps its instruction schedule (a runtime branch inside the
// k-loop made hipcc stop hoisting the next tile's global loads above the MFMA block).
// EPI: 0 raw slabs; 1 fused store (epi_store); 2 RMSprop in the store with the parameter / square-average tile PREFETCHED
// into registers right after the first k-tile, so that their 2 x 16 loads per lane land during the k-loop instead of
// stalling the store (the store then only adds the second write stream).
// The square averages are touched by the weight-gradient + RMSprop launches only, once per step (45.6 MB read + written per critic step): as
// NON-TEMPORAL accesses they stream past the caches instead of evicting what the next launches read.  Same-session A/B of
// the whole step (round 3): everything cacheable 3493, everything streaming (-DWFD_RMS_NT=1) 3560, parameters' stores
// cacheable (=2) 3570, ONLY the square averages streaming (=3, the default) 3586 it/s: +2.7 % -- the new parameters are
// what the next step's products read, so they stay cacheable.  The launch itself takes the same time either way; the
// gain is in the launches after it.
#ifndef WFD_RMS_NT
#define WFD_RMS_NT 3
#endif
#define WFD_NT_LOAD(p) __builtin_nontemporal_load(p)
#define WFD_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#define WFD_T_LOAD(p) (*(p))
#define WFD_T_STORE(v, p) (*(p) = (v))
#if WFD_RMS_NT == 1            // everything streams
#define WFD_RMS_P_LOAD WFD_NT_LOAD
#define WFD_RMS_P_STORE WFD_NT_STORE
#define WFD_RMS_S_LOAD WFD_NT_LOAD
#define WFD_RMS_S_STORE WFD_NT_STORE
#elif WFD_RMS_NT == 2          // the new parameters stay cacheable (the next step's products read them)
#define WFD_RMS_P_LOAD WFD_NT_LOAD
#define WFD_RMS_P_STORE WFD_T_STORE
#define WFD_RMS_S_LOAD WFD_NT_LOAD
#define WFD_RMS_S_STORE WFD_NT_STORE
#elif WFD_RMS_NT == 3          // only the square averages stream
#define WFD_RMS_P_LOAD WFD_T_LOAD
#define WFD_RMS_P_STORE WFD_T_STORE
#define WFD_RMS_S_LOAD WFD_NT_LOAD
#define WFD_RMS_S_STORE WFD_NT_STORE
#else
#define WFD_RMS_P_LOAD WFD_T_LOAD
#define WFD_RMS_P_STORE WFD_T_STORE
#define WFD_RMS_S_LOAD WFD_T_LOAD
#define WFD_RMS_S_STORE WFD_T_STORE
#endif

template <bool AKC, bool BKC, int VA, int VB, bool IM2COL, bool NARROW, int EPI>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, float* __restrict__ lds, int bx, int by, int bz) {
    // lds: 4 * TILE_F floats; buffer c: A tile at lds + 2c*TILE_F, B tile right behind it

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    const int m0 = by * BM, n0 = bx * (NARROW ? 32 : BN), z = bz;
    // narrow: wave column index = half of the k-tile it multiplies (kk 0-1 / 2-3), both on columns n0..n0+31
    const int bcol0 = NARROW ? 0 : wn * 32;
    const int kbeg = z * g.ktiles_per_split * BK;
    const int kend = min(g.K, kbeg + g.ktiles_per_split * BK);
    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    WFD_STAMP(0)
    float ra[8], rb[8];
    constexpr bool FAST = (VA == 4 && VB == 4);
    int offa[2], offb[2], lpa[2], lpb[2];
    int stepa = 0, stepb = 0;
    if constexpr (FAST) {
        tile_offsets<AKC, IM2COL && AKC>(offa, lpa, g.lda, m0, g.M, kbeg, tid, g.convL, g.convCin, g.convPad);
        tile_offsets<BKC, IM2COL && !AKC>(offb, lpb, g.ldb, n0, g.N, kbeg, tid, g.convL, g.convCin, g.convPad);
        stepa = AKC ? BK : BK * g.lda;
        stepb = BKC ? BK : BK * ((IM2COL && !AKC) ? g.convCin : g.ldb);
    }
    auto load_both = [&](int kt) {
        const int k0 = kbeg + kt * BK;
        bool fast = false;
        if constexpr (FAST) fast = k0 + BK <= kend;        // wave-uniform: interior tile
        if (fast) {
            load_tile_fast<AKC, IM2COL && AKC>(ra, g.A, offa, lpa, kt * stepa, k0, tid, g.convL, g.convCin);
            load_tile_fast<BKC, IM2COL && !AKC>(rb, g.B, offb, lpb, kt * stepb, k0, tid, g.convL, g.convCin);
        } else {
            load_tile<AKC, VA, IM2COL && AKC>(ra, g.A, g.lda, m0, g.M, k0, kend, tid, g.convL, g.convCin, g.convPad);
            load_tile<BKC, VB, IM2COL && !AKC>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid, g.convL, g.convCin, g.convPad);
        }
    };
    if (nk > 0) {
        load_both(0);
        store_tile<AKC, VA>(ra, lds, tid);
        store_tile<BKC, VB>(rb, lds + TILE_F, tid);
    }
    __syncthreads();
    WFD_STAMP(1)
    float pp[16], sv[16];
    auto prefetch_params = [&]() {
        // issued AFTER the last operand tile's loads: loads retire in order, so a prefetch issued earlier would sit in front
        // of every tile the k-loop still waits for
        const int pcol = min(n0 + bcol0 + i, g.N - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1);
            pp[r] = g.P[(long)prow * g.ldp + pcol];
            sv[r] = WFD_RMS_S_LOAD(g.sq + (long)prow * g.ldp + pcol);
        }
    };
    int cur = 0;
    auto step = [&](int kt, bool more, bool pf) {
        if (more) load_both(kt + 1);
        if constexpr (EPI == 2) { if (pf) prefetch_params(); }
        const float* a_s = lds + cur * 2 * TILE_F;
        const float* b_s = a_s + TILE_F;
        if constexpr (!NARROW) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 a = read_frag<AKC>(a_s, wm * 32, kk, i, h);
                const f32x4 b = read_frag<BKC>(b_s, wn * 32, kk, i, h);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 a = read_frag<AKC>(a_s, wm * 32, 2 * wn + j, i, h);
                const f32x4 b = read_frag<BKC>(b_s, 0, 2 * wn + j, i, h);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            }
        }
        if (more) {
            store_tile<AKC, VA>(ra, lds + (cur ^ 1) * 2 * TILE_F, tid);
            store_tile<BKC, VB>(rb, lds + (cur ^ 1) * 2 * TILE_F + TILE_F, tid);
        }
        __syncthreads();
        cur ^= 1;
    };
    if constexpr (EPI == 2) {
        // peeled: the parameter / square-average tile is requested right behind the loads of the LAST operand tile, so it
        // lands during the last two tiles' MFMA blocks (and the other resident workgroups' loops)
        for (int kt = 0; kt + 2 < nk; ++kt) step(kt, true, false);
        if (nk >= 2) step(nk - 2, true, true);
        if (nk < 2) prefetch_params();
        if (nk >= 1) step(nk - 1, false, false);
    } else {
        for (int kt = 0; kt < nk; ++kt) step(kt, kt + 1 < nk, false);
    }

    WFD_STAMP(2)
    // C/D map of the 32x32 accumulator: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* P = g.P + (long)z * g.slab_stride;
    if constexpr (NARROW) {
        // fold the two k-halves: waves with wn = 1 park their accumulator in LDS, wn = 0 adds and stores
        float* red = lds + wm * (16 * 64);
        if (wn == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wn == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane];
    }
    const int col = n0 + bcol0 + i;
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < g.M) {
                if constexpr (EPI == 2) {
                    rmsprop_update(pp[r], sv[r], acc[r], g.lr, g.alpha, g.eps);
                    WFD_RMS_S_STORE(sv[r], g.sq + (long)row * g.ldp + col);
                    P[(long)row * g.ldp + col] = pp[r];
                } else if constexpr (EPI == 1) {
                    epi_store(g, P, acc[r], row, col);
                } else {
                    P[(long)row * g.ldp + col] = acc[r];
                }
            }
        }
    }
#ifdef WFD_STAMPS
    __builtin_amdgcn_s_waitcnt(0);     // the stores have left the wave (vmcnt 0) before the last stamp
#endif
    WFD_STAMP(3)
}

#ifdef WFD_STAMPS
}  // namespace
// diagnostic entry: a raw wfd_gemm_f32-style launch whose workgroups write their stamps to `stamps` (nwg x 4 u64)
extern "C" int wfd_gemm_f32_stamped(int layout, const float* A, int lda, const float* B, int ldb, float* P, int ldp,
                                    long slab_stride, int M, int N, int K, int splits, unsigned long long* stamps, int nwg,
                                    void* stream);
namespace {
#endif

// rgy > 0: a 1-D grid whose index is remapped so that the rgy row tiles of one (column tile, split) land on ONE XCD
// (workgroups b and b + 8 share an XCD): with a column-tile count that is no multiple of 8 the plain (x, y, z) order puts
// them on different XCDs and each of their L2s fetches the same weight tile from HBM -- the critic's input gradient
// (128 x 1401 x 2048: 22 column tiles, 2 row tiles, 11 splits) read 32 MB for a 12.3 MB weight matrix.  Speed only: any
// placement gives the same result.
template <bool AKC, bool BKC, int VA, int VB, bool IM2COL, bool NARROW, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmArgs g, int rgx, int rgy, int rgz) {
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE_F];
    if (rgy > 0) {
        const int id = blockIdx.x, by = (id >> 3) % rgy, t = (id / (8 * rgy)) * 8 + (id & 7);
        if (t >= rgx * rgz) return;                  // uniform over the workgroup, before any barrier
        gemm_f32_body<AKC, BKC, VA, VB, IM2COL, NARROW, EPI>(g, lds, t % rgx, by, t / rgx);
        return;
    }
    gemm_f32_body<AKC, BKC, VA, VB, IM2COL, NARROW, EPI>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Two INDEPENDENT products in one launch (a weight gradient beside the next step of the backward chain).  At batch 64
// every workgroup of a GEMM launch is resident at once, so its first-tile load latency and its final store burst overlap
// nobody's main loop (about half of the launch: DESIGN.md section 6); sharing a grid lets one product's fill / drain
// overlap the other's MFMA work without the cross-stream hand-offs a two-stream schedule costs.  The workgroups of the
// smaller product are spread evenly through the block order.  Problem 1: layout (A1KC, B1KC); problem 2: (A2KC, B2KC);
// both 16-byte aligned, raw slabs.
struct PairDims { int gx1, gy1, nb1, gx2, gy2, nb2; };

template <bool A1KC, bool B1KC, bool A2KC, bool B2KC>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_pair_kernel(GemmArgs g1, GemmArgs g2, PairDims d) {
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE_F];
    const int id = blockIdx.x, tot = d.nb1 + d.nb2;
    const int before = (int)((long)id * d.nb2 / tot), upto = (int)((long)(id + 1) * d.nb2 / tot);
    if (upto > before) {                   // wave-uniform: this block belongs to problem 2
        const int j = before, per = d.gx2 * d.gy2;
        gemm_f32_body<A2KC, B2KC, 4, 4, false, false, 0>(g2, lds, (j % per) % d.gx2, (j % per) / d.gx2, j / per);
    } else {
        const int j = id - before, per = d.gx1 * d.gy1;
        gemm_f32_body<A1KC, B1KC, 4, 4, false, false, 0>(g1, lds, (j % per) % d.gx1, (j % per) / d.gx1, j / per);
    }
}

// Several weight gradients with RMSprop in their stores as ONE launch (the three critic layers at the end of a D-step, the
// four generator layers at the end of a G-step): the products are independent of each other, so their fill / drain phases
// overlap each other's k-loops and the step pays one launch instead of three or four.  Blocks are dealt to the problems in
// order (largest first); every problem is a TN product with 16-byte aligned operands and more than 32 columns.
constexpr int MULTI_MAX = 4;
#include "colsum_body.h"
// cs_blocks > 0: the launch also carries the step's column sums (wfd_colsum_multi_rmsprop's work: bias / head gradients with
// their RMSprop step, loss scalars, the step's bookkeeping) in its FIRST cs_blocks workgroups, one 32-column block each:
// they read what the products read and write other memory, and they are done before the last tiles have started.
// rr_blocks / bn.bn_blocks > 0: likewise the RMSprop of the other vector ranges and the BatchNorm running statistics (the
// rest of wfd_g_end's work), in the workgroups behind the column sums.
struct MultiArgs { GemmArgs g[MULTI_MAX]; int nb[MULTI_MAX]; int gx[MULTI_MAX]; int n;
                   const long long* cs_desc; int cs_nseg, cs_blocks; ColsumOpt cs;
                   RmsRanges rr; int rr_blocks; BnRun bn; };

__global__ __launch_bounds__(NTHREADS) void gemm_f32_rms_multi_kernel(MultiArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE_F];
    if ((int)blockIdx.x < a.cs_blocks) {                                 // uniform over the workgroup
        colsum_multi_body256<true>(a.cs_desc, a.cs_nseg, a.cs, blockIdx.x, lds);
        return;
    }
    int id = blockIdx.x - a.cs_blocks, k = 0;
    if (id < a.rr_blocks) {
        rmsprop_ranges_body(a.cs.pbase, a.cs.gbase, a.cs.sqbase, a.rr, a.cs.lr, a.cs.alpha, a.cs.eps,
                            (long)id * NTHREADS + threadIdx.x);
        return;
    }
    id -= a.rr_blocks;
    if (id < 2 * a.bn.bn_blocks) {
        bn_running_layer_body(id / a.bn.bn_blocks ? a.bn.b : a.bn.a, (id % a.bn.bn_blocks) * NTHREADS + threadIdx.x, a.bn.count,
                              a.bn.momentum, a.bn.groups, a.bn.nupd_last);
        return;
    }
    id -= 2 * a.bn.bn_blocks;
    while (k < a.n - 1 && id >= a.nb[k]) { id -= a.nb[k]; ++k; }       // uniform over the workgroup
    gemm_f32_body<false, false, 4, 4, false, false, 2>(a.g[k], lds, id % a.gx[k], id / a.gx[k], 0);
}

// ---- the same launch as PERSISTENT workgroups (round 3; what the step runs) ---------------------------------------------
// gemm_f32_rms_multi_kernel above deals one 64 x 64 tile to every workgroup: at the critic's shapes that is 1408 workgroups
// against 768 resident slots, i.e. a full round of workgroups that are all in the same phase at the same time (all loading
// their first operand tile, then all in their k-loop, then all storing: the stamps of profiles/r02_rms_multi_stamps.txt)
// followed by a 37 %-full second round that pays the first-tile latency again, and every XCD's L2 fetches every operand
// panel.  Here `8 * nslots` workgroups (three per CU) stay resident and each walks a LIST of tiles as ONE flattened
// sequence of k-steps:
//   * the first operand tile of tile t+1 is requested during the LAST k-step of tile t and is already in LDS when tile
//     t's store begins, so a tile boundary costs no load latency, and tile t's parameter / square-average stores are in
//     flight while tile t+1's MFMAs run (the stores of a tile never sit at the end of the launch, except the last one's);
//   * XCD-aware tile lists: workgroups b and b + 8 share an XCD (round-robin dispatch; used for speed only, any placement
//     gives the same result), and group x = b % 8 owns a contiguous rows x columns SLAB of every problem's tile grid, so an
//     XCD's L2 fetches only its slab's operand panels (host picks the 8x1 / 4x2 / 2x4 / 1x8 split per problem that balances
//     the tile counts and then minimises panel fetches).
// Every tile is computed exactly as gemm_f32_body<false, false, 4, 4, false, false, 2> computes it (same k order, same
// MFMA sequence, same rmsprop_update), so the launch is bit-identical to the one-tile-per-workgroup form; which workgroup
// computes a tile cannot matter.  Exit: a workgroup leaves when its list index passes the group's tile count (static lists,
// no queue, no flag, no wait on another workgroup).
struct PersistArgs {
    MultiArgs m;
    int small_blocks, nslots;
    int gy[MULTI_MAX], lpr[MULTI_MAX];       // tile rows of problem k; its grid is split 2^lpr x 2^(3 - lpr) over the XCD groups
    unsigned long long* stamps;              // diagnostic build only (WFD_STAMPS): 16 s_memrealtime slots per workgroup
};
#ifdef WFD_STAMPS
#define WFD_PSTAMP(slot) if (a.stamps && threadIdx.x == 0 && (slot) < 16) a.stamps[(long)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define WFD_PSTAMP(slot)
#endif

// slab of XCD group x in a gy x gx tile grid split pr x pc (pr = 2^lpr, pc = 8 / pr): rows [rlo, rhi) x columns [clo, chi)
__host__ __device__ __forceinline__ void persist_slab(int gy, int gx, int lpr, int x, int& rlo, int& rhi, int& clo, int& chi) {
    const int lpc = 3 - lpr, xr = x >> lpc, xc = x & ((1 << lpc) - 1);
    rlo = (gy * xr) >> lpr; rhi = (gy * (xr + 1)) >> lpr;
    clo = (gx * xc) >> lpc; chi = (gx * (xc + 1)) >> lpc;
}

// RB = row blocks of 32 per wave: 1 -> 64 x 64 tiles, three workgroups per CU (any problem); 2 -> 128 x 64 tiles (each wave
// 64 x 32: two accumulators sharing every B fragment), two workgroups per CU -- half the operand bytes staged per MFMA, and
// a k-step's 32 MFMAs per wave cover the latency of the next operand tile's loads even while the operands are still cold in
// the XCD's L2 (all workgroups of a group walk k in step, so every k-step's first touch is a miss for all of them).
// GRAD: the data-parallel form of the same launch -- the tiles are STORED (P is then the gradient matrix; the ranks' gradients
// have to meet before RMSprop may see them), the column sums are stored without their optimizer step, and there are no
// RMSprop ranges.
template <int RB, int WPE, bool GRAD = false>
__global__ __launch_bounds__(NTHREADS, WPE) void gemm_f32_rms_persist_kernel(PersistArgs a) {
    constexpr int BMT = 64 * RB, NSUB = RB + 1, BUF_F = NSUB * TILE_F;       // A sub-tiles of 64 rows, then the B tile
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF_F];
    const int b = blockIdx.x;
    if (b < a.small_blocks) {                                           // uniform over the workgroup
        int id = b;
        if (id < a.m.cs_blocks) { colsum_multi_body256<!GRAD>(a.m.cs_desc, a.m.cs_nseg, a.m.cs, id, lds); return; }
        id -= a.m.cs_blocks;
        if constexpr (!GRAD) {
            if (id < a.m.rr_blocks) {
                rmsprop_ranges_body(a.m.cs.pbase, a.m.cs.gbase, a.m.cs.sqbase, a.m.rr, a.m.cs.lr, a.m.cs.alpha, a.m.cs.eps,
                                    (long)id * NTHREADS + threadIdx.x);
                return;
            }
            id -= a.m.rr_blocks;
        }
        bn_running_layer_body(id / a.m.bn.bn_blocks ? a.m.bn.b : a.m.bn.a, (id % a.m.bn.bn_blocks) * NTHREADS + threadIdx.x,
                              a.m.bn.count, a.m.bn.momentum, a.m.bn.groups, a.m.bn.nupd_last);
        return;
    }
    const int x = b & 7;                                                // workgroups b and b + 8 share an XCD
    const int slot = (b - (a.small_blocks + ((x - a.small_blocks) & 7))) >> 3;
    // tile qq of this group -> (problem, tile row, tile column): the group's slabs in problem order, row-major inside a
    // slab; false when the list is exhausted.  All scalar (uniform over the workgroup); the argument block is indexed
    // statically so that it stays in the kernarg segment.
    const int np = a.m.n;
    const int gy0 = a.gy[0], gy1 = a.gy[1], gy2 = a.gy[2], gy3 = a.gy[3];
    const int gx0 = a.m.gx[0], gx1 = a.m.gx[1], gx2 = a.m.gx[2], gx3 = a.m.gx[3];
    const int lp0 = a.lpr[0], lp1 = a.lpr[1], lp2 = a.lpr[2], lp3 = a.lpr[3];
    auto decode = [&](int qq, int& k, int& by, int& bx) -> bool {
        bool found = false;
        k = by = bx = 0;
        auto try_k = [&](int t, int gy, int gx, int lpr) {
            if (t < np && !found) {
                int rlo, rhi, clo, chi;
                persist_slab(gy, gx, lpr, x, rlo, rhi, clo, chi);
                const int w = chi - clo, c = (rhi - rlo) * w;
                if (qq < c) { found = true; k = t; by = rlo + qq / w; bx = clo + qq % w; }
                else qq -= c;
            }
        };
        try_k(0, gy0, gx0, lp0); try_k(1, gy1, gx1, lp1); try_k(2, gy2, gx2, lp2); try_k(3, gy3, gx3, lp3);
        return found;
    };
    int q = slot, k, by, bx;
    if (!decode(q, k, by, bx)) return;
    WFD_PSTAMP(0)
    int tno = 0;
    (void)tno;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    // the fields of the problem a tile belongs to: operands (until the tile's last loads are out) / parameters (until its
    // store).  Read through the kernarg segment pointer with a uniform dynamic index (scalar loads at a computed offset):
    // indexing the by-value argument itself with a runtime index makes the compiler copy all of it to private memory.
    struct Opnd { const float* A; const float* B; int M, N, K, lda, ldb; };
    struct Parm { float* P; float* sq; int M, N, ldp; };
    const PersistArgs* ap = (const PersistArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    auto opnd = [&](int kk) {
        const GemmArgs& s = ap->m.g[kk];
        return Opnd{s.A, s.B, s.M, s.N, s.K, s.lda, s.ldb};
    };
    auto parm = [&](int kk) {
        const GemmArgs& s = ap->m.g[kk];
        return Parm{s.P, s.sq, s.M, s.N, s.ldp};
    };
    Opnd g = opnd(k);
    Parm pm = parm(k);
    int m0 = by * BMT, n0 = bx * BN, nk = (g.K + BK - 1) / BK;
    int offa[RB][2], offb[2], lp[2];
    float ra[RB][8], rb[8], pp[RB][16], sv[RB][16];
    auto offsets = [&](const Opnd& gg, int mm0, int nn0) {
#pragma unroll
        for (int j = 0; j < RB; ++j) tile_offsets<false, false>(offa[j], lp, gg.lda, mm0 + 64 * j, gg.M, 0, tid, 0, 0, 0);
        tile_offsets<false, false>(offb, lp, gg.ldb, nn0, gg.N, 0, tid, 0, 0, 0);
    };
    auto load_both = [&](const Opnd& gg, int mm0, int nn0, int kt) {
        const int k0 = kt * BK;
        if (k0 + BK <= gg.K) {                                          // wave-uniform: interior k-tile
#pragma unroll
            for (int j = 0; j < RB; ++j) load_tile_fast<false, false>(ra[j], gg.A, offa[j], lp, kt * BK * gg.lda, k0, tid, 0, 0);
            load_tile_fast<false, false>(rb, gg.B, offb, lp, kt * BK * gg.ldb, k0, tid, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < RB; ++j)
                load_tile<false, 4, false>(ra[j], gg.A, gg.lda, mm0 + 64 * j, gg.M, k0, gg.K, tid, 0, 0, 0);
            load_tile<false, 4, false>(rb, gg.B, gg.ldb, nn0, gg.N, k0, gg.K, tid, 0, 0, 0);
        }
    };
    int cur = 0;
    auto park = [&](int buf) {          // the operand tile in registers -> LDS buffer `buf`
#pragma unroll
        for (int j = 0; j < RB; ++j) store_tile<false, 4>(ra[j], lds + buf * BUF_F + j * TILE_F, tid);
        store_tile<false, 4>(rb, lds + buf * BUF_F + RB * TILE_F, tid);
    };
    f32x16 acc[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    offsets(g, m0, n0);
    load_both(g, m0, n0, 0);
    park(0);
    __syncthreads();
    WFD_PSTAMP(1)
    // wave (wm, wn): RB = 1: rows wm * 32 of the one A sub-tile; RB = 2: both 32-row blocks of A sub-tile wm; columns wn * 32
    auto mfma_block = [&]() {
        const float* a_s = lds + cur * BUF_F + (RB == 2 ? wm * TILE_F : 0);
        const float* b_s = lds + cur * BUF_F + RB * TILE_F;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 bv = read_frag<false>(b_s, wn * 32, kk, i, h);
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const f32x4 av = read_frag<false>(a_s, RB == 2 ? j * 32 : wm * 32, kk, i, h);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[j], 0, 0, 0);
            }
        }
    };
    // first row of accumulator block j of this wave, inside the tile
    auto row0 = [&](int j) { return RB == 2 ? wm * 64 + j * 32 : wm * 32; };
    auto prefetch_params = [&]() {
        if constexpr (GRAD) return;
        // parameter / square-average tile: requested behind the LAST operand tile's loads (loads retire in order)
        const int pcol = min(n0 + wn * 32 + i, pm.N - 1);
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = min(m0 + row0(j) + (r & 3) + 8 * (r >> 2) + 4 * h, pm.M - 1);
                pp[j][r] = WFD_RMS_P_LOAD(pm.P + (long)prow * pm.ldp + pcol);
                sv[j][r] = WFD_RMS_S_LOAD(pm.sq + (long)prow * pm.ldp + pcol);
            }
    };
    for (;;) {
        // k-steps 0 .. nk-3: next operand tile of THIS output tile
        for (int kt = 0; kt + 2 < nk; ++kt) {
            load_both(g, m0, n0, kt + 1);
            mfma_block();
            park(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
        if (nk >= 2) {                  // k-step nk-2: the last operand tile, and behind it the parameter tile
            load_both(g, m0, n0, nk - 1);
            prefetch_params();
            mfma_block();
            park(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            prefetch_params();
        }
        // k-step nk-1: the NEXT output tile's first operand tile is requested here and parked in LDS behind the MFMAs
        int kn, byn, bxn;
        const bool has_next = decode(q + a.nslots, kn, byn, bxn);       // uniform over the workgroup
        const int m0n = byn * BMT, n0n = bxn * BN;
        if (has_next) {
            g = opnd(kn);
            offsets(g, m0n, n0n);
            load_both(g, m0n, n0n, 0);
        }
        mfma_block();
        if (has_next) park(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        WFD_PSTAMP(2 + 2 * tno)
        // the tile's store: RMSprop on the prefetched parameter tile (C/D map of the 32 x 32 accumulator as in gemm_f32_body)
        const int col = n0 + wn * 32 + i;
        if (col < pm.N) {
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + row0(j) + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < pm.M) {
                        if constexpr (GRAD) {
                            pm.P[(long)row * pm.ldp + col] = acc[j][r];
                        } else {
                            rmsprop_update(pp[j][r], sv[j][r], acc[j][r], a.m.g[0].lr, a.m.g[0].alpha, a.m.g[0].eps);
                            WFD_RMS_S_STORE(sv[j][r], pm.sq + (long)row * pm.ldp + col);
                            WFD_RMS_P_STORE(pp[j][r], pm.P + (long)row * pm.ldp + col);
                        }
                    }
                }
        }
        WFD_PSTAMP(3 + 2 * tno)
#ifdef WFD_STAMPS
        ++tno;
        if (!has_next) { __builtin_amdgcn_s_waitcnt(0); WFD_PSTAMP(15) }
#endif
        if (!has_next) break;
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        q += a.nslots; pm = parm(kn); m0 = m0n; n0 = n0n; nk = (g.K + BK - 1) / BK;
    }
}

template <bool AKC, bool BKC, bool IM2COL, bool NARROW, int EPI>
int launch_gemm2(const GemmArgs& g, int splits, bool va4, bool vb4, hipStream_t st) {
    dim3 grid(NARROW ? cdiv(g.N, 32) : cdiv(g.N, BN), cdiv(g.M, BM), splits), block(NTHREADS);
    int rgx = 0, rgy = 0, rgz = 0;
    // (the switch: same-session A/B.  WFD_GEMM_XCD_PAIR=0: plain (x, y, z) grid everywhere)
    static const int xcd_pair = getenv("WFD_GEMM_XCD_PAIR") ? atoi(getenv("WFD_GEMM_XCD_PAIR")) : 1;
    if (xcd_pair && grid.y >= 2 && grid.y <= 8 && (grid.x & 7) != 0) {
        rgx = (int)grid.x; rgy = (int)grid.y; rgz = (int)grid.z;
        grid = dim3((unsigned)(((rgx * rgz + 7) / 8) * 8 * rgy), 1, 1);
    }
    if (va4 && vb4) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, 4, 4, IM2COL, NARROW, EPI>), grid, block, 0, st, g, rgx, rgy, rgz);
    else if (va4) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, 4, 1, IM2COL, NARROW, EPI>), grid, block, 0, st, g, rgx, rgy, rgz);
    else if (vb4) hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, 1, 4, IM2COL, NARROW, EPI>), grid, block, 0, st, g, rgx, rgy, rgz);
    else hipLaunchKernelGGL((gemm_f32_kernel<AKC, BKC, 1, 1, IM2COL, NARROW, EPI>), grid, block, 0, st, g, rgx, rgy, rgz);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// k=8 "SAME" conv with Cin = 32 on a long sequence (the DF observer's first block at L = 1400: M = B*L rows, K = 256):
// the implicit-GEMM form above re-reads every activation row eight times (once per tap) from L2 and streams the
// weights per k-tile.  Here a workgroup keeps its 128 + 7 input rows AND the 32 x 256 weight tile in LDS
// (19.4 + 33.3 KB), so the K loop touches no global memory and has no barrier: the A fragment of tap t is the SAME
// LDS tile read t rows further down.  4 waves x (32 rows x 32 columns), 128 MFMAs each.  A workgroup walks `cpw`
// consecutive chunks with the weight tile kept in LDS and the next chunk's rows prefetched into registers.
// Rows are flattened (sample, position) pairs; a tap that would cross a sample edge is zeroed per lane.
constexpr int C32 = 32, C32_ROWS = 128, XS_LD = C32 + 4, WS_LD = 8 * C32 + 4;
constexpr int C32_LDS_FLOATS = (C32_ROWS + 7) * XS_LD + 32 * WS_LD;

template <bool EPI>
__global__ __launch_bounds__(NTHREADS) void conv8_c32_kernel(GemmArgs g, int cpw, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                              // [135][36]  rows m0 - pad .. m0 + 127 + 7 - pad
    float* ws = smem + (C32_ROWS + 7) * XS_LD;     // [32][260]  weights of columns n0 .. n0 + 31, k-contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * 32, pad = g.convPad, L = g.convL;
    // the weight tile is loaded once per workgroup and serves its `cpw` consecutive 128-row chunks
    for (int c = tid; c < 32 * 64; c += NTHREADS) {
        const int r = c >> 6, q = c & 63;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + r < g.N) v = *reinterpret_cast<const float4*>(g.B + (long)(n0 + r) * (8 * C32) + 4 * q);
        *reinterpret_cast<float4*>(ws + r * WS_LD + 4 * q) = v;
    }
    // input rows of a chunk travel global -> registers -> LDS; the next chunk's are in flight during the MFMA block
    constexpr int XN = (C32_ROWS + 7) * 8, XQ = (XN + NTHREADS - 1) / NTHREADS;
    float4 xr[XQ];
    // pooled forward: the row grid is shifted by -3, so that every wave's 32 rows are four whole pooling windows
    const int shift = (EPI && g.pool_out) ? 3 : 0;
    auto load_x = [&](int chunk) {
        const long base = (long)chunk * C32_ROWS - pad - shift;
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int c = tid + u * NTHREADS;
            const long src = base + (c >> 3);
            xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < XN && src >= 0 && src < g.M) xr[u] = *reinterpret_cast<const float4*>(g.A + src * C32 + 4 * (c & 7));
        }
    };
    const int c0 = blockIdx.x * cpw, c1 = min(c0 + cpw, nchunks);
    load_x(c0);
    for (int ch = c0; ch < c1; ++ch) {
        if (ch > c0) __syncthreads();              // every wave is done reading the previous chunk's rows
#pragma unroll
        for (int u = 0; u < XQ; ++u) {
            const int c = tid + u * NTHREADS;
            if (c < XN) *reinterpret_cast<float4*>(xs + (c >> 3) * XS_LD + 4 * (c & 7)) = xr[u];
        }
        __syncthreads();
        if (ch + 1 < c1) load_x(ch + 1);

        const int m0 = ch * C32_ROWS - shift;
        const int mrow = m0 + wave * 32 + i;       // the output row whose A values this lane feeds
        const int l = (mrow >= 0 && mrow < g.M) ? mrow % L : -(1 << 20);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // epi 2 (the input-gradient conv: out = acc * f(D)): the D tile is requested before the MFMA block instead of
        // behind it (-8 us per group of the B=64 step: the 16 dependent loads per lane no longer sit in front of the stores)
        float dv[16];
        // epi 1 (forward conv: out = act(acc * scale[n] + shift[n])): the column's scale / shift likewise
        const bool pf1 = EPI && g.epi == 1;
        float psc = 1.f, psh = 0.f;
        if (pf1) {
            const int pcol = min(n0 + i, g.N - 1);
            if (g.scale) psc = g.scale[pcol];
            if (g.shift) psh = g.shift[pcol];
        }
        const bool pfd = EPI && g.epi == 2 && g.dmode;
        if (pfd) {
            const int pcol = min(n0 + i, g.N - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int prow = max(min(m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1), 0);
                dv[r] = g.D[(long)prow * g.ldd + pcol];
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool ok = (unsigned)(l + t - pad) < (unsigned)L;
            const float* ap = xs + (wave * 32 + i + t) * XS_LD + 4 * h;
            const float* bp = ws + i * WS_LD + t * C32 + 4 * h;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * kk);
                const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 8 * kk);
                if (!ok) { a[0] = 0.f; a[1] = 0.f; a[2] = 0.f; a[3] = 0.f; }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], b[s2], acc, 0, 0, 0);
            }
        }
        const int col = n0 + i;
        float pv[16];                              // activated outputs as the pool sees them (pad / foreign rows: 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = 0.f;
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= 0 && row < g.M) {
                    if constexpr (EPI) {
                        if (pfd) {
                            float v = acc[r];
                            const float d = dv[r];
                            if (g.dmode == 1) v *= d;
                            else if (g.dmode == 2) v = d > 0.f ? v : 0.f;
                            else v = d > 0.f ? v : v * (d + 1.f);
                            if (g.scale) v *= g.scale[col];
                            g.P[(long)row * g.ldp + col] = v;
                        } else if (pf1) {
                            float v = g.scale ? __fmaf_rn(acc[r], psc, psh) : acc[r] + psh;          // = epi_apply, epi 1
                            if (g.act == 1) v = fmaxf(v, 0.f);
                            else if (g.act == 2) v = v > 0.f ? v : elu_expm1(v);
                            else if (g.act == 3) v = 1.0f / (1.0f + expf(-v));
                            g.P[(long)row * g.ldp + col] = v;
                            // the last three positions of a sample lie in no window of their own: in this tiling they are
                            // the left padding of the next sample's first window, which counts as 0
                            pv[r] = (row % L < L - 3) ? v : 0.f;
                        } else {
                            epi_store(g, g.P, acc[r], row, col);
                        }
                    } else {
                        g.P[(long)row * g.ldp + col] = acc[r];
                    }
                }
            }
        }
        if constexpr (EPI) {
            if (g.pool_out) {
                // window w of this wave = its rows 8w .. 8w + 7: registers 4w .. 4w + 3 of the lower (h = 0) and of the upper
                // half-wave; the comparison is pool8_fwd_kernel's (v > mx ? v : mx from -inf)
                const int nwin = g.M >> 3;         // = Bn * L / 8
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 4; ++q) mx = pv[4 * w + q] > mx ? pv[4 * w + q] : mx;
                    const float other = __shfl_xor(mx, 32, 64);
                    // rows 8w .. 8w + 3 come first in the window: lower half, then upper half
                    const float lo = h == 0 ? mx : other, hi = h == 0 ? other : mx;
                    const float res = hi > lo ? hi : lo;
                    const int wid = (m0 + wave * 32 + 8 * w + 3) >> 3;
                    if (h == 0 && col < g.N && wid < nwin) g.pool_out[(long)wid * g.N + col] = res;
                }
            }
        }
    }
}

// chunks per workgroup: three workgroups fit a CU (LDS) and the kernel needs all three to hide its LDS / store phases
// (measured at M = 5.7 M rows: 94 TF with 3 per CU, 83 TF with 2, 55 TF with 1), so the chunks are dealt to 768 workgroups
inline int conv8_c32_cpw(int nchunks, int gy) {
    const int slots = cdiv(768, gy);
    return cdiv(nchunks, slots);
}

template <bool EPI>
int launch_conv8_c32(const GemmArgs& g, hipStream_t st) {
    const size_t lds = (size_t)C32_LDS_FLOATS * sizeof(float);
    WFD_LDS_ATTR((conv8_c32_kernel<EPI>), lds);
    const int nchunks = cdiv(g.M + ((EPI && g.pool_out) ? 3 : 0), C32_ROWS), gy = cdiv(g.N, 32), cpw = conv8_c32_cpw(nchunks, gy);
    hipLaunchKernelGGL((conv8_c32_kernel<EPI>), dim3(cdiv(nchunks, cpw), gy), dim3(NTHREADS), lds, st, g, cpw, nchunks);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// the LDS-resident form pays when there are enough 128-row tiles to fill the chip
inline bool use_conv8_c32(int M, int Cin, int Cout) { return Cin == C32 && (long)cdiv(M, C32_ROWS) * cdiv(Cout, 32) >= 128; }

// ---------------------------------------------------------------------------------------------------------------
// Large-M NT product (bulk sampling, large-batch training): 128x128x32 tile, 2x2 waves of 64x64 -- four 32x32
// accumulators per wave, so every A / B fragment read from LDS feeds two MFMAs instead of one (half the LDS traffic per
// FLOP of the 64x64 tile above).  Same staging (global -> registers -> LDS, double buffered, [row][36] tiles), no K
// split (used only when the 128x128 tiles alone fill the chip), workgroup ids remapped so that one XCD walks
// neighbouring tiles.  Dynamic LDS: 2 x (128 + 128) x 36 floats = 73.7 KB -> two workgroups per CU.
constexpr int T128 = 128, T128_F = T128 * LDK;            // floats per operand tile

// PF2 (K % 32 == 0, >= 3 k-tiles): operand tiles are requested TWO k-tiles ahead into two register sets (the loop is unrolled
// by two so that each set has a fixed role per half) -- with two waves per SIMD one k-tile of MFMAs (1.7 us) does not always
// cover an HBM miss of the activation stream, two do; the wait in front of the LDS stores then only covers the OLDER set
// (vmcnt leaves the younger set's 8 loads in flight).
template <bool EPI, bool PF2 = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f32_nt128_kernel(GemmArgs g, int gx, int gy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int nb = gx * gy;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    const int m0 = (bid / gx) * T128, n0 = (bid % gx) * T128;
    const int nk = (g.K + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float ra[2][8], rb[2][8];
    int offa[2][2], offb[2][2], lpu[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        tile_offsets<true, false>(offa[q], lpu, g.lda, m0 + 64 * q, g.M, 0, tid, 0, 0, 0);
        tile_offsets<true, false>(offb[q], lpu, g.ldb, n0 + 64 * q, g.N, 0, tid, 0, 0, 0);
    }
    auto load_all = [&](int k0) {
        if (k0 + BK <= g.K) {           // interior tile: branch-free loads at clamped offsets
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                load_tile_fast<true, false>(ra[q], g.A, offa[q], lpu, k0, k0, tid, 0, 0);
                load_tile_fast<true, false>(rb[q], g.B, offb[q], lpu, k0, k0, tid, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                load_tile<true, 4, false>(ra[q], g.A, g.lda, m0 + 64 * q, g.M, k0, g.K, tid, 0, 0, 0);
                load_tile<true, 4, false>(rb[q], g.B, g.ldb, n0 + 64 * q, g.N, k0, g.K, tid, 0, 0, 0);
            }
        }
    };
    auto compute = [&](const float* a_s) __attribute__((always_inline)) {
        const float* b_s = a_s + T128_F;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = read_frag<true>(a_s, wm * 64 + 32 * t, kk, i, h);
                bf[t] = read_frag<true>(b_s, wn * 64 + 32 * t, kk, i, h);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s2], bf[b][s2], acc[a][b], 0, 0, 0);
        }
    };
    auto to_lds = [&](float (&xa)[2][8], float (&xb)[2][8], float* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            store_tile<true, 4>(xa[q], dst + 64 * q * LDK, tid);
            store_tile<true, 4>(xb[q], dst + T128_F + 64 * q * LDK, tid);
        }
    };
    if constexpr (PF2) {
        float rc[2][8], rd[2][8];                 // the second register set
        auto load_fast = [&](float (&xa)[2][8], float (&xb)[2][8], int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                load_tile_fast<true, false>(xa[q], g.A, offa[q], lpu, k0, k0, tid, 0, 0);
                load_tile_fast<true, false>(xb[q], g.B, offb[q], lpu, k0, k0, tid, 0, 0);
            }
        };
        load_fast(ra, rb, 0);
        to_lds(ra, rb, smem);
        load_fast(rc, rd, BK);                    // tile 1 (host: nk >= 3)
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; kt += 2) {
            // even half: tile kt from LDS; (rc, rd) hold tile kt + 1, in flight; tile kt + 2 goes to (ra, rb)
            if (kt + 2 < nk) load_fast(ra, rb, (kt + 2) * BK);
            __builtin_amdgcn_sched_barrier(0);
            compute(smem + cur * 2 * T128_F);
            if (kt + 1 < nk) to_lds(rc, rd, smem + (cur ^ 1) * 2 * T128_F);
            __syncthreads();
            cur ^= 1;
            if (kt + 1 >= nk) break;
            // odd half: tile kt + 1; (ra, rb) hold tile kt + 2; tile kt + 3 goes to (rc, rd)
            if (kt + 3 < nk) load_fast(rc, rd, (kt + 3) * BK);
            __builtin_amdgcn_sched_barrier(0);
            compute(smem + cur * 2 * T128_F);
            if (kt + 2 < nk) to_lds(ra, rb, smem + (cur ^ 1) * 2 * T128_F);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        load_all(0);
        to_lds(ra, rb, smem);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) load_all((kt + 1) * BK);
            compute(smem + cur * 2 * T128_F);
            if (more) to_lds(ra, rb, smem + (cur ^ 1) * 2 * T128_F);
            __syncthreads();
            cur ^= 1;
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + 32 * b + i;
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.M) continue;
                if constexpr (EPI) epi_store(g, g.P, acc[a][b][r], row, col);
                else g.P[(long)row * g.ldp + col] = acc[a][b][r];
            }
        }
}

template <bool EPI>
int launch_nt128(const GemmArgs& g, hipStream_t st) {
    const size_t lds = (size_t)4 * T128_F * sizeof(float);
    WFD_LDS_ATTR((gemm_f32_nt128_kernel<EPI>), lds);
    const int gx = cdiv(g.N, T128), gy = cdiv(g.M, T128);
    static const int pf2 = getenv("WFD_NT128_PF2") ? atoi(getenv("WFD_NT128_PF2")) : 1;        // (the switch: same-session A/B)
    if (pf2 && g.K % BK == 0 && g.K >= 3 * BK) {
        WFD_LDS_ATTR((gemm_f32_nt128_kernel<EPI, true>), lds);
        hipLaunchKernelGGL((gemm_f32_nt128_kernel<EPI, true>), dim3(gx * gy), dim3(NTHREADS), lds, st, g, gx, gy);
        WFD_CHECK_LAUNCH();
        return WFD_OK;
    }
    hipLaunchKernelGGL((gemm_f32_nt128_kernel<EPI>), dim3(gx * gy), dim3(NTHREADS), lds, st, g, gx, gy);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// the tall-tile form pays when its tiles alone give every CU work (no K split needed)
inline bool use_nt128(int M, int N, int K, bool va4, bool vb4) {
    if (!(va4 && vb4 && M >= 1024 && K >= 64)) return false;
    const long tiles = (long)cdiv(M, T128) * cdiv(N, T128);
    // two workgroups fit a CU: skip grids whose last wave of 512 would run mostly empty (N = 1401: 704 tiles)
    return tiles >= 192 && 5 * tiles >= 4 * 512L * cdiv(tiles, 512);
}

template <bool AKC, bool BKC, bool IM2COL>
int launch_gemm(const GemmArgs& g, int splits, bool va4, bool vb4, hipStream_t st) {
    if (g.narrow) {
        if (g.epi) return launch_gemm2<AKC, BKC, IM2COL, true, true>(g, splits, va4, vb4, st);
        return launch_gemm2<AKC, BKC, IM2COL, true, false>(g, splits, va4, vb4, st);
    }
    if (g.epi) return launch_gemm2<AKC, BKC, IM2COL, false, true>(g, splits, va4, vb4, st);
    return launch_gemm2<AKC, BKC, IM2COL, false, false>(g, splits, va4, vb4, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the kernels index operands with 32-bit element offsets
inline bool fits31(long rows, long ld) { return rows * ld < (1L << 31); }
inline bool operands_fit(int layout, int M, int N, int K, int lda, int ldb) {
    const long ra = layout == 2 ? K : M, rb = layout == 0 ? N : K;
    return fits31(ra, lda) && fits31(rb, ldb);
}

// operand mode of the contraction entry points: 0 = f32-input MFMA (exact f32 products, the parity path),
// 1 = bf16 operands, 2 = split-bf16 x3 (gemm_mx.hip).  Read when a launch is issued (or captured).
thread_local int g_gemm_mode = 0;      // per host thread: two threads driving plans of different modes do not race

}  // namespace

int wfd_mx_launch(const GemmArgs& g, int layout, int splits, int terms, hipStream_t st);     // gemm_mx.hip
bool wfd_mx_conv_direct(int M, int Cin, int Cout, int terms);

extern "C" {

#ifdef WFD_STAMPS
static unsigned long long* g_multi_stamps = nullptr;
static int g_multi_nslots = 0;
// diagnostic build: the next wfd_gemm_f32_rmsprop_multi launches write their stamps here (nslots x 4 u64 per problem;
// tools/rms_multi_stamps.py)
int wfd_rms_multi_set_stamps(unsigned long long* stamps, int nslots) { g_multi_stamps = stamps; g_multi_nslots = nslots; return 0; }
#endif

int wfd_set_gemm_mode(int mode) {
    if (mode < 0 || mode > 2) return WFD_ERR_ARG;
    g_gemm_mode = mode;
    return WFD_OK;
}

int wfd_get_gemm_mode(void) { return g_gemm_mode; }

// Number of K splits the library would pick for an (M,N,K) product: enough workgroups to cover the 256 CUs once or
// twice, never thinner than 2 k-tiles per split.  The caller sizes the slab workspace with it.
int wfd_gemm_pick_splits(int M, int N, int K) {
    const int tiles = cdiv(M, BM) * (N <= 32 ? cdiv(N, 32) : cdiv(N, BN));
    const int ktiles = cdiv(K, BK);
    // two workgroups per CU when the tiles alone cover a good part of the chip, one per CU below that: every extra split
    // is one more slab for the consumer to read back, and with few tiles the slabs outweigh the second workgroup
    // (measured on the config-2 step: 32 tiles x 8 splits beats x 16 by 2.5 % of the whole step, 64 tiles x 8 beats x 4)
    const int target = tiles >= 40 ? 512 : 256;
    int s = target / (tiles > 0 ? tiles : 1);
    if (s < 1) s = 1;
    if (s > 16) s = 16;
    while (s > 1 && ktiles / s < 2) --s;
    return s;
}

// The same for a given operand mode (wfd_set_gemm_mode): the bf16-pipe core (modes 1, 2) switches to 128x128 tiles once
// they fill the chip, so mid-sized products are split for THAT tile (about 512 workgroups, >= 2 of its k-tiles per split).
int wfd_gemm_pick_splits_mode(int M, int N, int K, int mode) {
    if (mode == 0) return wfd_gemm_pick_splits(M, N, K);
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (t128 >= 256) return 1;
    if (t128 >= 64) {
        // two 128x128 workgroups fit a CU: a "wave" is 512 of them.  Take the split count whose grid wastes the least
        // of its last wave (smallest on ties), with at least four 64-deep k-tiles per split.
        const int ktiles = cdiv(K, 64);
        int smax = ktiles / 4;
        if (smax > 16) smax = 16;
        int best = 1;
        double best_eff = 0.0;
        for (int s = cdiv(512, t128); s <= smax; ++s) {
            const double w = (double)t128 * s / 512.0;
            const double eff = w / (double)cdiv(t128 * s, 512);
            if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
        }
        return best;
    }
    // small products: the bf16 pipe is not the limit, slab traffic is -- at most 8 splits
    const int s = wfd_gemm_pick_splits(M, N, K);
    return s > 8 ? 8 : s;
}

// 1 when the fused conv entry points run a (Bn*L, Cin, Cout) conv with the LDS-resident kernel (no K split needed: the
// caller should take the fused single-launch path whatever wfd_gemm_pick_splits says)
int wfd_conv8_direct(int M, int Cin, int Cout) { return use_conv8_c32(M, Cin, Cout) ? 1 : 0; }
int wfd_conv8_direct_mode(int M, int Cin, int Cout, int mode) {
    if (mode == 0) return wfd_conv8_direct(M, Cin, Cout);
    return wfd_mx_conv_direct(M, Cin, Cout, mode) ? 1 : 0;
}

// layout: 0 = NT, 1 = NN, 2 = TN.  P receives `splits` slabs of M x ldp floats, slab_stride floats apart.
int wfd_gemm_f32(int layout, const float* A, int lda, const float* B, int ldb, float* P, int ldp, long slab_stride,
                 int M, int N, int K, int splits, void* stream) {
    if (!A || !B || !P || M <= 0 || N <= 0 || K <= 0 || splits < 1 || ldp < N) return WFD_ERR_ARG;
    if (splits > 1 && slab_stride < (long)M * ldp) return WFD_ERR_ARG;
    if (!operands_fit(layout, M, N, K, lda, ldb)) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.B = B; g.P = P; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldp = ldp;
    g.slab_stride = slab_stride;
    g.ktiles_per_split = cdiv(cdiv(K, BK), splits);
    g.narrow = N <= 32;
    hipStream_t st = (hipStream_t)stream;
    const bool va4 = aligned16(A) && (lda % 4 == 0), vb4 = aligned16(B) && (ldb % 4 == 0);
    if (g_gemm_mode && va4 && vb4 && layout >= 0 && layout <= 2) {
        if (layout == 2 ? (lda < M || ldb < N) : (lda < K || ldb < (layout == 0 ? K : N))) return WFD_ERR_ARG;
        return wfd_mx_launch(g, layout, splits, g_gemm_mode, st);
    }
    switch (layout) {
        case 0:
            if (lda < K || ldb < K) return WFD_ERR_ARG;
            if (splits == 1 && use_nt128(M, N, K, va4, vb4)) return launch_nt128<false>(g, st);
            return launch_gemm<true, true, false>(g, splits, va4, vb4, st);
        case 1: if (lda < K || ldb < N) return WFD_ERR_ARG; return launch_gemm<true, false, false>(g, splits, va4, vb4, st);
        case 2: if (lda < M || ldb < N) return WFD_ERR_ARG; return launch_gemm<false, false, false>(g, splits, va4, vb4, st);
        default: return WFD_ERR_ARG;
    }
}

// Two independent products in one launch: problem 1 = a TN weight gradient written in full to P1 (one split), problem 2
// = an NT (layout2 = 0) or NN (layout2 = 1) product into `splits2` slabs.  Both with 16-byte aligned operands and
// leading dimensions that are multiples of 4 floats; returns WFD_ERR_ARG otherwise (the caller launches them one by one).
int wfd_gemm_f32_pair(const float* A1, int lda1, const float* B1, int ldb1, float* P1, int ldp1, int M1, int N1, int K1,
                      int layout2, const float* A2, int lda2, const float* B2, int ldb2, float* P2, int ldp2,
                      long slab_stride2, int M2, int N2, int K2, int splits2, void* stream) {
    if (!A1 || !B1 || !P1 || !A2 || !B2 || !P2 || M1 <= 0 || N1 <= 0 || K1 <= 0 || M2 <= 0 || N2 <= 0 || K2 <= 0) return WFD_ERR_ARG;
    if (splits2 < 1 || ldp1 < N1 || ldp2 < N2 || (layout2 != 0 && layout2 != 1) || g_gemm_mode != 0) return WFD_ERR_ARG;
    if (splits2 > 1 && slab_stride2 < (long)M2 * ldp2) return WFD_ERR_ARG;
    if (lda1 < M1 || ldb1 < N1 || lda2 < K2 || ldb2 < (layout2 == 0 ? K2 : N2)) return WFD_ERR_ARG;
    if (N1 <= 32 || N2 <= 32) return WFD_ERR_ARG;
    const bool ok = aligned16(A1) && aligned16(B1) && aligned16(A2) && aligned16(B2) && lda1 % 4 == 0 && ldb1 % 4 == 0 &&
                    lda2 % 4 == 0 && ldb2 % 4 == 0;
    if (!ok || !operands_fit(2, M1, N1, K1, lda1, ldb1) || !operands_fit(layout2, M2, N2, K2, lda2, ldb2)) return WFD_ERR_ARG;
    GemmArgs g1{}, g2{};
    g1.A = A1; g1.B = B1; g1.P = P1; g1.M = M1; g1.N = N1; g1.K = K1; g1.lda = lda1; g1.ldb = ldb1; g1.ldp = ldp1;
    g1.ktiles_per_split = cdiv(K1, BK);
    g2.A = A2; g2.B = B2; g2.P = P2; g2.M = M2; g2.N = N2; g2.K = K2; g2.lda = lda2; g2.ldb = ldb2; g2.ldp = ldp2;
    g2.slab_stride = slab_stride2;
    g2.ktiles_per_split = cdiv(cdiv(K2, BK), splits2);
    PairDims d;
    d.gx1 = cdiv(N1, BN); d.gy1 = cdiv(M1, BM); d.nb1 = d.gx1 * d.gy1;
    d.gx2 = cdiv(N2, BN); d.gy2 = cdiv(M2, BM); d.nb2 = d.gx2 * d.gy2 * splits2;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(d.nb1 + d.nb2), block(NTHREADS);
    if (layout2 == 0) hipLaunchKernelGGL((gemm_f32_pair_kernel<false, false, true, true>), grid, block, 0, st, g1, g2, d);
    else hipLaunchKernelGGL((gemm_f32_pair_kernel<false, false, true, false>), grid, block, 0, st, g1, g2, d);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// "SAME" k=8 stride-1 conv as an implicit GEMM over a channels-last activation X[B][L][Cin]:
//   P[b*L + l][co] = sum_{t,ci} X[b][l + t - pad][ci] * Wp[co][t*Cin + ci]      (zero outside [0,L))
// pad = 3 for the forward conv (src/model.py:96-101: p = 7, left 3 / right 4) and 4 for its transposed
// (input-gradient) form with flipped taps.  Cin % 4 == 0 required (Cin = 1 uses the direct kernel in df.hip).
int wfd_conv8_gemm_f32(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, float* P, int ldp,
                       long slab_stride, int splits, void* stream) {
    if (!X || !Wp || !P || Bn <= 0 || L <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || ldp < Cout) return WFD_ERR_ARG;
    if (!aligned16(X) || !aligned16(Wp) || !fits31((long)Bn * L, Cin)) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = X; g.B = Wp; g.P = P; g.M = Bn * L; g.N = Cout; g.K = 8 * Cin; g.lda = Cin; g.ldb = 8 * Cin; g.ldp = ldp;
    g.slab_stride = slab_stride;
    g.ktiles_per_split = cdiv(cdiv(g.K, BK), splits);
    g.convL = L; g.convCin = Cin; g.convPad = pad;
    g.narrow = Cout <= 32;
    if (splits > 1 && slab_stride < (long)g.M * ldp) return WFD_ERR_ARG;
    if (g_gemm_mode) return wfd_mx_launch(g, 3, splits, g_gemm_mode, (hipStream_t)stream);
    if (splits == 1 && use_conv8_c32(g.M, Cin, Cout)) return launch_conv8_c32<false>(g, (hipStream_t)stream);
    return launch_gemm<true, true, true>(g, splits, true, true, (hipStream_t)stream);
}

// Same contraction with the consumer's element-wise math fused into the store (one split, no slab round trip):
//   epi 1: out = act(acc * scale[n] + shift[n])         (wfd_ep_affine_act)
//   epi 2: out = acc * f(D[m][n]) * scale[n]            (wfd_ep_mul without the second copy)
int wfd_gemm_f32_fused(int layout, const float* A, int lda, const float* B, int ldb, float* out, int ldo, int M, int N,
                       int K, int epi, const float* scale, const float* shift, int act, const float* D, int ldd,
                       int dmode, void* stream) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || ldo < N || epi < 1 || epi > 2) return WFD_ERR_ARG;
    if (epi == 2 && dmode && !D) return WFD_ERR_ARG;
    if (!operands_fit(layout, M, N, K, lda, ldb)) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.B = B; g.P = out; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldp = ldo;
    g.ktiles_per_split = cdiv(K, BK);
    g.narrow = N <= 32;
    g.epi = epi; g.act = act; g.dmode = dmode; g.ldd = ldd; g.scale = scale; g.shift = shift; g.D = D;
    hipStream_t st = (hipStream_t)stream;
    const bool va4 = aligned16(A) && (lda % 4 == 0), vb4 = aligned16(B) && (ldb % 4 == 0);
    if (g_gemm_mode && va4 && vb4 && layout >= 0 && layout <= 2) {
        if (layout == 2 ? (lda < M || ldb < N) : (lda < K || ldb < (layout == 0 ? K : N))) return WFD_ERR_ARG;
        return wfd_mx_launch(g, layout, 1, g_gemm_mode, st);
    }
    switch (layout) {
        case 0:
            if (lda < K || ldb < K) return WFD_ERR_ARG;
            if (use_nt128(M, N, K, va4, vb4)) return launch_nt128<true>(g, st);
            return launch_gemm<true, true, false>(g, 1, va4, vb4, st);
        case 1: if (lda < K || ldb < N) return WFD_ERR_ARG; return launch_gemm<true, false, false>(g, 1, va4, vb4, st);
        case 2: if (lda < M || ldb < N) return WFD_ERR_ARG; return launch_gemm<false, false, false>(g, 1, va4, vb4, st);
        default: return WFD_ERR_ARG;
    }
}

// A weight gradient that never reaches memory: g = the (M, N, K) product of the given layout (2 = TN for dW = dZ^T H), and in
// the store  sq = alpha sq + (1 - alpha) g^2;  P -= lr g / (sqrt(sq) + eps)  on the parameter matrix P[M][ldp] and its
// square average SQ (same layout) -- loss.backward() + optimizer.step() for one weight matrix
// (src/mlp_df_wgan_train.py:229-230, 273-275; torch.optim.RMSprop defaults).  Same bits as wfd_gemm_f32 into a gradient
// buffer followed by wfd_rmsprop.  Single-GPU path only (data parallel must all-reduce the gradient first).  One split:
// the caller uses it for contractions short enough to go unsplit (K = rows of the batch).
int wfd_gemm_f32_rmsprop(int layout, const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* P,
                         int ldp, float* SQ, float lr, float alpha, float eps, void* stream) {
    if (!A || !B || !P || !SQ || M <= 0 || N <= 0 || K <= 0 || ldp < N) return WFD_ERR_ARG;
    if (!operands_fit(layout, M, N, K, lda, ldb)) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.B = B; g.P = P; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldp = ldp;
    g.ktiles_per_split = cdiv(K, BK);
    g.narrow = N <= 32;
    g.epi = 4; g.sq = SQ; g.lr = lr; g.alpha = alpha; g.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const bool va4 = aligned16(A) && (lda % 4 == 0), vb4 = aligned16(B) && (ldb % 4 == 0);
    if (g_gemm_mode && va4 && vb4 && layout >= 0 && layout <= 2) {
        if (layout == 2 ? (lda < M || ldb < N) : (lda < K || ldb < (layout == 0 ? K : N))) return WFD_ERR_ARG;
        return wfd_mx_launch(g, layout, 1, g_gemm_mode, st);
    }
    if (layout == 2 && va4 && vb4 && !g.narrow) {        // the form the training step runs: tile prefetched during the k-loop
        if (lda < M || ldb < N) return WFD_ERR_ARG;
        return launch_gemm2<false, false, false, false, 2>(g, 1, true, true, st);
    }
    switch (layout) {
        case 0: if (lda < K || ldb < K) return WFD_ERR_ARG; return launch_gemm<true, true, false>(g, 1, va4, vb4, st);
        case 1: if (lda < K || ldb < N) return WFD_ERR_ARG; return launch_gemm<true, false, false>(g, 1, va4, vb4, st);
        case 2: if (lda < M || ldb < N) return WFD_ERR_ARG; return launch_gemm<false, false, false>(g, 1, va4, vb4, st);
        default: return WFD_ERR_ARG;
    }
}

// desc (HOST memory): n problems x 10 64-bit words [A, lda, B, ldb, M, N, K, P, ldp, SQ], each as wfd_gemm_f32_rmsprop with
// layout 2 (TN).  WFD_ERR_ARG when a problem does not fit the fast form (unaligned operands, N <= 32, n > 4): the caller
// then launches them one by one.
static int rms_multi_launch(const long long* desc, int n, float lr, float alpha, float eps, const long long* cs_desc,
                            int cs_nseg, int cs_blocks, const ColsumOpt& cs, void* stream, const RmsRanges* rr = nullptr,
                            int rr_blocks = 0, const BnRun* bn = nullptr, bool grad = false);

// The data-parallel form: the same products as ONE launch with their tiles STORED to the gradient matrices (desc words 7 / 8:
// the gradient matrix and its row stride; word 9 unused), optionally carrying wfd_colsum_multi(cs_desc, cs_nseg, cs_blocks)
// -- the step's column sums WITHOUT an optimizer step -- in its first workgroups.  WFD_ERR_ARG when the fast form does not
// apply (fewer than 8 tiles, unaligned operands, N <= 32): the caller launches the products one by one.
int wfd_gemm_f32_multi_grad(const long long* desc, int n, const long long* cs_desc, int cs_nseg, int cs_blocks, void* stream) {
    if (cs_desc ? (cs_nseg <= 0 || cs_blocks <= 0) : (cs_nseg != 0 || cs_blocks != 0)) return WFD_ERR_ARG;
    return rms_multi_launch(desc, n, 0.f, 0.f, 0.f, cs_desc, cs_nseg, cs_blocks, ColsumOpt{}, stream, nullptr, 0, nullptr, true);
}

int wfd_gemm_f32_rmsprop_multi(const long long* desc, int n, float lr, float alpha, float eps, void* stream) {
    return rms_multi_launch(desc, n, lr, alpha, eps, nullptr, 0, 0, ColsumOpt{}, stream);
}

// the same launch also carrying wfd_colsum_multi_rmsprop(cs_desc, cs_nseg, cs_blocks, gbase, pbase, sqbase, numel, lr, alpha,
// eps, state, inc0, inc1): the whole end of a critic step (weight gradients, bias / head gradients, every RMSprop update, loss
// scalars, bookkeeping) in one launch.  The column sums must not read what the products write (the parameter matrices).
int wfd_gemm_f32_rmsprop_multi_colsum(const long long* desc, int n, float lr, float alpha, float eps, const long long* cs_desc,
                                      int cs_nseg, int cs_blocks, const float* gbase, float* pbase, float* sqbase, long numel,
                                      unsigned long long* state, unsigned long long inc0, unsigned long long inc1,
                                      void* stream) {
    if (!cs_desc || cs_nseg <= 0 || cs_blocks <= 0 || !gbase || !pbase || !sqbase || numel <= 0) return WFD_ERR_ARG;
    const ColsumOpt cs{gbase, pbase, sqbase, numel, lr, alpha, eps, state, inc0, inc1};
    return rms_multi_launch(desc, n, lr, alpha, eps, cs_desc, cs_nseg, cs_blocks, cs, stream);
}

// ... and also wfd_g_end's other two parts (RMSprop on `ranges`, the BatchNorm running statistics; meanA = NULL leaves those
// out): the whole end of a generator step as ONE launch.
int wfd_gemm_f32_rmsprop_multi_end(const long long* desc, int n, float lr, float alpha, float eps, const long long* cs_desc,
                                   int cs_nseg, int cs_blocks, const float* gbase, float* pbase, float* sqbase, long numel,
                                   const long* ranges, int nranges, unsigned long long* state, unsigned long long inc0,
                                   unsigned long long inc1, const float* meanA, const float* m2A, int NA, float* rmeanA,
                                   float* rvarA, const float* meanB, const float* m2B, int NB, float* rmeanB, float* rvarB,
                                   float count, float momentum, int groups, int nupd_last, void* stream) {
    if (!cs_desc || cs_nseg <= 0 || cs_blocks <= 0 || !gbase || !pbase || !sqbase || numel <= 0) return WFD_ERR_ARG;
    if (!ranges || nranges < 1 || nranges > 8) return WFD_ERR_ARG;
    const ColsumOpt cs{gbase, pbase, sqbase, numel, lr, alpha, eps, state, inc0, inc1};
    RmsRanges r{};
    long total = 0;
    for (int k = 0; k < nranges; ++k) {
        r.beg[k] = ranges[2 * k]; r.end[k] = ranges[2 * k + 1];
        if (r.beg[k] < 0 || r.end[k] < r.beg[k] || r.end[k] > numel) return WFD_ERR_ARG;
        total += r.end[k] - r.beg[k];
    }
    r.n = nranges;
    if (total <= 0) return WFD_ERR_ARG;
    BnRun bn{};
    if (meanA) {
        if (!m2A || !rmeanA || !rvarA || !meanB || !m2B || !rmeanB || !rvarB || NA < 1 || NB < 1 || groups < 1 ||
            nupd_last < 0 || count <= 1.f)
            return WFD_ERR_ARG;
        bn.a = BnLayer{meanA, m2A, rmeanA, rvarA, NA};
        bn.b = BnLayer{meanB, m2B, rmeanB, rvarB, NB};
        bn.count = count; bn.momentum = momentum; bn.groups = groups; bn.nupd_last = nupd_last;
        bn.bn_blocks = cdiv(NA > NB ? NA : NB, NTHREADS);
    }
    return rms_multi_launch(desc, n, lr, alpha, eps, cs_desc, cs_nseg, cs_blocks, cs, stream, &r, cdiv(total, NTHREADS), &bn);
}

static int rms_multi_launch(const long long* desc, int n, float lr, float alpha, float eps, const long long* cs_desc,
                            int cs_nseg, int cs_blocks, const ColsumOpt& cs, void* stream, const RmsRanges* rr, int rr_blocks,
                            const BnRun* bn, bool grad) {
    // (runs the f32-operand kernel whatever wfd_set_gemm_mode says: weight gradients contract over the batch rows only
    // (K = 64 .. 512), where the bf16 pipe buys nothing, so every operand mode shares this launch)
    if (!desc || n < 1 || n > MULTI_MAX) return WFD_ERR_ARG;
    MultiArgs a{};
    a.cs_desc = cs_desc; a.cs_nseg = cs_nseg; a.cs_blocks = cs_blocks; a.cs = cs;
    if (rr) { a.rr = *rr; a.rr_blocks = rr_blocks; }
    if (bn) a.bn = *bn;
    const int small_blocks = cs_blocks + a.rr_blocks + 2 * a.bn.bn_blocks;
    a.n = n;
    long total = 0;
    for (int k = 0; k < n; ++k) {
        const long long* d = desc + 10 * k;
        const float* A = reinterpret_cast<const float*>(d[0]);
        const float* B = reinterpret_cast<const float*>(d[2]);
        float* P = reinterpret_cast<float*>(d[7]);
        float* SQ = reinterpret_cast<float*>(d[9]);
        const int lda = (int)d[1], ldb = (int)d[3], M = (int)d[4], N = (int)d[5], K = (int)d[6], ldp = (int)d[8];
        if (!A || !B || !P || (!SQ && !grad) || M <= 0 || N <= 32 || K <= 0 || ldp < N || lda < M || ldb < N) return WFD_ERR_ARG;
        if (!aligned16(A) || !aligned16(B) || (lda & 3) || (ldb & 3) || !operands_fit(2, M, N, K, lda, ldb)) return WFD_ERR_ARG;
        GemmArgs& g = a.g[k];
        g.A = A; g.B = B; g.P = P; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldp = ldp;
        g.ktiles_per_split = cdiv(K, BK);
        g.epi = 4; g.sq = SQ; g.lr = lr; g.alpha = alpha; g.eps = eps;
        a.gx[k] = cdiv(N, BN);
        a.nb[k] = a.gx[k] * cdiv(M, BM);
        total += a.nb[k];
    }
#ifdef WFD_STAMPS
    if (g_multi_stamps) for (int k = 0; k < n; ++k) {
        a.g[k].D = reinterpret_cast<const float*>(g_multi_stamps + (long)k * g_multi_nslots * 4); a.g[k].ldd = g_multi_nslots;
    }
#endif
    // (the switches: same-session A/B.  WFD_RMS_PERSIST=0: one tile per workgroup; WFD_RMS_RB: 1 = 64 x 64 tiles at three
    // workgroups per CU, 2 = 128 x 64 tiles at two per CU; WFD_RMS_SLOTS: persistent workgroups per XCD group)
    static const int persist = getenv("WFD_RMS_PERSIST") ? atoi(getenv("WFD_RMS_PERSIST")) : 1;
    static const int rb_env = getenv("WFD_RMS_RB") ? atoi(getenv("WFD_RMS_RB")) : 1;
    static const int slots_env = getenv("WFD_RMS_SLOTS") ? atoi(getenv("WFD_RMS_SLOTS")) : 0;
    if ((persist || grad) && total >= 8) {
        PersistArgs pa{};
        pa.m = a;
        pa.small_blocks = small_blocks;
#ifdef WFD_STAMPS
        pa.stamps = g_multi_stamps;
#endif
        const int rb = (rb_env == 2 && !grad) ? 2 : 1;
        long worst = 0;
        for (int k = 0; k < n; ++k) {
            // split of problem k's gy x gx tile grid over the 8 XCD groups as pr x pc slabs: least tiles on the fullest
            // group first, then fewest operand panels fetched per group (rows + columns of a slab)
            const int gx = a.gx[k], gy = cdiv(a.g[k].M, 64 * rb);
            int best_lpr = 3; long best_cost = -1;
            for (int lpr = 0; lpr <= 3; ++lpr) {
                const int pr = 1 << lpr, pc = 8 >> lpr;
                const long mx = (long)cdiv(gy, pr) * cdiv(gx, pc);
                const long cost = mx * 100000 + cdiv(gy, pr) * rb + cdiv(gx, pc);
                if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_lpr = lpr; }
            }
            pa.gy[k] = gy; pa.lpr[k] = best_lpr;
        }
        for (int k = n; k < MULTI_MAX; ++k) { pa.gy[k] = 1; pa.lpr[k] = 0; pa.m.gx[k] = 1; }
        for (int x = 0; x < 8; ++x) {
            long c = 0;
            for (int k = 0; k < n; ++k) {
                int rlo, rhi, clo, chi;
                persist_slab(pa.gy[k], pa.m.gx[k], pa.lpr[k], x, rlo, rhi, clo, chi);
                c += (long)(rhi - rlo) * (chi - clo);
            }
            if (c > worst) worst = c;
        }
        int nslots = slots_env >= 1 ? slots_env : (rb == 2 ? 64 : 96);
        if (nslots > worst) nslots = (int)worst;            // never more workgroups than the fullest group has tiles
        pa.nslots = nslots;
        const dim3 grid((unsigned)(small_blocks + 8 * nslots));
        hipStream_t st = (hipStream_t)stream;
        if (grad) hipLaunchKernelGGL((gemm_f32_rms_persist_kernel<1, 3, true>), grid, dim3(NTHREADS), 0, st, pa);
        else if (rb == 2) hipLaunchKernelGGL((gemm_f32_rms_persist_kernel<2, 2>), grid, dim3(NTHREADS), 0, st, pa);
        else hipLaunchKernelGGL((gemm_f32_rms_persist_kernel<1, 3>), grid, dim3(NTHREADS), 0, st, pa);
        WFD_CHECK_LAUNCH();
        return WFD_OK;
    }
    if (grad) return WFD_ERR_ARG;            // (the one-tile-per-workgroup form only exists with RMSprop in its store)
    hipLaunchKernelGGL(gemm_f32_rms_multi_kernel, dim3((unsigned)(total + small_blocks)), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_conv8_gemm_f32_fused(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, float* out,
                             int ldo, int epi, const float* scale, const float* shift, int act, const float* D, int ldd,
                             int dmode, void* stream) {
    if (!X || !Wp || !out || Bn <= 0 || L <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || ldo < Cout) return WFD_ERR_ARG;
    if (!aligned16(X) || !aligned16(Wp) || epi < 1 || epi > 2 || (epi == 2 && dmode && !D)) return WFD_ERR_ARG;
    if (!fits31((long)Bn * L, Cin)) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = X; g.B = Wp; g.P = out; g.M = Bn * L; g.N = Cout; g.K = 8 * Cin; g.lda = Cin; g.ldb = 8 * Cin; g.ldp = ldo;
    g.ktiles_per_split = cdiv(g.K, BK);
    g.convL = L; g.convCin = Cin; g.convPad = pad;
    g.narrow = Cout <= 32;
    g.epi = epi; g.act = act; g.dmode = dmode; g.ldd = ldd; g.scale = scale; g.shift = shift; g.D = D;
    if (g_gemm_mode) return wfd_mx_launch(g, 3, 1, g_gemm_mode, (hipStream_t)stream);
    if (use_conv8_c32(g.M, Cin, Cout)) return launch_conv8_c32<true>(g, (hipStream_t)stream);
    return launch_gemm<true, true, true>(g, 1, true, true, (hipStream_t)stream);
}

// wfd_conv8_gemm_f32_fused (epi 1) whose launch also writes the max-pool of its output (wfd_df_pool8_fwd's values) to PO:
// only the LDS-resident Cin = 32 form on the f32 pipe with L % 8 == 0 and pad = 3 can (WFD_ERR_ARG otherwise: the caller then
// issues the two launches).
int wfd_conv8_gemm_f32_fused_pool(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, float* out,
                                  int ldo, const float* scale, const float* shift, int act, float* PO, void* stream) {
    if (!X || !Wp || !out || !PO || Bn <= 0 || L <= 0 || (L & 7) || Cin != C32 || Cout <= 0 || ldo < Cout || pad != 3)
        return WFD_ERR_ARG;
    if (!aligned16(X) || !aligned16(Wp) || !fits31((long)Bn * L, Cin) || g_gemm_mode != 0) return WFD_ERR_ARG;
    if (!use_conv8_c32(Bn * L, Cin, Cout)) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = X; g.B = Wp; g.P = out; g.M = Bn * L; g.N = Cout; g.K = 8 * Cin; g.lda = Cin; g.ldb = 8 * Cin; g.ldp = ldo;
    g.ktiles_per_split = cdiv(g.K, BK);
    g.convL = L; g.convCin = Cin; g.convPad = pad;
    g.narrow = Cout <= 32;
    g.epi = 1; g.act = act; g.scale = scale; g.shift = shift;
    g.pool_out = PO;
    return launch_conv8_c32<true>(g, (hipStream_t)stream);
}

// Weight gradient of the k=8 "SAME" conv (DF observer training, src/train_df.py:121-131 backward):
//   P[co][t*Cin + ci] = sum_{b,l} dY[b*L + l][co] * X[b][l + t - 3][ci]        (zero outside [0,L))
// a TN product whose B operand is the channels-last activation read as its im2col matrix; the contraction runs over
// all Bn*L rows, so it is split over gridDim.z into slabs (Cout x 8*Cin each).
int wfd_conv8_wgrad_f32(const float* dY, const float* X, int Bn, int L, int Cin, int Cout, float* P, int ldp,
                        long slab_stride, int splits, void* stream) {
    if (!dY || !X || !P || Bn <= 0 || L <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || (Cout & 3) || splits < 1)
        return WFD_ERR_ARG;
    if (!aligned16(dY) || !aligned16(X) || ldp < 8 * Cin) return WFD_ERR_ARG;
    if (!fits31((long)Bn * L, Cin) || !fits31((long)Bn * L, Cout)) return WFD_ERR_ARG;
    if (splits > 1 && slab_stride < (long)Cout * ldp) return WFD_ERR_ARG;
    GemmArgs g{};
    g.A = dY; g.B = X; g.P = P; g.M = Cout; g.N = 8 * Cin; g.K = Bn * L; g.lda = Cout; g.ldb = 8 * Cin; g.ldp = ldp;
    g.slab_stride = slab_stride;
    g.ktiles_per_split = cdiv(cdiv(g.K, BK), splits);
    g.convL = L; g.convCin = Cin; g.convPad = 3;
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), splits), block(NTHREADS);
    hipLaunchKernelGGL((gemm_f32_kernel<false, false, 4, 4, true, false, false>), grid, block, 0, (hipStream_t)stream, g, 0, 0, 0);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

}  // extern "C"

#ifdef WFD_STAMPS
extern "C" int wfd_gemm_f32_stamped(int layout, const float* A, int lda, const float* B, int ldb, float* P, int ldp,
                                    long slab_stride, int M, int N, int K, int splits, unsigned long long* stamps, int nwg,
                                    void* stream) {
    GemmArgs g{};
    g.A = A; g.B = B; g.P = P; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldp = ldp;
    g.slab_stride = slab_stride;
    g.ktiles_per_split = cdiv(cdiv(K, BK), splits);
    g.D = reinterpret_cast<const float*>(stamps); g.ldd = nwg;
    hipStream_t st = (hipStream_t)stream;
    switch (layout) {
        case 0: return launch_gemm<true, true, false>(g, splits, true, true, st);
        case 1: return launch_gemm<true, false, false>(g, splits, true, true, st);
        case 2: return launch_gemm<false, false, false>(g, splits, true, true, st);
        default: return WFD_ERR_ARG;
    }
}
#endif
