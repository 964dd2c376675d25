// Data-side kernels of the wfd-gan hot path (gfx950): batch assembly, device RNG (fast mode), burst clipping,
// and the sampler's inverse-scale / round-half-even / trim epilogue (src/generate.py:59-63).  All HBM-bound.
#include "common.h"

namespace {

// ---- Philox4x32-10 (counter-based; one 128-bit block per 4 outputs) ---------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// dropout keep-masks with values {0, 1/(1-p)} (what nn.Dropout multiplies by); n multiple of 4 handled generally
__global__ void rng_dropout_mask_kernel(float* __restrict__ out, long n, float p, unsigned long long seed,
                                        const unsigned long long* __restrict__ state, unsigned long long base) {
    const unsigned long long offset = (state ? state[0] : 0ull) + base;
    const float keep = 1.0f / (1.0f - p);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += stride) {
        const unsigned long long c = offset + (unsigned long long)q;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6d61736bu, 0u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const float v[4] = {u01(r.x) >= p ? keep : 0.f, u01(r.y) >= p ? keep : 0.f, u01(r.z) >= p ? keep : 0.f,
                            u01(r.w) >= p ? keep : 0.f};
        for (int k = 0; k < 4; ++k)
            if (q * 4 + k < n) out[q * 4 + k] = v[k];
    }
}

// the same for up to 8 mask buffers in one launch (the masks of every iteration of a group of n_critic iterations):
// blockIdx.y = slot; slot s draws with its own seed at counter state[0] + base[s]
struct MaskSlots {
    float* out[8];
    long n[8];
    unsigned long long seed[8];
    unsigned long long base[8];
};
__device__ __forceinline__ void mask_slot_body(const MaskSlots& a, int s, long bx, long nbx, float p,
                                               const unsigned long long* __restrict__ state) {
    float* __restrict__ out = a.out[s];
    const long n = a.n[s];
    const unsigned long long seed = a.seed[s], offset = (state ? state[0] : 0ull) + a.base[s];
    const float keep = 1.0f / (1.0f - p);
    const long stride = nbx * blockDim.x;
    for (long q = bx * blockDim.x + threadIdx.x; q * 4 < n; q += stride) {
        const unsigned long long c = offset + (unsigned long long)q;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6d61736bu, 0u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const float v[4] = {u01(r.x) >= p ? keep : 0.f, u01(r.y) >= p ? keep : 0.f, u01(r.z) >= p ? keep : 0.f,
                            u01(r.w) >= p ? keep : 0.f};
        for (int k = 0; k < 4; ++k)
            if (q * 4 + k < n) out[q * 4 + k] = v[k];
    }
}
__global__ void rng_dropout_mask_slots_kernel(MaskSlots a, float p, const unsigned long long* __restrict__ state) {
    mask_slot_body(a, blockIdx.y, blockIdx.x, gridDim.x, p, state);
}

// standard normal into rows of a strided matrix: out[m][0..Z) (Box-Muller on Philox uniforms)
__device__ __forceinline__ void normal_rows_body(float* __restrict__ out, int ld, int M, int Z, unsigned long long seed,
                                                 unsigned long long offset, long bx, long nbx) {
    const long total4 = ((long)M * Z + 3) / 4;
    const long stride = nbx * blockDim.x;
    for (long q = bx * blockDim.x + threadIdx.x; q < total4; q += stride) {
        const unsigned long long c = offset + (unsigned long long)q;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6e6f726du, 0u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
        sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
        const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
        for (int k = 0; k < 4; ++k) {
            const long e = q * 4 + k;
            if (e < (long)M * Z) out[(e / Z) * ld + (e % Z)] = v[k];
        }
    }
}
__global__ void rng_normal_rows_kernel(float* __restrict__ out, int ld, int M, int Z, unsigned long long seed,
                                       const unsigned long long* __restrict__ state, unsigned long long base) {
    normal_rows_body(out, ld, M, Z, seed, (state ? state[0] : 0ull) + base, blockIdx.x, gridDim.x);
}

// Assemble one training batch on device (replaces the DataLoader + np.eye one-hot of
// src/mlp_df_wgan_train.py:184-186):  for b < B:
//   real[b][0..S) = X[idx[b]][0..S);  real[b][S..S+C) = onehot(y[idx[b]]);  fake[b][S..S+C) = same one-hot
//   zin[b][Z..Z+C) = same one-hot;  labels_out[b] = y[idx[b]]
struct GatherArgs {
    const float* X; int ldX; const int* y; const long long* idx; const unsigned long long* state;
    long n_idx, n_rows, row_offset; int S, C; float* real; float* fake; int ldd; float* zin; int ldz, Z;
    int* labels_out; int rows_per_slot; long slot_stride, slot_idx_stride;
};
__device__ __forceinline__ void gather_row_body(const GatherArgs& a, int b) {
    float* __restrict__ real = a.real;
    float* __restrict__ fake = a.fake;
    float* __restrict__ zin = a.zin;
    int* __restrict__ labels_out = a.labels_out;
    long pos = a.row_offset + b;
    if (a.rows_per_slot > 0) {
        // stacked batches of several iterations: slot s = b / rows_per_slot reads the permutation slot_idx_stride
        // entries further on (= the rows all ranks consume per iteration) and writes its own real / fake buffers;
        // the generator input and the labels are stacked contiguously
        const int slot = b / a.rows_per_slot, r = b - slot * a.rows_per_slot;
        pos = a.row_offset + (long)slot * a.slot_idx_stride + r;
        real += (long)slot * a.slot_stride;
        if (fake) fake += (long)slot * a.slot_stride;
        if (zin) zin += (long)slot * a.rows_per_slot * a.ldz;
        if (labels_out) labels_out += slot * a.rows_per_slot;
        b = r;
    }
    // the cursor wraps inside the permutation and the row index is clamped: a stale cursor can never make
    // this kernel read outside the dataset
    long long src = a.idx[((a.state ? (long long)a.state[1] : 0ll) + pos) % a.n_idx];
    src = src < 0 ? 0 : (src >= a.n_rows ? a.n_rows - 1 : src);
    const int lab = a.y[src];
    const int S = a.S, C = a.C, ldd = a.ldd;
    for (int j = threadIdx.x; j < S; j += blockDim.x) real[(long)b * ldd + j] = a.X[src * a.ldX + j];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float v = c == lab ? 1.f : 0.f;
        real[(long)b * ldd + S + c] = v;
        if (fake) fake[(long)b * ldd + S + c] = v;
        if (zin) zin[(long)b * a.ldz + a.Z + c] = v;
    }
    if (threadIdx.x == 0 && labels_out) labels_out[b] = lab;
}
__global__ void gather_batch_kernel(GatherArgs a) { gather_row_body(a, blockIdx.x); }

// Everything a group of n_critic iterations draws before its first product, in ONE launch: the dropout keep-masks of its
// critic passes and of its G-step (the last mblocks * nslots blocks), the gathered real batches with their one-hot columns
// (the first gblocks blocks, one per row), and the generator's normal inputs (the next zblocks).  The three parts write
// disjoint memory; each is the body of the kernel it replaces.
struct GroupInputs {
    MaskSlots m; int nslots, mblocks; float p; const unsigned long long* mstate;
    GatherArgs g; int gblocks;
    unsigned long long zseed; int zblocks, zrows;
};
__global__ void group_inputs_kernel(GroupInputs a) {
    // the gathered rows first: each is a chain of four dependent loads (cursor -> permutation -> label / data row), which then
    // runs under the mask generation instead of behind it
    int b = blockIdx.x;
    if (b < a.gblocks) { gather_row_body(a.g, b); return; }
    b -= a.gblocks;
    if (b < a.zblocks) {
        normal_rows_body(a.g.zin, a.g.ldz, a.zrows, a.g.Z, a.zseed, a.g.state ? a.g.state[0] : 0ull, b, a.zblocks);
        return;
    }
    b -= a.zblocks;
    mask_slot_body(a.m, b / a.mblocks, b % a.mblocks, a.mblocks, a.p, a.mstate);
}

// z rows and the dropout keep-masks of one step in a single launch (same streams as the two kernels above:
// blocks [0, zblocks) draw the normals, the rest the masks)
__global__ void rng_step_inputs_kernel(float* __restrict__ zout, int ld, int M, int Z, unsigned long long zseed,
                                       float* __restrict__ mout, long nmask, float p, unsigned long long mseed,
                                       const unsigned long long* __restrict__ state, int zblocks) {
    const unsigned long long offset = state ? state[0] : 0ull;
    if ((int)blockIdx.x < zblocks) {
        const long total4 = ((long)M * Z + 3) / 4;
        for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (long)zblocks * blockDim.x) {
            const unsigned long long c = offset + (unsigned long long)q;
            const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6e6f726du, 0u),
                                          make_uint2((uint32_t)zseed, (uint32_t)(zseed >> 32)));
            const float r0 = sqrtf(-2.f * logf(u01(r.x))), r1 = sqrtf(-2.f * logf(u01(r.z)));
            float s0, c0, s1, c1;
            sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
            sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
            const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
            for (int k = 0; k < 4; ++k) {
                const long e = q * 4 + k;
                if (e < (long)M * Z) zout[(e / Z) * ld + (e % Z)] = v[k];
            }
        }
    } else {
        const float keep = 1.0f / (1.0f - p);
        const long nb = (long)gridDim.x - zblocks;
        for (long q = ((long)blockIdx.x - zblocks) * blockDim.x + threadIdx.x; q * 4 < nmask; q += nb * blockDim.x) {
            const unsigned long long c = offset + (unsigned long long)q;
            const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6d61736bu, 0u),
                                          make_uint2((uint32_t)mseed, (uint32_t)(mseed >> 32)));
            const float v[4] = {u01(r.x) >= p ? keep : 0.f, u01(r.y) >= p ? keep : 0.f, u01(r.z) >= p ? keep : 0.f,
                                u01(r.w) >= p ? keep : 0.f};
            for (int k = 0; k < 4; ++k)
                if (q * 4 + k < nmask) mout[q * 4 + k] = v[k];
        }
    }
}

// U[0,1) row weights (np.random.random((B,1)), src/utils.py:67)
__global__ void rng_uniform_kernel(float* __restrict__ out, long n, unsigned long long seed,
                                   const unsigned long long* __restrict__ state, unsigned long long base) {
    const unsigned long long offset = (state ? state[0] : 0ull) + base;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += stride) {
        const unsigned long long c = offset + (unsigned long long)q;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x756e6966u, 0u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const float v[4] = {(r.x >> 8) * (1.0f / 16777216.0f), (r.y >> 8) * (1.0f / 16777216.0f),
                            (r.z >> 8) * (1.0f / 16777216.0f), (r.w >> 8) * (1.0f / 16777216.0f)};
        for (int k = 0; k < 4; ++k)
            if (q * 4 + k < n) out[q * 4 + k] = v[k];
    }
}

// WGAN-GP interpolates (src/utils.py:69): out[b][j] = a_b * real[b][j] + ((1 - a_b) * fake[b][j]) for j < S,
// class columns j in [S, W) copied from real.  Separate roundings, as the tensor expression evaluates.
__global__ void interpolate_rows_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                        const float* __restrict__ alpha, float* __restrict__ out, int S, int W,
                                        int ld) {
    const int b = blockIdx.x;
    const float a = alpha[b], na = __fsub_rn(1.0f, a);
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        const long o = (long)b * ld + j;
        out[o] = j < S ? __fadd_rn(__fmul_rn(a, real[o]), __fmul_rn(na, fake[o])) : real[o];
    }
}

// per-iteration device state: state[0] = Philox counter offset, state[1] = cursor into the epoch permutation
__global__ void iter_advance_kernel(unsigned long long* __restrict__ state, unsigned long long inc0,
                                    unsigned long long inc1) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[0] += inc0; state[1] += inc1; }
}

// one-hot class columns for the sampler: zin[m][Z..Z+C) = onehot(labels[m])
__global__ void onehot_rows_kernel(const int* __restrict__ labels, int M, int C, float* __restrict__ out, int ld,
                                   int col0) {
    const int m = blockIdx.x;
    const int lab = labels[m];
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(long)m * ld + col0 + c] = c == lab ? 1.f : 0.f;
}

// --clip preprocessing (src/mlp_df_wgan_train.py:82-91) on the float64 feature matrix, in place
__global__ void clip_bursts_kernel(double* __restrict__ X, long N, int S, double max_len, double max_out, double max_in) {
    const long total = N * S;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int j = (int)(o % S);
        const double cap = j == 0 ? max_len : ((j & 1) ? max_out : max_in);
        const double v = X[o];
        if (v > cap) X[o] = cap;
    }
}

// Sampler epilogue for one generated row y (fp32, tail already zeroed by the length mask):
//   x = float(double(y) - min_)  ;  x = float(double(x) / scale_)      MinMaxScaler.inverse_transform on fp32
//   lo/hi = first/last non-zero of x (np.trim_zeros, both ends); out = rint(x[lo..hi]) as int, 0 -> 1
// out row is zero-filled past the length.  One workgroup per row.
__global__ void sample_post_kernel(const float* __restrict__ Y, int ldy, int S, const double* __restrict__ min_,
                                   const double* __restrict__ scale_, int* __restrict__ out, int ldo,
                                   int* __restrict__ len_out) {
    extern __shared__ float xs[];
    __shared__ int s_lo, s_hi;
    const int m = blockIdx.x;
    if (threadIdx.x == 0) { s_lo = S; s_hi = -1; }
    __syncthreads();
    int lo = S, hi = -1;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        float x = (float)((double)Y[(long)m * ldy + j] - min_[j]);
        x = (float)((double)x / scale_[j]);
        xs[j] = x;
        if (x != 0.f) { lo = min(lo, j); hi = max(hi, j); }
    }
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    lo = s_lo; hi = s_hi;
    const int n = hi >= lo ? hi - lo + 1 : 0;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        int v = 0;
        if (j < n) {
            v = (int)rintf(xs[lo + j]);
            if (v == 0) v = 1;
        }
        __builtin_nontemporal_store(v, out + (long)m * ldo + j);      // never read again on the device: streams past the caches
    }
    if (threadIdx.x == 0) len_out[m] = n;
}


// The same with 16-byte loads of the fp32 row (ldy % 4 == 0, Y 16-byte aligned): the bulk sampler's form.
__global__ void sample_post_v4_kernel(const float* __restrict__ Y, int ldy, int S, const double* __restrict__ min_,
                                      const double* __restrict__ scale_, int* __restrict__ out, int ldo,
                                      int* __restrict__ len_out) {
    extern __shared__ float xs[];
    __shared__ int s_lo, s_hi;
    const int m = blockIdx.x;
    if (threadIdx.x == 0) { s_lo = S; s_hi = -1; }
    __syncthreads();
    int lo = S, hi = -1;
    const float* __restrict__ yrow = Y + (long)m * ldy;
    const int S4 = S >> 2;
    for (int q = threadIdx.x; q < S4; q += blockDim.x) {
        const int j = 4 * q;
        const float4 y = *reinterpret_cast<const float4*>(yrow + j);
        const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = (float)((double)yy[k] - min_[j + k]);
            x = (float)((double)x / scale_[j + k]);
            xs[j + k] = x;
            if (x != 0.f) { lo = min(lo, j + k); hi = max(hi, j + k); }
        }
    }
    for (int j = 4 * S4 + threadIdx.x; j < S; j += blockDim.x) {
        float x = (float)((double)yrow[j] - min_[j]);
        x = (float)((double)x / scale_[j]);
        xs[j] = x;
        if (x != 0.f) { lo = min(lo, j); hi = max(hi, j); }
    }
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    lo = s_lo; hi = s_hi;
    const int n = hi >= lo ? hi - lo + 1 : 0;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        int v = 0;
        if (j < n) {
            v = (int)rintf(xs[lo + j]);
            if (v == 0) v = 1;
        }
        __builtin_nontemporal_store(v, out + (long)m * ldo + j);      // never read again on the device: streams past the caches
    }
    if (threadIdx.x == 0) len_out[m] = n;
}

// The sampler epilogues of the generator's other consumers, selected by `mode` (x = inverse_transform(y) as above):
//   0 generate.py:59-63      trim zeros of x at both ends, rint, 0 -> 1                      (= sample_post_kernel)
//   1 gan_wt.py:79-87        len = min(int(x0), S-1) made even; out = rint(x[1 .. 1+len)), 0 -> 1
//   2 gan_glue.py:114-124    len = int(x0); v = trunc(x[1 .. 1+len)); trim trailing zeros of v; odd length drops the
//                            last; v <= 0 -> 1; signs alternate +,-
//   3 synthesize.py:86-90    v = rint(x) over the whole row; trim zeros of v at both ends (interior zeros stay)
// A negative burst count (outside the reference's domain: counts are >= 0) is treated as 0.
__global__ void sample_post_mode_kernel(const float* __restrict__ Y, int ldy, int S, const double* __restrict__ min_,
                                        const double* __restrict__ scale_, int mode, int* __restrict__ out, int ldo,
                                        int* __restrict__ len_out) {
    extern __shared__ float xs[];
    __shared__ int s_lo, s_hi;
    const int m = blockIdx.x;
    if (threadIdx.x == 0) { s_lo = S; s_hi = -1; }
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        float x = (float)((double)Y[(long)m * ldy + j] - min_[j]);
        xs[j] = (float)((double)x / scale_[j]);
    }
    __syncthreads();
    int cnt = (int)xs[0];                       // int(x0): truncation
    cnt = cnt < 0 ? 0 : (cnt > S - 1 ? S - 1 : cnt);
    // candidate window [a, b) and the non-zero test that drives the trimming
    const int a = (mode == 1 || mode == 2) ? 1 : 0;
    const int b = (mode == 1 || mode == 2) ? 1 + cnt : S;
    int lo = S, hi = -1;
    if (mode != 1)
        for (int j = a + threadIdx.x; j < b; j += blockDim.x) {
            const float x = xs[j];
            const bool nz = mode == 0 ? (x != 0.f) : (mode == 2 ? ((int)x != 0) : ((int)rintf(x) != 0));
            if (nz) { lo = min(lo, j); hi = max(hi, j); }
        }
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    lo = s_lo; hi = s_hi;
    int first, n;
    if (mode == 1) { first = 1; n = cnt & ~1; }
    else if (mode == 2) { first = 1; n = hi >= 1 ? hi : 0; n &= ~1; }
    else { first = lo; n = hi >= lo ? hi - lo + 1 : 0; }
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        int v = 0;
        if (j < n) {
            const float x = xs[first + j];
            if (mode == 2) {
                v = (int)x;
                if (v <= 0) v = 1;
                if (j & 1) v = -v;
            } else {
                v = (int)rintf(x);
                if (mode != 3 && v == 0) v = 1;
            }
        }
        __builtin_nontemporal_store(v, out + (long)m * ldo + j);      // never read again on the device: streams past the caches
    }
    if (threadIdx.x == 0) len_out[m] = n;
}


// MinMaxScaler.fit on the float64 feature matrix (src/mlp_df_wgan_train.py:132-133): per-column min / max.
// Stage 1: a block of 64 columns x 16 row groups reduces a chunk of rows; stage 2 folds the chunks.  min / max are
// exact in any order, so the result is the bit pattern sklearn computes.
__global__ void colminmax_partial_kernel(const double* __restrict__ X, long N, int S, long rows_per_chunk,
                                         double* __restrict__ part) {
    __shared__ double smn[16][64], smx[16][64];
    const int c = blockIdx.x * 64 + threadIdx.x, g = threadIdx.y;
    const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    double mn = INFINITY, mx = -INFINITY;
    if (c < S)
        for (long r = r0 + g; r < r1; r += 16) {
            const double v = X[r * S + c];
            mn = fmin(mn, v);
            mx = fmax(mx, v);
        }
    smn[g][threadIdx.x] = mn; smx[g][threadIdx.x] = mx;
    __syncthreads();
    if (g == 0 && c < S) {
        for (int k = 1; k < 16; ++k) { mn = fmin(mn, smn[k][threadIdx.x]); mx = fmax(mx, smx[k][threadIdx.x]); }
        part[((long)blockIdx.y * 2 + 0) * S + c] = mn;
        part[((long)blockIdx.y * 2 + 1) * S + c] = mx;
    }
}
__global__ void colminmax_final_kernel(const double* __restrict__ part, int nchunk, int S, double* __restrict__ mn_out,
                                       double* __restrict__ mx_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= S) return;
    double mn = INFINITY, mx = -INFINITY;
    for (int k = 0; k < nchunk; ++k) {
        mn = fmin(mn, part[((long)k * 2 + 0) * S + c]);
        mx = fmax(mx, part[((long)k * 2 + 1) * S + c]);
    }
    mn_out[c] = mn; mx_out[c] = mx;
}
// MinMaxScaler.transform (X *= scale_; X += min_: two float64 roundings, no fused multiply-add) followed by the
// .float() of :184 -> the fp32 matrix the training step reads
__global__ void minmax_transform_kernel(const double* __restrict__ X, long total, int S, const double* __restrict__ scale,
                                        const double* __restrict__ min_, float* __restrict__ out) {
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)          // two roundings, as numpy's X *= scale_; X += min_  (no fused multiply-add)
        const int c = (int)(o % S);
        const double t = X[o] * scale[c];
        out[o] = (float)(t + min_[c]);
    }
}

// Checkpoint tail (src/mlp_df_wgan_train.py:291-292).  total_real: scaler.inverse_transform(scaler.transform(x)) in
// float64 -- numpy's four in-place passes X *= scale_; X += min_; X -= min_; X /= scale_, one rounding each.
__global__ void minmax_roundtrip_kernel(const double* __restrict__ X, long total, int S, const double* __restrict__ scale,
                                        const double* __restrict__ min_, double* __restrict__ out) {
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
        const int c = (int)(o % S);
        double t = X[o] * scale[c];
        t = t + min_[c];
        t = t - min_[c];
        out[o] = t / scale[c];
    }
}
// total_fake: scaler.inverse_transform on the float32 generated rows -- X -= min_; X /= scale_ with float64 constants on a
// float32 array: each pass computes in double and rounds to float32 (the arithmetic of sample_post's first two steps)
__global__ void minmax_inverse_f32_kernel(const float* __restrict__ Y, int ldy, long M, int S, const double* __restrict__ scale,
                                          const double* __restrict__ min_, float* __restrict__ out) {
    const long total = M * S;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
        const long m = o / S;
        const int c = (int)(o - m * S);
        const float t = (float)((double)Y[m * ldy + c] - min_[c]);
        out[o] = (float)((double)t / scale[c]);
    }
}

// ---- burst extraction (src/extract.py:58-122) -------------------------------------------------------------------
// One workgroup per packet trace (ragged: packets off[i] .. off[i+1] of the concatenated time / size arrays, float64).
// Long-gap truncation (first / last inter-packet gap > 10 s), skip of the leading incoming packets, then the trace is
// cut into runs of equal-sign packets: a run's thread (the one at its first packet) adds the sizes left to right, as
// the reference's loop does, so the sums are bit-identical for any data.  Run indices come from a ballot / prefix
// count per 256-packet block.
__device__ __forceinline__ int sgn_f64(double x) { return (x > 0.0) - (x < 0.0); }

__global__ __launch_bounds__(256) void extract_bursts_kernel(const double* __restrict__ T, const double* __restrict__ Sz,
                                                             const long long* __restrict__ off, int S, double cell,
                                                             double* __restrict__ feat, double* __restrict__ bsum,
                                                             double* __restrict__ btime, int* __restrict__ meta) {
    __shared__ int s_first, s_last, s_p0, s_wave[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long o = off[i];
    const int n = (int)(off[i + 1] - o);
    const double* t = T + o;
    const double* s = Sz + o;
    if (tid == 0) { s_first = 0x7fffffff; s_last = -1; s_p0 = 0x7fffffff; }
    __syncthreads();
    int lf = 0x7fffffff, ll = -1;
    for (int j = tid; j + 1 < n; j += 256)
        if (t[j + 1] - t[j] > 10.0) { lf = min(lf, j); ll = max(ll, j); }
    if (ll >= 0) { atomicMin(&s_first, lf); atomicMax(&s_last, ll); }
    __syncthreads();
    int start = 0, end = n;
    if (s_last >= 0) {
        if (s_first < 50) start = s_first + 1;
        if (s_last > 50) end = s_last + 1;
    }
    int lp = 0x7fffffff;
    for (int j = start + tid; j < end; j += 256)
        if (s[j] > 0.0) { lp = j; break; }
    if (lp != 0x7fffffff) atomicMin(&s_p0, lp);
    __syncthreads();
    const int p0 = s_p0 == 0x7fffffff ? end - 1 : s_p0;
    const double t0 = t[p0];
    int base = 0;
    for (int c = p0; c < end; c += 256) {
        const int q = c + tid;
        const bool in = q < end;
        const int sq = in ? sgn_f64(s[q]) : 0;
        const bool flag = in && (q == p0 || sq != sgn_f64(s[q - 1]));
        const unsigned long long m = __ballot(flag);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = __popcll(m & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += s_wave[w];
            total += s_wave[w];
        }
        if (flag) {
            const int r = base + before;
            double sum = s[q];
            for (int k = q + 1; k < end && sgn_f64(s[k]) == sq; ++k) sum += s[k];
            bsum[o + r] = sum;
            btime[o + r] = t[q] - t0;
            if (r + 1 < S) feat[(long)i * S + 1 + r] = cell > 0.0 ? fabs(sum) / cell : fabs(sum);
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) {
        feat[(long)i * S] = (double)base;
        meta[3 * i] = p0; meta[3 * i + 1] = end; meta[3 * i + 2] = base;
    }
    for (int j = 1 + min(base, S - 1) + tid; j < S; j += 256) feat[(long)i * S + j] = 0.0;
}

}  // namespace

extern "C" {

int wfd_rng_dropout_mask_slots(const unsigned long long* slots, int nslots, float p, const unsigned long long* state,
                               void* stream) {
    if (!slots || nslots < 1 || nslots > 8 || p < 0.f || p >= 1.f) return WFD_ERR_ARG;
    MaskSlots a{};
    long nmax = 0;
    for (int s = 0; s < nslots; ++s) {
        a.out[s] = reinterpret_cast<float*>(slots[4 * s]);
        a.n[s] = (long)slots[4 * s + 1];
        a.seed[s] = slots[4 * s + 2];
        a.base[s] = slots[4 * s + 3];
        if (!a.out[s] || a.n[s] <= 0) return WFD_ERR_ARG;
        if (a.n[s] > nmax) nmax = a.n[s];
    }
    long blocks = (nmax / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rng_dropout_mask_slots_kernel, dim3((unsigned)blocks, nslots), dim3(256), 0, (hipStream_t)stream, a, p,
                       state);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_rng_dropout_mask(float* out, long n, float p, unsigned long long seed, const unsigned long long* state,
                         unsigned long long base, void* stream) {
    if (!out || n <= 0 || p < 0.f || p >= 1.f) return WFD_ERR_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rng_dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, p,
                       seed, state, base);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_rng_normal_rows(float* out, int ld, int M, int Z, unsigned long long seed, const unsigned long long* state,
                        unsigned long long base, void* stream) {
    if (!out || M <= 0 || Z <= 0 || ld < Z) return WFD_ERR_ARG;
    long blocks = (((long)M * Z + 3) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rng_normal_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, ld, M, Z,
                       seed, state, base);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_gather_batch(const float* X, int ldX, const int* y, const long long* idx, const unsigned long long* state,
                     long n_idx, long n_rows, long row_offset, int B, int S, int C, float* real, float* fake, int ldd,
                     float* zin, int ldz, int Z, int* labels_out, void* stream) {
    if (!X || !y || !idx || !real || B <= 0 || n_idx <= 0 || n_rows <= 0 || row_offset < 0) return WFD_ERR_ARG;
    const GatherArgs a{X, ldX, y, idx, state, n_idx, n_rows, row_offset, S, C, real, fake, ldd, zin, ldz, Z, labels_out, 0, 0l, 0l};
    hipLaunchKernelGGL(gather_batch_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_gather_batch_slots(const float* X, int ldX, const int* y, const long long* idx, const unsigned long long* state,
                           long n_idx, long n_rows, long row_offset, int B, int slots, long slot_idx_stride, int S, int C,
                           float* real, float* fake, int ldd, long slot_stride, float* zin, int ldz, int Z,
                           int* labels_out, void* stream) {
    if (!X || !y || !idx || !real || B <= 0 || slots <= 0 || n_idx <= 0 || n_rows <= 0 || row_offset < 0) return WFD_ERR_ARG;
    const GatherArgs a{X, ldX, y, idx, state, n_idx, n_rows, row_offset, S, C, real, fake, ldd, zin, ldz, Z, labels_out, B,
                       slot_stride, slot_idx_stride};
    hipLaunchKernelGGL(gather_batch_kernel, dim3(B * slots), dim3(256), 0, (hipStream_t)stream, a);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// wfd_rng_dropout_mask_slots + wfd_gather_batch_slots + wfd_rng_normal_rows (rows B * slots of zin, columns [0, Z), key zseed,
// counter state[0]) as one launch; same results as the three
int wfd_group_inputs(const unsigned long long* mslots, int nslots, float p, const unsigned long long* mask_state,
                     const float* X, int ldX, const int* y, const long long* idx, const unsigned long long* state,
                     long n_idx, long n_rows, long row_offset, int B, int slots, long slot_idx_stride, int S, int C,
                     float* real, float* fake, int ldd, long slot_stride, float* zin, int ldz, int Z, int* labels_out,
                     unsigned long long zseed, void* stream) {
    if (!mslots || nslots < 1 || nslots > 8 || p < 0.f || p >= 1.f) return WFD_ERR_ARG;
    if (!X || !y || !idx || !real || !zin || B <= 0 || slots <= 0 || n_idx <= 0 || n_rows <= 0 || row_offset < 0 || Z <= 0 ||
        ldz < Z) return WFD_ERR_ARG;
    GroupInputs a{};
    long nmax = 0;
    for (int s = 0; s < nslots; ++s) {
        a.m.out[s] = reinterpret_cast<float*>(mslots[4 * s]);
        a.m.n[s] = (long)mslots[4 * s + 1];
        a.m.seed[s] = mslots[4 * s + 2];
        a.m.base[s] = mslots[4 * s + 3];
        if (!a.m.out[s] || a.m.n[s] <= 0) return WFD_ERR_ARG;
        if (a.m.n[s] > nmax) nmax = a.m.n[s];
    }
    long mb = (nmax / 4 + 255) / 256;
    if (mb > 1024) mb = 1024;
    if (mb < 1) mb = 1;
    a.nslots = nslots; a.mblocks = (int)mb; a.p = p; a.mstate = mask_state;
    a.g = GatherArgs{X, ldX, y, idx, state, n_idx, n_rows, row_offset, S, C, real, fake, ldd, zin, ldz, Z, labels_out, B,
                     slot_stride, slot_idx_stride};
    a.gblocks = B * slots;
    a.zseed = zseed; a.zrows = B * slots;
    long zb = (((long)a.zrows * Z + 3) / 4 + 255) / 256;
    if (zb > 256) zb = 256;
    a.zblocks = (int)zb;
    hipLaunchKernelGGL(group_inputs_kernel, dim3((unsigned)(mb * nslots + a.gblocks + zb)), dim3(256), 0, (hipStream_t)stream, a);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_rng_step_inputs(float* zout, int ld, int M, int Z, unsigned long long zseed, float* mout, long nmask, float p,
                        unsigned long long mseed, const unsigned long long* state, void* stream) {
    if (!zout || !mout || M <= 0 || Z <= 0 || ld < Z || nmask <= 0 || p < 0.f || p >= 1.f) return WFD_ERR_ARG;
    long zb = (((long)M * Z + 3) / 4 + 255) / 256, mb = (nmask / 4 + 255) / 256;
    if (zb > 256) zb = 256;
    if (mb > 2048) mb = 2048;
    if (mb < 1) mb = 1;
    hipLaunchKernelGGL(rng_step_inputs_kernel, dim3((unsigned)(zb + mb)), dim3(256), 0, (hipStream_t)stream, zout, ld, M,
                       Z, zseed, mout, nmask, p, mseed, state, (int)zb);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_rng_uniform(float* out, long n, unsigned long long seed, const unsigned long long* state,
                    unsigned long long base, void* stream) {
    if (!out || n <= 0) return WFD_ERR_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rng_uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                       state, base);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_interpolate_rows(const float* real, const float* fake, const float* alpha, float* out, int B, int S, int W,
                         int ld, void* stream) {
    if (!real || !fake || !alpha || !out || B <= 0 || W < S || ld < W) return WFD_ERR_ARG;
    hipLaunchKernelGGL(interpolate_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, real, fake, alpha, out, S,
                       W, ld);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_iter_advance(unsigned long long* state, unsigned long long inc0, unsigned long long inc1, void* stream) {
    if (!state) return WFD_ERR_ARG;
    hipLaunchKernelGGL(iter_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, inc0, inc1);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_onehot_rows(const int* labels, int M, int C, float* out, int ld, int col0, void* stream) {
    if (!labels || !out || M <= 0) return WFD_ERR_ARG;
    hipLaunchKernelGGL(onehot_rows_kernel, dim3(M), dim3(128), 0, (hipStream_t)stream, labels, M, C, out, ld, col0);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_clip_bursts(double* X, long N, int S, double max_len, double max_out, double max_in, void* stream) {
    if (!X || N <= 0 || S <= 0) return WFD_ERR_ARG;
    long blocks = (N * S + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clip_bursts_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, N, S, max_len,
                       max_out, max_in);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_sample_post_mode(const float* Y, int ldy, int M, int S, const double* min_, const double* scale_, int mode, int* out,
                         int ldo, int* len_out, void* stream) {
    if (!Y || !min_ || !scale_ || !out || !len_out || M <= 0 || S <= 0 || ldy < S || ldo < S) return WFD_ERR_ARG;
    if (mode < 0 || mode > 3 || (size_t)S * sizeof(float) > 60000) return WFD_ERR_ARG;
    hipLaunchKernelGGL(sample_post_mode_kernel, dim3(M), dim3(256), (size_t)S * sizeof(float), (hipStream_t)stream, Y, ldy,
                       S, min_, scale_, mode, out, ldo, len_out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_sample_post(const float* Y, int ldy, int M, int S, const double* min_, const double* scale_, int* out, int ldo,
                    int* len_out, void* stream) {
    if (!Y || !min_ || !scale_ || !out || !len_out || (size_t)S * 4 > 60000) return WFD_ERR_ARG;
    if (M >= 1024 && ((uintptr_t)Y & 15) == 0 && (ldy & 3) == 0)
        hipLaunchKernelGGL(sample_post_v4_kernel, dim3(M), dim3(256), (size_t)S * sizeof(float), (hipStream_t)stream, Y, ldy, S,
                           min_, scale_, out, ldo, len_out);
    else
        hipLaunchKernelGGL(sample_post_kernel, dim3(M), dim3(256), (size_t)S * sizeof(float), (hipStream_t)stream, Y, ldy, S,
                           min_, scale_, out, ldo, len_out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_version(void) { return 1; }

int wfd_colminmax_f64(const double* X, long N, int S, double* part, int max_chunks, double* mn, double* mx, void* stream) {
    if (!X || !part || !mn || !mx || N <= 0 || S <= 0 || max_chunks <= 0) return WFD_ERR_ARG;
    long nchunk = (N + 255) / 256;
    if (nchunk > max_chunks) nchunk = max_chunks;
    const long rpc = (N + nchunk - 1) / nchunk;
    nchunk = (N + rpc - 1) / rpc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colminmax_partial_kernel, dim3((S + 63) / 64, (unsigned)nchunk), dim3(64, 16), 0, st, X, N, S, rpc, part);
    WFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(colminmax_final_kernel, dim3((S + 255) / 256), dim3(256), 0, st, part, (int)nchunk, S, mn, mx);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_minmax_transform_f64(const double* X, long N, int S, const double* scale, const double* min_, float* out,
                             void* stream) {
    if (!X || !scale || !min_ || !out || N <= 0 || S <= 0) return WFD_ERR_ARG;
    const long total = N * S;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(minmax_transform_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, total, S, scale,
                       min_, out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_minmax_roundtrip_f64(const double* X, long N, int S, const double* scale, const double* min_, double* out,
                             void* stream) {
    if (!X || !scale || !min_ || !out || N <= 0 || S <= 0) return WFD_ERR_ARG;
    const long total = N * S;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(minmax_roundtrip_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, total, S, scale,
                       min_, out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_minmax_inverse_f32(const float* Y, int ldy, long M, int S, const double* scale, const double* min_, float* out,
                           void* stream) {
    if (!Y || !scale || !min_ || !out || M <= 0 || S <= 0 || ldy < S) return WFD_ERR_ARG;
    long blocks = (M * S + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(minmax_inverse_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, Y, ldy, M, S,
                       scale, min_, out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_extract_bursts(const double* times, const double* sizes, const long long* off, int ntraces, int S, double cell_norm,
                       double* feat, double* bsum, double* btime, int* meta, void* stream) {
    if (!times || !sizes || !off || !feat || !bsum || !btime || !meta || ntraces <= 0 || S < 1 || cell_norm < 0.0)
        return WFD_ERR_ARG;
    hipLaunchKernelGGL(extract_bursts_kernel, dim3(ntraces), dim3(256), 0, (hipStream_t)stream, times, sizes, off, S, cell_norm,
                       feat, bsum, btime, meta);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

}  // extern "C"
