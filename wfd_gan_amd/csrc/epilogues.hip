// Fused epilogue / reduction kernels of the MLP Generator + Discriminator step (gfx950).
//
// Every kernel here consumes the split-K slabs P[z][M][ldp] written by gemm.hip, summing them in fixed order
// (deterministic) while applying the math that follows the contraction in the reference:
//   bias + ReLU / sigmoid / affine(BN-eval)         src/model.py:20-34 (Generator blocks), DF layers
//   bias + LeakyReLU(0.2) + Dropout(0.2)            src/model.py:56-66 (Discriminator)
//   (.) o d  chain steps of the closed-form input gradient   src/mlp_df_wgan_train.py:203-213 (autograd.grad)
//   row-wise ||g||^2, the W-div / WGAN-GP penalty and its gradient seed   :207-215, src/utils.py:82-83
//   BatchNorm1d(eps=0.8) training forward / backward  src/model.py:23
//   sigmoid + learned-length tail mask              src/model.py:42-47
//   RMSprop                                          src/mlp_df_wgan_train.py:163-164 (torch defaults)
// All are HBM/L2-bound streaming kernels: coalesced along the feature dimension, wave-reduced where a row or a
// column is reduced.
#include "common.h"

namespace {

// Sum of the split-K slabs of one element.  Every slab load is issued before the first add (the loads are
// independent; a loop that waits for each one pays the L2/HBM latency ns times), slabs beyond ns re-read slab 0 and
// are discarded.  The sum is a fixed pairwise tree over 16 slots with zeros in the unused ones, so the result does
// not depend on which NS instantiation ran: ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)) + ...
template <int NS>
__device__ __forceinline__ float slab_tree(const float* __restrict__ p, int ns, long ss) {
    float v[NS];
#pragma unroll
    for (int z = 0; z < NS; ++z) v[z] = p[(z < ns ? (long)z : 0l) * ss];
#pragma unroll
    for (int z = 0; z < NS; ++z) v[z] = z < ns ? v[z] : 0.f;
#pragma unroll
    for (int w = 1; w < NS; w *= 2)
#pragma unroll
        for (int z = 0; z + w < NS; z += 2 * w) v[z] += v[z + w];
    return v[0];
}
__device__ __forceinline__ float slab_sum(const float* __restrict__ P, int ns, long ss, long idx) {
    const float* p = P + idx;
    if (ns <= 2) return slab_tree<2>(p, ns, ss);
    if (ns <= 4) return slab_tree<4>(p, ns, ss);
    if (ns <= 8) return slab_tree<8>(p, ns, ss);
    float a = slab_tree<16>(p, ns, ss);
    for (int z = 16; z < ns; z += 16) a += slab_tree<16>(p + (long)z * ss, ns - z, ss);   // many-way splits (weight grads)
    return a;
}
template <int NS>
__device__ __forceinline__ float4 slab_tree4(const float* __restrict__ p, int ns, long ss) {
    float4 v[NS];
#pragma unroll
    for (int z = 0; z < NS; ++z) v[z] = *reinterpret_cast<const float4*>(p + (z < ns ? (long)z : 0l) * ss);
#pragma unroll
    for (int z = 0; z < NS; ++z)
        if (z >= ns) v[z] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 1; w < NS; w *= 2)
#pragma unroll
        for (int z = 0; z + w < NS; z += 2 * w) {
            v[z].x += v[z + w].x; v[z].y += v[z + w].y; v[z].z += v[z + w].z; v[z].w += v[z + w].w;
        }
    return v[0];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------------------
// Element-wise epilogues: one thread per 4 consecutive columns (float4 when every operand is 16-byte aligned),
// flat over M x N/4 so that tall-skinny conv outputs (M = B*L, N = 32) and short-wide MLP outputs fill the chip.
template <bool VEC>
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ P, int ns, long ss, long idx, int nvalid) {
    float4 a;
    if constexpr (VEC) {
        const float* p = P + idx;
        if (ns <= 2) a = slab_tree4<2>(p, ns, ss);
        else if (ns <= 4) a = slab_tree4<4>(p, ns, ss);
        else if (ns <= 8) a = slab_tree4<8>(p, ns, ss);
        else {
            a = slab_tree4<16>(p, ns, ss);
            for (int z = 16; z < ns; z += 16) {
                const float4 b0 = slab_tree4<16>(p + (long)z * ss, ns - z, ss);
                a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            }
        }
    } else {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nvalid; ++k) t[k] = slab_sum(P, ns, ss, idx + k);
        a = make_float4(t[0], t[1], t[2], t[3]);
    }
    return a;
}
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int nvalid) {
    if constexpr (VEC) return *reinterpret_cast<const float4*>(p);
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nvalid; ++k) t[k] = p[k];
    return make_float4(t[0], t[1], t[2], t[3]);
}
template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, float4 v, int nvalid) {
    if constexpr (VEC) { *reinterpret_cast<float4*>(p) = v; return; }
    const float t[4] = {v.x, v.y, v.z, v.w};
    for (int k = 0; k < nvalid; ++k) p[k] = t[k];
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : elu_expm1(v);
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    return v;
}

// out[m][n] = act(sumP * scale[n] + shift[n]);  act: 0 none, 1 relu, 2 elu, 3 sigmoid
template <bool VEC>
__global__ void ep_affine_act_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                                     const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                     float* __restrict__ out, int ldo) {
    const int n4 = (N + 3) >> 2;
    const long total = (long)M * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4, nv = min(4, N - n);
        float4 v = slab_sum4<VEC>(P, ns, ss, (long)m * ldp + n, nv);
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < nv) {
                const float sh = shift ? shift[n + k] : 0.f;
                r[k] = scale ? __fmaf_rn(r[k], scale[n + k], sh) : r[k] + sh;
                r[k] = act_apply(r[k], act);
            }
        }
        store4<VEC>(out + (long)m * ldo + n, make_float4(r[0], r[1], r[2], r[3]), nv);
    }
}

// Discriminator hidden layer: z = sumP + b; h = leaky(z) * mask; d = mask * (z > 0 ? 1 : slope)
template <bool VEC>
__global__ void ep_lrelu_drop_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                                     const float* __restrict__ bias, const float* __restrict__ mask, int ldmask,
                                     float slope, float* __restrict__ H, int ldh, float* __restrict__ D, int ldd) {
    const int n4 = (N + 3) >> 2;
    const long total = (long)M * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4, nv = min(4, N - n);
        const float4 v = slab_sum4<VEC>(P, ns, ss, (long)m * ldp + n, nv);
        const float4 mk4 = load4<VEC>(mask + (long)m * ldmask + n, nv);
        const float z[4] = {v.x, v.y, v.z, v.w}, mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
        float h[4], d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float zz = z[k] + (k < nv ? bias[n + k] : 0.f);
            const float lr = zz > 0.f ? zz : zz * slope;
            h[k] = lr * mk[k];
            d[k] = zz > 0.f ? mk[k] : mk[k] * slope;
        }
        store4<VEC>(H + (long)m * ldh + n, make_float4(h[0], h[1], h[2], h[3]), nv);
        store4<VEC>(D + (long)m * ldd + n, make_float4(d[0], d[1], d[2], d[3]), nv);
    }
}

// Discriminator head (N = 1 GEMV) + start of the input-gradient chain.  One workgroup per row.
//   v[m] = <H3[m], w4> + b4;  delta3[m] = w4 o D3[m];  delta3s[m] = rowscale[m] * delta3[m]
__global__ void d_head_kernel(const float* __restrict__ H3, int ldh, const float* __restrict__ D3, int ldd,
                              const float* __restrict__ w4, const float* __restrict__ b4, int K,
                              const float* __restrict__ rowscale, float* __restrict__ v,
                              float* __restrict__ delta, float* __restrict__ delta_s, int ldo) {
    __shared__ float red[16];
    const int m = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float w = w4[k];
        acc = __fmaf_rn(H3[(long)m * ldh + k], w, acc);
        const float dl = w * D3[(long)m * ldd + k];
        delta[(long)m * ldo + k] = dl;
        if (delta_s) delta_s[(long)m * ldo + k] = dl * rowscale[m];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) v[m] = acc + b4[0];
}

// The last hidden layer's epilogue and the head in one launch (one dependent step less per critic pass): per row,
// ep_lrelu_drop's math on the slabs of layer 3 (H3, D3 are still written: the weight gradients and the t-chain read them)
// followed by d_head's on the values just computed.  Same operations in the same order as the two kernels.
__global__ void d_head_lrelu_kernel(const float* __restrict__ P, int ns, long ss, int ldp,
                                    const float* __restrict__ bias, const float* __restrict__ mask, int ldmask,
                                    float slope, float* __restrict__ H3, int ldh, float* __restrict__ D3, int ldd,
                                    const float* __restrict__ w4, const float* __restrict__ b4, int K,
                                    const float* __restrict__ rowscale, float* __restrict__ v,
                                    float* __restrict__ delta, float* __restrict__ delta_s, int ldo) {
    __shared__ float red[16];
    const int m = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float zz = slab_sum(P, ns, ss, (long)m * ldp + k) + bias[k];
        const float mk = mask[(long)m * ldmask + k];
        const float lr = zz > 0.f ? zz : zz * slope;
        const float h = lr * mk, d = zz > 0.f ? mk : mk * slope;
        H3[(long)m * ldh + k] = h;
        D3[(long)m * ldd + k] = d;
        const float w = w4[k];
        acc = __fmaf_rn(h, w, acc);
        const float dl = w * d;
        delta[(long)m * ldo + k] = dl;
        if (delta_s) delta_s[(long)m * ldo + k] = dl * rowscale[m];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) v[m] = acc + b4[0];
}

// out = sumP * f(D) * colscale;  dmode: 0 no D, 1 multiply by D, 2 multiply by (D > 0), 3 ELU' from its output
// (D > 0 ? 1 : D + 1).  Optional second copy out2 = out * rowscale[m].
template <bool VEC>
__global__ void ep_mul_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                              const float* __restrict__ D, int ldd, int dmode, const float* __restrict__ colscale,
                              const float* __restrict__ rowscale, float* __restrict__ out, float* __restrict__ out2,
                              int ldo) {
    const int n4 = (N + 3) >> 2;
    const long total = (long)M * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4, nv = min(4, N - n);
        const float4 v4 = slab_sum4<VEC>(P, ns, ss, (long)m * ldp + n, nv);
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        if (dmode) {
            const float4 d4 = load4<VEC>(D + (long)m * ldd + n, nv);
            const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (dmode == 1) v[k] *= d[k];
                else if (dmode == 2) v[k] = d[k] > 0.f ? v[k] : 0.f;
                else v[k] = d[k] > 0.f ? v[k] : v[k] * (d[k] + 1.f);
            }
        }
        if (colscale) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < nv) v[k] *= colscale[n + k];
        }
        store4<VEC>(out + (long)m * ldo + n, make_float4(v[0], v[1], v[2], v[3]), nv);
        if (out2) {
            const float rs = rowscale[m];
            store4<VEC>(out2 + (long)m * ldo + n, make_float4(v[0] * rs, v[1] * rs, v[2] * rs, v[3] * rs), nv);
        }
    }
}

// Penalty on one row of the input gradient g = sumP[m][0..S): n = ||g||^2.
//   mode 0 (W-div, k=2, p=6):  pen[m] = n^3,              seed = 6 n^2 invB * g
//   mode 1 (WGAN-GP):          pen[m] = (sqrt(n) - 1)^2,  seed = 2 lambda invB (sqrt(n)-1)/sqrt(n) * g
// seed (= d loss / d g) is written to GH[m][0..S).  One workgroup per row.
__global__ void ep_penalty_row_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int S, int mode,
                                      float invB, float lambda, const float* __restrict__ rowflag,
                                      float* __restrict__ nsq, float* __restrict__ pen,
                                      float* __restrict__ GH, int ldg) {
    __shared__ float red[16];
    extern __shared__ float grow[];
    const int m = blockIdx.x;
    float acc = 0.f;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        const float g = slab_sum(P, ns, ss, (long)m * ldp + j);
        grow[j] = g;
        acc = __fmaf_rn(g, g, acc);
    }
    const float n = block_sum(acc, red);
    float coef, pv;
    if (mode == 0) {
        pv = n * n * n;
        coef = 6.f * n * n * invB;
    } else {
        const float nr = sqrtf(n);
        pv = (nr - 1.f) * (nr - 1.f);
        coef = nr > 0.f ? 2.f * lambda * invB * (nr - 1.f) / nr : 0.f;
    }
    if (rowflag && rowflag[m] == 0.f) { coef = 0.f; pv = 0.f; }     // row carries no penalty term
    if (threadIdx.x == 0) { nsq[m] = n; pen[m] = pv; }
    for (int j = threadIdx.x; j < S; j += blockDim.x) GH[(long)m * ldg + j] = coef * grow[j];
}

#include "colsum_body.h"

__device__ __forceinline__ float rg_reduce(float v, float (*red)[CT]) {
    __syncthreads();
    red[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int g = 0; g < RG; ++g) r += red[g][threadIdx.x];
    return r;
}

// out[n] (+)= sum_m X[m][n] * (rowscale ? rowscale[m] : 1)
__global__ void colsum_kernel(const float* __restrict__ X, int ldx, int M, int N, const float* __restrict__ rowscale,
                              float* __restrict__ out, int accumulate) {
    __shared__ float red[RG][CT];
    const int n = blockIdx.x * CT + threadIdx.x;
    float a = 0.f;
    if (n < N) a = col_partial(X, ldx, rowscale, M, n, threadIdx.y, a);
    a = rg_reduce(a, red);
    if (n < N && threadIdx.y == 0) out[n] = accumulate ? out[n] + a : a;
}

// Several column sums in ONE launch (all bias / head gradients and the loss scalars of a step).  desc holds, per
// segment, 11 words: [X, ldx, rowscale, X2, ldx2, rowscale2, M, N, out, first_block, acc]   (X2 = 0: single source)
//   out[n] = sum_m X[m][n] * rs[m] + sum_m X2[m][n] * rs2[m];   acc[n] += out[n] when acc != 0 (epoch sums)
// OPT: an output that lies inside the flat gradient buffer [gbase, gbase + numel) is a parameter gradient: its parameter
// (same offset in pbase / sqbase) takes its RMSprop step right here (the vectors of a network whose matrices are updated in
// their gradients' stores), and block 0 does the step's bookkeeping (state[0] += inc0, state[1] += inc1).
template <bool OPT>
__device__ __forceinline__ void colsum_multi_body(const long long* __restrict__ desc, int nseg, const ColsumOpt& o, int bx,
                                                  float (*red)[CT]) {
    if constexpr (OPT) {
        if (o.state && bx == 0 && threadIdx.x == 0 && threadIdx.y == 0) { o.state[0] += o.inc0; o.state[1] += o.inc1; }
    }
    int sgi = 0;
    for (int k = 1; k < nseg; ++k)
        if ((int)desc[k * 11 + 9] <= bx) sgi = k;
    const long long* d = desc + sgi * 11;
    const float* X = reinterpret_cast<const float*>(d[0]);
    const int ldx = (int)d[1];
    const float* rs = reinterpret_cast<const float*>(d[2]);
    const float* X2 = reinterpret_cast<const float*>(d[3]);
    const int ldx2 = (int)d[4];
    const float* rs2 = reinterpret_cast<const float*>(d[5]);
    const int M = (int)d[6], N = (int)d[7];
    float* out = reinterpret_cast<float*>(d[8]);
    float* acc = reinterpret_cast<float*>(d[10]);
    const int n = (bx - (int)d[9]) * CT + threadIdx.x;
    float a = 0.f;
    if (n < N) {
        a = col_partial(X, ldx, rs, M, n, threadIdx.y, a);
        if (X2) a += col_partial(X2, ldx2, rs2, M, n, threadIdx.y, 0.f);
    }
    a = rg_reduce(a, red);
    if (n < N && threadIdx.y == 0) {
        out[n] = a;
        if (acc) acc[n] += a;
        if constexpr (OPT) {
            const long off = (out + n) - o.gbase;
            if (off >= 0 && off < o.numel) {
                float pv = o.pbase[off], sv = o.sqbase[off];
                rmsprop_update(pv, sv, a, o.lr, o.alpha, o.eps);
                o.sqbase[off] = sv;
                o.pbase[off] = pv;
            }
        }
    }
}
template <bool OPT>
__global__ void colsum_multi_kernel(const long long* __restrict__ desc, int nseg, ColsumOpt o) {
    __shared__ float red[RG][CT];
    colsum_multi_body<OPT>(desc, nseg, o, blockIdx.x, red);
}

// RMSprop (torch defaults: alpha .99, eps 1e-8, no momentum, not centered, wd 0), one flat pass
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, long n,
                               float lr, float alpha, float eps, float gscale,
                               unsigned long long* __restrict__ state, unsigned long long inc0,
                               unsigned long long inc1) {
    // end-of-step bookkeeping rides on the last kernel of the step: advance the Philox counter / batch cursor
    // (nothing in this kernel reads them)
    if (state && blockIdx.x == 0 && threadIdx.x == 0) { state[0] += inc0; state[1] += inc1; }
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            // gradients and square averages are touched once per step: they stream past the caches (as in the fused launch)
            float4 gv = nt_load4(g + i);
            float4 sv = nt_load4(sq + i);
            float4 pv = *reinterpret_cast<const float4*>(p + i);
            rmsprop_update(pv.x, sv.x, gv.x * gscale, lr, alpha, eps);
            rmsprop_update(pv.y, sv.y, gv.y * gscale, lr, alpha, eps);
            rmsprop_update(pv.z, sv.z, gv.z * gscale, lr, alpha, eps);
            rmsprop_update(pv.w, sv.w, gv.w * gscale, lr, alpha, eps);
            nt_store4(sq + i, sv);
            *reinterpret_cast<float4*>(p + i) = pv;
        } else {
            for (long j = i; j < n; ++j) {
                float pj = p[j], sj = sq[j];
                rmsprop_update(pj, sj, g[j] * gscale, lr, alpha, eps);
                sq[j] = sj;
                p[j] = pj;
            }
        }
    }
}

// The same update on up to 8 element ranges [beg, end) of the flat buffers: the vectors (biases, BatchNorm affine, the
// critic's head) of a network whose weight matrices were updated in their gradient's store (wfd_gemm_f32_rmsprop).
__global__ void rmsprop_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, RmsRanges r,
                                      float lr, float alpha, float eps, unsigned long long* __restrict__ state,
                                      unsigned long long inc0, unsigned long long inc1) {
    if (state && blockIdx.x == 0 && threadIdx.x == 0) { state[0] += inc0; state[1] += inc1; }
    rmsprop_ranges_body(p, g, sq, r, lr, alpha, eps, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm1d training forward, split so that data-parallel ranks can combine statistics between the halves.
// stats kernel: Z = sumP + bias; per-column local mean and M2 = sum (z - mean)^2 (two passes, no cancellation).
// blockDim = (CT columns, RG row groups).
__global__ void bn_stats_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                                const float* __restrict__ bias, float* __restrict__ Z, int ldz,
                                float* __restrict__ mean, float* __restrict__ m2) {
    __shared__ float red[RG][CT];
    const int n = blockIdx.x * CT + threadIdx.x, rg = threadIdx.y;
    const bool ok = n < N;
    const float b = ok && bias ? bias[n] : 0.f;
    float s = 0.f;
    if (ok)
        for (int m = rg; m < M; m += RG) {
            const float z = slab_sum(P, ns, ss, (long)m * ldp + n) + b;
            Z[(long)m * ldz + n] = z;
            s += z;
        }
    const float mu = rg_reduce(s, red) / (float)M;
    float q = 0.f;
    if (ok)
        for (int m = rg; m < M; m += RG) {
            const float d = Z[(long)m * ldz + n] - mu;
            q = __fmaf_rn(d, d, q);
        }
    q = rg_reduce(q, red);
    if (ok && rg == 0) { mean[n] = mu; m2[n] = q; }
}

// Single-rank form: statistics + normalise + ReLU + running-stat update in one launch (same arithmetic as
// bn_stats followed by bn_apply_relu).
__global__ void bn_train_relu_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                                     const float* __restrict__ bias, float eps, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ XH, float* __restrict__ H,
                                     int ldo, float* __restrict__ mean, float* __restrict__ m2,
                                     float* __restrict__ invstd_out, float* __restrict__ rmean,
                                     float* __restrict__ rvar, float momentum, int nupd) {
    __shared__ float red[RG][CT];
    const int n = blockIdx.x * CT + threadIdx.x, rg = threadIdx.y;
    const bool ok = n < N;
    const float b = ok && bias ? bias[n] : 0.f;
    float s = 0.f;
    if (ok)
        for (int m = rg; m < M; m += RG) {
            const float z = slab_sum(P, ns, ss, (long)m * ldp + n) + b;
            XH[(long)m * ldo + n] = z;            // parked; each thread re-reads only what it wrote
            s += z;
        }
    const float mu = rg_reduce(s, red) / (float)M;
    float q = 0.f;
    if (ok)
        for (int m = rg; m < M; m += RG) {
            const float d = XH[(long)m * ldo + n] - mu;
            q = __fmaf_rn(d, d, q);
        }
    q = rg_reduce(q, red);
    if (!ok) return;
    const float istd = 1.0f / sqrtf(q / (float)M + eps);
    const float ga = gamma[n], be = beta[n];
    for (int m = rg; m < M; m += RG) {
        const float xh = (XH[(long)m * ldo + n] - mu) * istd;
        XH[(long)m * ldo + n] = xh;
        H[(long)m * ldo + n] = fmaxf(__fmaf_rn(xh, ga, be), 0.f);
    }
    if (rg == 0) {
        mean[n] = mu; m2[n] = q; invstd_out[n] = istd;
        if (rmean) {
            const float uv = q / ((float)M - 1.f);
            float rm = rmean[n], rv = rvar[n];
            for (int u = 0; u < nupd; ++u) {
                rm = (1.f - momentum) * rm + momentum * mu;
                rv = (1.f - momentum) * rv + momentum * uv;
            }
            rmean[n] = rm; rvar[n] = rv;
        }
    }
}

// Small-batch form of bn_train_relu (M <= RG * RPT rows): each thread keeps its rows of the column in registers
// instead of parking them in memory between the three passes, and the block is CTS columns wide so that a
// 1024-column layer spreads over N / CTS workgroups.  Same summation order as the kernels above (bit-identical).
template <int CTS>
__device__ __forceinline__ float rg_reduce_t(float v, float (*red)[CTS]) {
    __syncthreads();
    red[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int g = 0; g < RG; ++g) r += red[g][threadIdx.x];
    return r;
}

template <int CTS, int RPT>
__global__ __launch_bounds__(CTS * RG) void bn_train_relu_small_kernel(
    const float* __restrict__ P, int ns, long ss, int ldp, int M, int N, const float* __restrict__ bias, float eps,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ XH, float* __restrict__ H,
    int ldo, float* __restrict__ mean, float* __restrict__ m2, float* __restrict__ invstd_out,
    float* __restrict__ rmean, float* __restrict__ rvar, float momentum, int nupd) {
    __shared__ float red[RG][CTS];
    const int n = blockIdx.x * CTS + threadIdx.x, rg = threadIdx.y;
    // blockIdx.y = row group (batched generator forwards: each group of M rows is its own BatchNorm batch)
    P += (long)blockIdx.y * M * ldp;
    XH += (long)blockIdx.y * M * ldo;
    H += (long)blockIdx.y * M * ldo;
    mean += (long)blockIdx.y * N; m2 += (long)blockIdx.y * N; invstd_out += (long)blockIdx.y * N;
    const bool ok = n < N;
    const float b = ok && bias ? bias[n] : 0.f;
    const float ga = ok ? gamma[n] : 0.f, be = ok ? beta[n] : 0.f;        // requested with the rows, not behind the reductions
    float z[RPT];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int m = rg + RG * r;
        z[r] = 0.f;
        if (ok && m < M) {
            z[r] = slab_sum(P, ns, ss, (long)m * ldp + n) + b;
            s += z[r];
        }
    }
    const float mu = rg_reduce_t<CTS>(s, red) / (float)M;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        if (ok && rg + RG * r < M) {
            const float d = z[r] - mu;
            q = __fmaf_rn(d, d, q);
        }
    q = rg_reduce_t<CTS>(q, red);
    if (!ok) return;
    const float istd = 1.0f / sqrtf(q / (float)M + eps);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int m = rg + RG * r;
        if (m < M) {
            const float xh = (z[r] - mu) * istd;
            XH[(long)m * ldo + n] = xh;
            H[(long)m * ldo + n] = fmaxf(__fmaf_rn(xh, ga, be), 0.f);
        }
    }
    if (rg == 0) {
        mean[n] = mu; m2[n] = q; invstd_out[n] = istd;
        if (rmean) {
            const float uv = q / ((float)M - 1.f);
            float rm = rmean[n], rv = rvar[n];
            for (int u = 0; u < nupd; ++u) {
                rm = (1.f - momentum) * rm + momentum * mu;
                rv = (1.f - momentum) * rv + momentum * uv;
            }
            rmean[n] = rm; rvar[n] = rv;
        }
    }
}

// apply: var = m2 / count (biased); xhat = (z - mean) * rsqrt(var + eps); h = relu(gamma * xhat + beta).
// running stats: rm = (1-mom) rm + mom mean; rv = (1-mom) rv + mom * m2 / (count - 1), applied nupd times
// (the G-step re-runs the same forward on the same batch: src/mlp_df_wgan_train.py:196,240).
__global__ void bn_apply_relu_kernel(const float* __restrict__ Z, int ldz, int M, int N, const float* __restrict__ mean,
                                     const float* __restrict__ m2, float count, float eps,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ XH, float* __restrict__ H, int ldo,
                                     float* __restrict__ invstd_out, float* __restrict__ rmean, float* __restrict__ rvar,
                                     float momentum, int nupd) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float mu = mean[n], var = m2[n] / count;
    const float istd = 1.0f / sqrtf(var + eps);
    const float ga = gamma[n], be = beta[n];
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        const float xh = (Z[(long)m * ldz + n] - mu) * istd;
        XH[(long)m * ldo + n] = xh;
        H[(long)m * ldo + n] = fmaxf(__fmaf_rn(xh, ga, be), 0.f);
    }
    if (blockIdx.y == 0) {
        invstd_out[n] = istd;
        if (rmean) {
            const float uv = m2[n] / (count - 1.f);
            float rm = rmean[n], rv = rvar[n];
            for (int u = 0; u < nupd; ++u) {
                rm = (1.f - momentum) * rm + momentum * mu;
                rv = (1.f - momentum) * rv + momentum * uv;
            }
            rmean[n] = rm;
            rvar[n] = rv;
        }
    }
}

__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ m2, float count,
                                         int N, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                         int nupd, int groups) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float rm = rmean[n], rv = rvar[n];
    for (int g = 0; g < groups; ++g) {          // the groups' batches in order, as successive forward calls would
        const float mu = mean[(long)g * N + n], uv = m2[(long)g * N + n] / (count - 1.f);
        for (int u = 0; u < nupd; ++u) {
            rm = (1.f - momentum) * rm + momentum * mu;
            rv = (1.f - momentum) * rv + momentum * uv;
        }
    }
    rmean[n] = rm;
    rvar[n] = rv;
}

// Both BatchNorm layers of the generator in one launch, for a whole group of stacked forwards: blockIdx.y selects the
// layer; its `groups` batches are applied in order, the last one `nupd_last` times (the G-step's second forward of the same
// batch, src/mlp_df_wgan_train.py:240).  Element for element the operations of bn_running_update_kernel applied batch by batch.
__global__ void bn_running_update_layers_kernel(BnLayer a, BnLayer b, float count, float momentum, int groups, int nupd_last) {
    bn_running_layer_body(blockIdx.y ? b : a, blockIdx.x * blockDim.x + threadIdx.x, count, momentum, groups, nupd_last);
}

// The small work that ends a G-step whose weight matrices were updated in their gradients' stores, as ONE launch:
//   blocks [0, cs_blocks)            column sums (the first layer's bias gradient) with their RMSprop step   = colsum_multi<OPT>
//   the next rr_blocks               RMSprop on the other vector ranges of the flat buffers                   = rmsprop_ranges
//   the last 2 * bn_blocks           running statistics of the two BatchNorm layers, all stacked batches      = bn_running_update_layers
// The three parts touch disjoint memory and each is the body of the kernel it replaces.  Blocks are (CT, RG) threads.
__global__ void g_end_kernel(const long long* __restrict__ desc, int nseg, int cs_blocks, ColsumOpt o, RmsRanges r,
                             int rr_blocks, BnRun bn) {
    __shared__ float red[RG][CT];
    int b = blockIdx.x;
    if (b < cs_blocks) { colsum_multi_body<true>(desc, nseg, o, b, red); return; }
    b -= cs_blocks;
    const int tid = threadIdx.y * CT + threadIdx.x;
    if (b < rr_blocks) {
        rmsprop_ranges_body(o.pbase, o.gbase, o.sqbase, r, o.lr, o.alpha, o.eps, (long)b * (CT * RG) + tid);
        return;
    }
    b -= rr_blocks;
    if (bn.bn_blocks > 0)
        bn_running_layer_body(b / bn.bn_blocks ? bn.b : bn.a, (b % bn.bn_blocks) * (CT * RG) + tid, bn.count, bn.momentum,
                              bn.groups, bn.nupd_last);
}

// BatchNorm backward through ReLU.  dy = sumP o [H > 0].
//  phase & 1: sums[0][n] = sum_m dy, sums[1][n] = sum_m dy * xhat   (local; DP ranks all-reduce them)
//  phase & 2: dz = gamma * invstd * (dy - sums0/count - xhat * sums1/count);  dgamma = sums1; dbeta = sums0;
//             dbias[n] = sum_m dz (the Linear bias in front of BN; mathematically 0, computed as autograd does)
// thread per column.
__global__ void bn_bwd_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                              const float* __restrict__ H, const float* __restrict__ XH, int ldh,
                              const float* __restrict__ gamma, const float* __restrict__ invstd, float count,
                              float* __restrict__ sums, float* __restrict__ DZ, int lddz, float* __restrict__ dgamma,
                              float* __restrict__ dbeta, float* __restrict__ dbias, int phase) {
    __shared__ float red[RG][CT];
    const int n = blockIdx.x * CT + threadIdx.x, rg = threadIdx.y;
    const bool ok = n < N;
    float s0 = 0.f, s1 = 0.f;
    if (phase & 1) {
        // dy is parked in DZ between the two passes (each thread re-reads only what it wrote)
        if (ok)
            for (int m = rg; m < M; m += RG) {
                const long o = (long)m * ldh + n;
                const float dy = H[o] > 0.f ? slab_sum(P, ns, ss, (long)m * ldp + n) : 0.f;
                DZ[(long)m * lddz + n] = dy;
                s0 += dy;
                s1 = __fmaf_rn(dy, XH[o], s1);
            }
        s0 = rg_reduce(s0, red);
        s1 = rg_reduce(s1, red);
        if (ok && rg == 0) { sums[n] = s0; sums[N + n] = s1; }
    }
    if (phase & 2) {
        if (!(phase & 1) && ok) { s0 = sums[n]; s1 = sums[N + n]; }
        float sb = 0.f;
        if (ok) {
            const float k = gamma[n] * invstd[n], a0 = s0 / count, a1 = s1 / count;
            for (int m = rg; m < M; m += RG) {
                const float dy = DZ[(long)m * lddz + n];
                const float dz = k * (dy - a0 - XH[(long)m * ldh + n] * a1);
                DZ[(long)m * lddz + n] = dz;
                sb += dz;
            }
        }
        sb = rg_reduce(sb, red);
        if (ok && rg == 0) { dgamma[n] = s1; dbeta[n] = s0; dbias[n] = sb; }
    }
}

// Small-batch single-launch form of bn_bwd (phase 3 only): dy and xhat stay in registers.  Same summation order.
template <int CTS, int RPT>
__global__ __launch_bounds__(CTS * RG) void bn_bwd_small_kernel(
    const float* __restrict__ P, int ns, long ss, int ldp, int M, int N, const float* __restrict__ H,
    const float* __restrict__ XH, int ldh, const float* __restrict__ gamma, const float* __restrict__ invstd,
    float count, float* __restrict__ sums, float* __restrict__ DZ, int lddz, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ float red[RG][CTS];
    const int n = blockIdx.x * CTS + threadIdx.x, rg = threadIdx.y;
    const bool ok = n < N;
    // every global load of the launch up front: the column's gamma / 1/std (read behind the reductions they were a dependent
    // round trip in front of the stores), and the product unconditionally instead of behind the load of H that masks it
    const float k = ok ? gamma[n] * invstd[n] : 0.f;
    float dy[RPT], xh[RPT], hv[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int m = rg + RG * r;
        dy[r] = 0.f; xh[r] = 0.f; hv[r] = 0.f;
        if (ok && m < M) {
            const long o = (long)m * ldh + n;
            hv[r] = H[o];
            dy[r] = slab_sum(P, ns, ss, (long)m * ldp + n);
            xh[r] = XH[o];
        }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (ok && rg + RG * r < M) {
            dy[r] = hv[r] > 0.f ? dy[r] : 0.f;
            s0 += dy[r];
            s1 = __fmaf_rn(dy[r], xh[r], s1);
        }
    }
    s0 = rg_reduce_t<CTS>(s0, red);
    s1 = rg_reduce_t<CTS>(s1, red);
    float sb = 0.f;
    if (ok) {
        const float a0 = s0 / count, a1 = s1 / count;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int m = rg + RG * r;
            if (m < M) {
                const float dz = k * (dy[r] - a0 - xh[r] * a1);
                DZ[(long)m * lddz + n] = dz;
                sb += dz;
            }
        }
    }
    sb = rg_reduce_t<CTS>(sb, red);
    if (ok && rg == 0) { sums[n] = s0; sums[N + n] = s1; dgamma[n] = s1; dbeta[n] = s0; dbias[n] = sb; }
}

// Generator output: y = sigmoid(sumP + b); bl = y0 * (smax - smin) + smin (two fp32 roundings, as torch does);
// idx = trunc(bl) + 1; out[j] = j < idx ? y[j] : 0.  One workgroup per row.  idx >= S (the reference raises
// IndexError there) sets *err and clamps.
__global__ void ep_sigmoid_lenmask_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int S,
                                          const float* __restrict__ bias, float range, float smin,
                                          float* __restrict__ out, int ldo, int* __restrict__ idx_out,
                                          int* __restrict__ err, int rows_per_slot, long slot_stride,
                                          unsigned long long* __restrict__ adv_state, unsigned long long inc0,
                                          unsigned long long inc1) {
    const int m = blockIdx.x;
    // optional: the per-iteration device state (Philox counter, permutation cursor) advanced by this launch instead of by
    // one of its own (wfd_iter_advance); nothing in this kernel reads it
    if (adv_state && m == 0 && threadIdx.x == 0) { adv_state[0] += inc0; adv_state[1] += inc1; }
    // output rows of slot s = m / rows_per_slot start slot_stride floats after those of slot s - 1
    if (rows_per_slot > 0) out += (long)(m / rows_per_slot) * slot_stride - (long)(m / rows_per_slot) * rows_per_slot * ldo;
    const float y0 = 1.0f / (1.0f + expf(-(slab_sum(P, ns, ss, (long)m * ldp) + bias[0])));
    const float bl = __fadd_rn(__fmul_rn(y0, range), smin);
    long idx = (long)bl + 1;
    if (idx >= S || idx < 0) {
        if (threadIdx.x == 0) atomicOr(err, 1);
        idx = idx < 0 ? 0 : S;
    }
    if (threadIdx.x == 0) idx_out[m] = (int)idx;
    for (int j = threadIdx.x; j < S; j += blockDim.x) {
        float y = 0.f;
        if (j < idx) y = 1.0f / (1.0f + expf(-(slab_sum(P, ns, ss, (long)m * ldp + j) + bias[j])));
        out[(long)m * ldo + j] = y;
    }
}

// The same for an unsplit product with 16-byte aligned rows (bulk sampling: 32768 rows per chunk): four columns per thread and
// access.  Columns of the last, partial group are handled one by one -- behind column S - 1 the output row may hold other
// data (the one-hot columns of the critic's input).  Element for element the arithmetic of the kernel above.
__global__ void ep_sigmoid_lenmask_v4_kernel(const float* __restrict__ P, int ldp, int S, const float* __restrict__ bias,
                                             float range, float smin, float* __restrict__ out, int ldo,
                                             int* __restrict__ idx_out, int* __restrict__ err, int rows_per_slot,
                                             long slot_stride, unsigned long long* __restrict__ adv_state,
                                             unsigned long long inc0, unsigned long long inc1) {
    const int m = blockIdx.x;
    if (adv_state && m == 0 && threadIdx.x == 0) { adv_state[0] += inc0; adv_state[1] += inc1; }
    if (rows_per_slot > 0) out += (long)(m / rows_per_slot) * slot_stride - (long)(m / rows_per_slot) * rows_per_slot * ldo;
    const float* __restrict__ prow = P + (long)m * ldp;
    float* __restrict__ orow = out + (long)m * ldo;
    const float y0 = 1.0f / (1.0f + expf(-(prow[0] + bias[0])));
    const float bl = __fadd_rn(__fmul_rn(y0, range), smin);
    long idx = (long)bl + 1;
    if (idx >= S || idx < 0) {
        if (threadIdx.x == 0) atomicOr(err, 1);
        idx = idx < 0 ? 0 : S;
    }
    if (threadIdx.x == 0) idx_out[m] = (int)idx;
    const int S4 = S >> 2;
    for (int q = threadIdx.x; q < S4; q += blockDim.x) {
        const int j = 4 * q;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < idx) {                    // groups wholly behind the length index are zeros without reading the product
            const float4 p = *reinterpret_cast<const float4*>(prow + j);
            const float4 b = *reinterpret_cast<const float4*>(bias + j);
            y.x = 1.0f / (1.0f + expf(-(p.x + b.x)));
            if (j + 1 < idx) y.y = 1.0f / (1.0f + expf(-(p.y + b.y)));
            if (j + 2 < idx) y.z = 1.0f / (1.0f + expf(-(p.z + b.z)));
            if (j + 3 < idx) y.w = 1.0f / (1.0f + expf(-(p.w + b.w)));
        }
        *reinterpret_cast<float4*>(orow + j) = y;
    }
    for (int j = 4 * S4 + threadIdx.x; j < S; j += blockDim.x) {
        float y = 0.f;
        if (j < idx) y = 1.0f / (1.0f + expf(-(prow[j] + bias[j])));
        orow[j] = y;
    }
}

// Gradient at the generator's last pre-activation (thread per column j, loops rows):
//   dfake[m][j] = dscale * sumP[m][j] (D path, dscale = -1/B) + (j >= 1 ? dDF[m][j-1] : 0)
//   dz[m][j]    = j < idx[m] ? dfake * y (1 - y) : 0, y = fake[m][j];   dbias[j] = sum_m dz
__global__ void g_out_bwd_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int S, float dscale,
                                 const float* __restrict__ dDF, int lddf, const float* __restrict__ fake, int ldf,
                                 const int* __restrict__ idx, float* __restrict__ DZ, int lddz,
                                 float* __restrict__ dbias) {
    __shared__ float red[RG][CT];
    const int j = blockIdx.x * CT + threadIdx.x;
    float sb = 0.f;
    if (j < S)
        for (int m0 = threadIdx.y; m0 < M; m0 += 4 * RG) {
            // four rows per batch, every load issued before the arithmetic (rows past M re-read row m0)
            float g[4], dd[4], y[4];
            int lim[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + k * RG < M ? m0 + k * RG : m0;
                g[k] = slab_sum(P, ns, ss, (long)m * ldp + j);
                dd[k] = (dDF && j >= 1) ? dDF[(long)m * lddf + j - 1] : 0.f;
                y[k] = fake[(long)m * ldf + j];
                lim[k] = idx[m];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + k * RG;
                if (m < M) {
                    float dz = 0.f;
                    if (j < lim[k]) dz = (dscale * g[k] + dd[k]) * y[k] * (1.f - y[k]);
                    DZ[(long)m * lddz + j] = dz;
                    sb += dz;
                }
            }
        }
    sb = rg_reduce(sb, red);
    if (j < S && threadIdx.y == 0) dbias[j] = sb;
}

// ---- BatchNorm over a tall batch (M > 128 rows: the large-batch steps of BASELINE config 4) ------------------------
// The column-per-thread kernels above walk all M rows from N / 32 workgroups; here the rows are cut into chunks of
// TALL_ROWS, a (64-column x chunk) workgroup produces the chunk's partial statistics, and one thread per column
// combines the chunks in order (Chan's pairwise update for mean / M2, plain ordered sums for the backward), so the
// result is deterministic and every access is a coalesced 256-byte row segment.
constexpr int TALL_ROWS = 128, TALL_C = 64, TALL_R = 4;

__device__ __forceinline__ float tall_reduce(float v, float (*red)[TALL_C]) {
    __syncthreads();
    red[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int g = 0; g < TALL_R; ++g) r += red[g][threadIdx.x];
    return r;
}

// z = slab sum + bias -> Z (parked);  part[(2 c) N + n] = chunk mean, part[(2 c + 1) N + n] = chunk M2
__global__ void bn_tallf_stats_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                                      const float* __restrict__ bias, float* __restrict__ Z, int ldz,
                                      float* __restrict__ part) {
    __shared__ float red[TALL_R][TALL_C];
    const int n = blockIdx.x * TALL_C + threadIdx.x, c = blockIdx.y;
    const int m0 = c * TALL_ROWS, m1 = min(M, m0 + TALL_ROWS);
    const bool ok = n < N;
    const float b = ok && bias ? bias[n] : 0.f;
    float s = 0.f;
    if (ok)
        for (int m = m0 + threadIdx.y; m < m1; m += TALL_R) {
            const float z = slab_sum(P, ns, ss, (long)m * ldp + n) + b;
            Z[(long)m * ldz + n] = z;
            s += z;
        }
    const float mu = tall_reduce(s, red) / (float)(m1 - m0);
    float q = 0.f;
    if (ok)
        for (int m = m0 + threadIdx.y; m < m1; m += TALL_R) {
            const float d = Z[(long)m * ldz + n] - mu;
            q = __fmaf_rn(d, d, q);
        }
    q = tall_reduce(q, red);
    if (ok && threadIdx.y == 0) { part[(long)(2 * c) * N + n] = mu; part[(long)(2 * c + 1) * N + n] = q; }
}

__global__ void bn_tallf_combine_kernel(const float* __restrict__ part, int M, int N, float* __restrict__ mean,
                                        float* __restrict__ m2) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float na = 0.f, mu = 0.f, q = 0.f;
    for (int c = 0; c * TALL_ROWS < M; ++c) {
        const float nb = (float)(min(M, (c + 1) * TALL_ROWS) - c * TALL_ROWS);
        const float d = part[(long)(2 * c) * N + n] - mu, tot = na + nb;
        mu += d * (nb / tot);
        q += part[(long)(2 * c + 1) * N + n] + d * d * (na * nb / tot);
        na = tot;
    }
    mean[n] = mu;
    m2[n] = q;
}

// backward, pass 1: dy = sumP o [H > 0] -> DZ (parked); part[(2 c) N + n] = sum dy, part[(2 c + 1) N + n] = sum dy * xhat
__global__ void bn_tallb_sums_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int N,
                                     const float* __restrict__ H, const float* __restrict__ XH, int ldh,
                                     float* __restrict__ DZ, int lddz, float* __restrict__ part) {
    __shared__ float red[TALL_R][TALL_C];
    const int n = blockIdx.x * TALL_C + threadIdx.x, c = blockIdx.y;
    const int m0 = c * TALL_ROWS, m1 = min(M, m0 + TALL_ROWS);
    const bool ok = n < N;
    float s0 = 0.f, s1 = 0.f;
    if (ok)
        for (int m = m0 + threadIdx.y; m < m1; m += TALL_R) {
            const long o = (long)m * ldh + n;
            const float dy = H[o] > 0.f ? slab_sum(P, ns, ss, (long)m * ldp + n) : 0.f;
            DZ[(long)m * lddz + n] = dy;
            s0 += dy;
            s1 = __fmaf_rn(dy, XH[o], s1);
        }
    s0 = tall_reduce(s0, red);
    s1 = tall_reduce(s1, red);
    if (ok && threadIdx.y == 0) { part[(long)(2 * c) * N + n] = s0; part[(long)(2 * c + 1) * N + n] = s1; }
}

// ordered sum of `nslot` interleaved partial rows: out_k[n] = sum_c part[(nslot c + k) N + n]
__global__ void bn_tallb_combine_kernel(const float* __restrict__ part, int chunks, int nslot, int N, float* __restrict__ o0,
                                        float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int k = 0; k < nslot; ++k) {
        float a = 0.f;
        for (int c = 0; c < chunks; ++c) a += part[(long)(nslot * c + k) * N + n];
        float* o = k == 0 ? o0 : o1;
        float* oo = k == 0 ? o2 : o3;
        if (o) o[n] = a;
        if (oo) oo[n] = a;
    }
}

// backward, pass 2: dz = gamma * invstd * (dy - s0 / count - xhat * s1 / count) in place in DZ; part[c N + n] = sum dz
__global__ void bn_tallb_dz_kernel(int M, int N, const float* __restrict__ XH, int ldh, const float* __restrict__ gamma,
                                   const float* __restrict__ invstd, float count, const float* __restrict__ sums,
                                   float* __restrict__ DZ, int lddz, float* __restrict__ part) {
    __shared__ float red[TALL_R][TALL_C];
    const int n = blockIdx.x * TALL_C + threadIdx.x, c = blockIdx.y;
    const int m0 = c * TALL_ROWS, m1 = min(M, m0 + TALL_ROWS);
    const bool ok = n < N;
    float sb = 0.f;
    if (ok) {
        const float k = gamma[n] * invstd[n], a0 = sums[n] / count, a1 = sums[N + n] / count;
        for (int m = m0 + threadIdx.y; m < m1; m += TALL_R) {
            const float dy = DZ[(long)m * lddz + n];
            const float dz = k * (dy - a0 - XH[(long)m * ldh + n] * a1);
            DZ[(long)m * lddz + n] = dz;
            sb += dz;
        }
    }
    sb = tall_reduce(sb, red);
    if (ok && threadIdx.y == 0) part[(long)c * N + n] = sb;
}

// g_out_bwd without the bias-gradient sum, one thread per (row, column): the tall-batch form (the caller takes the column
// sums of DZ with wfd_colsum_tall)
__global__ void g_out_dz_kernel(const float* __restrict__ P, int ns, long ss, int ldp, int M, int S, float dscale,
                                const float* __restrict__ dDF, int lddf, const float* __restrict__ fake, int ldf,
                                const int* __restrict__ idx, float* __restrict__ DZ, int lddz) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= S) return;
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        float dz = 0.f;
        if (j < idx[m]) {
            const float y = fake[(long)m * ldf + j];
            const float dd = (dDF && j >= 1) ? dDF[(long)m * lddf + j - 1] : 0.f;
            dz = (dscale * slab_sum(P, ns, ss, (long)m * ldp + j) + dd) * y * (1.f - y);
        }
        DZ[(long)m * lddz + j] = dz;
    }
}

// D-step scalars.  v rows [0,B) real, [B,2B) fake; pen rows likewise (W-div) or B interpolate rows (GP).
// out: [0] d_loss [1] penalty [2] w_dist [3] mean_rv [4] mean_fv ; acc[0..2] += d_loss, w_dist, 1 (epoch sums)
__global__ void d_loss_kernel(const float* __restrict__ v, const float* __restrict__ pen, int B, int npen, float invB,
                              float pen_weight, float* __restrict__ out, float* __restrict__ acc) {
    __shared__ float red[16];
    float sr = 0.f, sf = 0.f, sp = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { sr += v[i]; sf += v[B + i]; }
    for (int i = threadIdx.x; i < npen; i += blockDim.x) sp += pen[i];
    sr = block_sum(sr, red);
    sf = block_sum(sf, red);
    sp = block_sum(sp, red);
    if (threadIdx.x == 0) {
        const float mr = sr * invB, mf = sf * invB, pv = sp * invB;
        const float dl = -mr + mf + pen_weight * pv;
        out[0] = dl; out[1] = pv; out[2] = mr - mf; out[3] = mr; out[4] = mf;
        if (acc) { acc[0] += dl; acc[1] += mr - mf; acc[2] += 1.f; }
    }
}

// G-step head scalars: g_loss = -mean(fv); nsel = #{fv > 0}.  out[0] g_loss, out[1] nsel
__global__ void g_head_kernel(const float* __restrict__ fv, int B, float invB, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { s += fv[i]; c += fv[i] > 0.f ? 1.f : 0.f; }
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) { out[0] = -s * invB; out[1] = c; }
}

}  // namespace

// ---- internal launchers (used by step.hip) and their extern "C" faces -----------------------------------------
static inline dim3 col_grid(int N, int M, int bx = 256) { return dim3(cdiv(N, bx), M < 64 ? M : 64); }
// threads per block of the flat element-wise epilogues: single-wave blocks while the whole tensor is a few hundred waves
// (batch-64 layers: the blocks then spread over all CUs instead of filling a quarter of them; +0.6 % on the step),
// 256 threads once there are enough of those to cover the chip
static inline int ep_block(int M, int N) { return (long)M * ((N + 3) / 4) <= 131072 ? 64 : 256; }
static inline unsigned flat_grid(int M, int N) {
    const long t = (long)M * ((N + 3) / 4);
    const int bs = ep_block(M, N);
    long b = (t + bs - 1) / bs;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}
// one workgroup per row in the head kernels: one column per thread at the reference's width (512), so the row is a single
// round of loads (same rule in wfd_d_head and wfd_d_head_lrelu: their sums stay bit-identical)
static inline int head_block(int K) { return K >= 512 ? 512 : 256; }
static inline bool al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int wfd_ep_affine_act(const float* P, int ns, long ss, int ldp, int M, int N, const float* scale, const float* shift,
                      int act, float* out, int ldo, void* stream) {
    if (!P || !out || M <= 0 || N <= 0) return WFD_ERR_ARG;
    const bool vec = al16(P) && al16(out) && !(ldp & 3) && !(ldo & 3) && !(ss & 3) && !(N & 3);
    if (vec)
        hipLaunchKernelGGL(ep_affine_act_kernel<true>, dim3(flat_grid(M, N)), dim3(ep_block(M, N)), 0, (hipStream_t)stream, P, ns,
                           ss, ldp, M, N, scale, shift, act, out, ldo);
    else
        hipLaunchKernelGGL(ep_affine_act_kernel<false>, dim3(flat_grid(M, N)), dim3(ep_block(M, N)), 0, (hipStream_t)stream, P, ns,
                           ss, ldp, M, N, scale, shift, act, out, ldo);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_ep_lrelu_drop(const float* P, int ns, long ss, int ldp, int M, int N, const float* bias, const float* mask,
                      int ldmask, float slope, float* H, int ldh, float* D, int ldd, void* stream) {
    if (!P || !bias || !mask || !H || !D) return WFD_ERR_ARG;
    const bool vec = al16(P) && al16(mask) && al16(H) && al16(D) && !(ldp & 3) && !(ldmask & 3) && !(ldh & 3) &&
                     !(ldd & 3) && !(ss & 3) && !(N & 3);
    if (vec)
        hipLaunchKernelGGL(ep_lrelu_drop_kernel<true>, dim3(flat_grid(M, N)), dim3(ep_block(M, N)), 0, (hipStream_t)stream, P, ns,
                           ss, ldp, M, N, bias, mask, ldmask, slope, H, ldh, D, ldd);
    else
        hipLaunchKernelGGL(ep_lrelu_drop_kernel<false>, dim3(flat_grid(M, N)), dim3(ep_block(M, N)), 0, (hipStream_t)stream, P, ns,
                           ss, ldp, M, N, bias, mask, ldmask, slope, H, ldh, D, ldd);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_d_head(const float* H3, int ldh, const float* D3, int ldd, const float* w4, const float* b4, int M, int K,
               const float* rowscale, float* v, float* delta, float* delta_s, int ldo, void* stream) {
    if (!H3 || !D3 || !w4 || !b4 || !v || !delta) return WFD_ERR_ARG;
    if (delta_s && !rowscale) return WFD_ERR_ARG;
    hipLaunchKernelGGL(d_head_kernel, dim3(M), dim3(head_block(K)), 0, (hipStream_t)stream, H3, ldh, D3, ldd, w4, b4, K, rowscale,
                       v, delta, delta_s, ldo);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_d_head_lrelu(const float* P, int ns, long ss, int ldp, const float* bias, const float* mask, int ldmask,
                     float slope, float* H3, int ldh, float* D3, int ldd, const float* w4, const float* b4, int M, int K,
                     const float* rowscale, float* v, float* delta, float* delta_s, int ldo, void* stream) {
    if (!P || !bias || !mask || !H3 || !D3 || !w4 || !b4 || !v || !delta || M <= 0 || K <= 0 || ns < 1) return WFD_ERR_ARG;
    if (delta_s && !rowscale) return WFD_ERR_ARG;
    hipLaunchKernelGGL(d_head_lrelu_kernel, dim3(M), dim3(head_block(K)), 0, (hipStream_t)stream, P, ns, ss, ldp, bias, mask, ldmask,
                       slope, H3, ldh, D3, ldd, w4, b4, K, rowscale, v, delta, delta_s, ldo);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_ep_mul(const float* P, int ns, long ss, int ldp, int M, int N, const float* D, int ldd, int dmode,
               const float* colscale, const float* rowscale, float* out, float* out2, int ldo, void* stream) {
    if (!P || !out || (dmode && !D) || (out2 && !rowscale)) return WFD_ERR_ARG;
    const bool vec = al16(P) && al16(D) && al16(out) && al16(out2) && !(ldp & 3) && !(ldd & 3) && !(ldo & 3) &&
                     !(ss & 3) && !(N & 3);
    if (vec)
        hipLaunchKernelGGL(ep_mul_kernel<true>, dim3(flat_grid(M, N)), dim3(ep_block(M, N)), 0, (hipStream_t)stream, P, ns, ss, ldp,
                           M, N, D, ldd, dmode, colscale, rowscale, out, out2, ldo);
    else
        hipLaunchKernelGGL(ep_mul_kernel<false>, dim3(flat_grid(M, N)), dim3(ep_block(M, N)), 0, (hipStream_t)stream, P, ns, ss,
                           ldp, M, N, D, ldd, dmode, colscale, rowscale, out, out2, ldo);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_ep_penalty_row(const float* P, int ns, long ss, int ldp, int M, int S, int mode, float invB, float lambda,
                       const float* rowflag, float* nsq, float* pen, float* GH, int ldg, void* stream) {
    if (!P || !nsq || !pen || !GH || S <= 0 || (size_t)S * 4 > 60000) return WFD_ERR_ARG;
    hipLaunchKernelGGL(ep_penalty_row_kernel, dim3(M), dim3(S >= 1024 ? 1024 : 512), (size_t)S * sizeof(float), (hipStream_t)stream, P, ns,
                       ss, ldp, S, mode, invB, lambda, rowflag, nsq, pen, GH, ldg);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_colsum(const float* X, int ldx, int M, int N, const float* rowscale, float* out, int accumulate,
               void* stream) {
    if (!X || !out) return WFD_ERR_ARG;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, CT)), dim3(CT, RG), 0, (hipStream_t)stream, X, ldx, M, N, rowscale,
                       out, accumulate);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_colsum_multi(const long long* desc, int nseg, int total_blocks, void* stream) {
    if (!desc || nseg <= 0 || total_blocks <= 0) return WFD_ERR_ARG;
    hipLaunchKernelGGL(colsum_multi_kernel<false>, dim3(total_blocks), dim3(CT, RG), 0, (hipStream_t)stream, desc, nseg,
                       ColsumOpt{});
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_colsum_multi_rmsprop(const long long* desc, int nseg, int total_blocks, const float* gbase, float* pbase,
                             float* sqbase, long numel, float lr, float alpha, float eps, unsigned long long* state,
                             unsigned long long inc0, unsigned long long inc1, void* stream) {
    if (!desc || nseg <= 0 || total_blocks <= 0 || !gbase || !pbase || !sqbase || numel <= 0) return WFD_ERR_ARG;
    ColsumOpt o{gbase, pbase, sqbase, numel, lr, alpha, eps, state, inc0, inc1};
    hipLaunchKernelGGL(colsum_multi_kernel<true>, dim3(total_blocks), dim3(CT, RG), 0, (hipStream_t)stream, desc, nseg, o);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_rmsprop(float* p, const float* g, float* sq, long n, float lr, float alpha, float eps, float gscale,
                unsigned long long* state, unsigned long long inc0, unsigned long long inc1, void* stream) {
    if (!p || !g || !sq || n <= 0) return WFD_ERR_ARG;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)sq) & 15) != 0) return WFD_ERR_ARG;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;          // one float4 per thread up to 8.4 M parameters (both networks here)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, sq, n, lr, alpha,
                       eps, gscale, state, inc0, inc1);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_rmsprop_ranges(float* p, const float* g, float* sq, const long* ranges, int nranges, float lr, float alpha, float eps,
                       unsigned long long* state, unsigned long long inc0, unsigned long long inc1, void* stream) {
    if (!p || !g || !sq || !ranges || nranges < 1 || nranges > 8) return WFD_ERR_ARG;
    RmsRanges r{};
    long total = 0;
    for (int k = 0; k < nranges; ++k) {
        r.beg[k] = ranges[2 * k]; r.end[k] = ranges[2 * k + 1];
        if (r.beg[k] < 0 || r.end[k] < r.beg[k]) return WFD_ERR_ARG;
        total += r.end[k] - r.beg[k];
    }
    r.n = nranges;
    if (total <= 0) return WFD_ERR_ARG;
    hipLaunchKernelGGL(rmsprop_ranges_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p, g, sq, r, lr, alpha,
                       eps, state, inc0, inc1);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_g_end(const long long* desc, int nseg, int cs_blocks, const float* gbase, float* pbase, float* sqbase, long numel,
              const long* ranges, int nranges, float lr, float alpha, float eps, unsigned long long* state,
              unsigned long long inc0, unsigned long long inc1, const float* meanA, const float* m2A, int NA, float* rmeanA,
              float* rvarA, const float* meanB, const float* m2B, int NB, float* rmeanB, float* rvarB, float count,
              float momentum, int groups, int nupd_last, void* stream) {
    if (!desc || nseg <= 0 || cs_blocks <= 0 || !gbase || !pbase || !sqbase || numel <= 0) return WFD_ERR_ARG;
    if (!ranges || nranges < 1 || nranges > 8) return WFD_ERR_ARG;
    ColsumOpt o{gbase, pbase, sqbase, numel, lr, alpha, eps, state, inc0, inc1};
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
    if (meanA) {          // optional part
        if (!m2A || !rmeanA || !rvarA || !meanB || !m2B || !rmeanB || !rvarB || NA < 1 || NB < 1 || groups < 1 ||
            nupd_last < 0 || count <= 1.f)
            return WFD_ERR_ARG;
        bn.a = BnLayer{meanA, m2A, rmeanA, rvarA, NA};
        bn.b = BnLayer{meanB, m2B, rmeanB, rvarB, NB};
        bn.count = count; bn.momentum = momentum; bn.groups = groups; bn.nupd_last = nupd_last;
        bn.bn_blocks = cdiv(NA > NB ? NA : NB, CT * RG);
    }
    const int rr_blocks = cdiv(total, CT * RG);
    hipLaunchKernelGGL(g_end_kernel, dim3(cs_blocks + rr_blocks + 2 * bn.bn_blocks), dim3(CT, RG), 0, (hipStream_t)stream, desc,
                       nseg, cs_blocks, o, r, rr_blocks, bn);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_stats(const float* P, int ns, long ss, int ldp, int M, int N, const float* bias, float* Z, int ldz,
                 float* mean, float* m2, void* stream) {
    if (!P || !Z || !mean || !m2) return WFD_ERR_ARG;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(cdiv(N, CT)), dim3(CT, RG), 0, (hipStream_t)stream, P, ns, ss, ldp, M, N,
                       bias, Z, ldz, mean, m2);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_train_relu(const float* P, int ns, long ss, int ldp, int M, int N, const float* bias, float eps,
                      const float* gamma, const float* beta, float* XH, float* H, int ldo, float* mean, float* m2,
                      float* invstd, float* rmean, float* rvar, float momentum, int nupd, void* stream) {
    if (!P || !gamma || !beta || !XH || !H || !mean || !m2 || !invstd || M < 2) return WFD_ERR_ARG;
    if (M <= RG * 8) {
        constexpr int CTS = 32;          // a block reads whole 128-byte lines of its rows (8 columns: 4x the bytes, 1 % of the B=64 step)
        if (M <= RG * 4)
            hipLaunchKernelGGL((bn_train_relu_small_kernel<CTS, 4>), dim3(cdiv(N, CTS)), dim3(CTS, RG), 0,
                               (hipStream_t)stream, P, ns, ss, ldp, M, N, bias, eps, gamma, beta, XH, H, ldo, mean, m2,
                               invstd, rmean, rvar, momentum, nupd);
        else
            hipLaunchKernelGGL((bn_train_relu_small_kernel<CTS, 8>), dim3(cdiv(N, CTS)), dim3(CTS, RG), 0,
                               (hipStream_t)stream, P, ns, ss, ldp, M, N, bias, eps, gamma, beta, XH, H, ldo, mean, m2,
                               invstd, rmean, rvar, momentum, nupd);
        WFD_CHECK_LAUNCH();
        return WFD_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    if (M <= 512) {          // up to a few row blocks the column-per-thread kernel is one launch and fast enough
        hipLaunchKernelGGL(bn_train_relu_kernel, dim3(cdiv(N, CT)), dim3(CT, RG), 0, st, P, ns, ss, ldp, M, N, bias, eps, gamma,
                           beta, XH, H, ldo, mean, m2, invstd, rmean, rvar, momentum, nupd);
        WFD_CHECK_LAUNCH();
        return WFD_OK;
    }
    // tall batch: chunk statistics -> ordered combine -> normalise.  z is parked in XH, the chunk partials
    // (2 * chunks * N floats <= M * ldo) in H, which the last kernel then overwrites.
    const int chunks = cdiv(M, TALL_ROWS);
    hipLaunchKernelGGL(bn_tallf_stats_kernel, dim3(cdiv(N, TALL_C), chunks), dim3(TALL_C, TALL_R), 0, st, P, ns, ss, ldp, M, N,
                       bias, XH, ldo, H);
    hipLaunchKernelGGL(bn_tallf_combine_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)H, M, N, mean, m2);
    hipLaunchKernelGGL(bn_apply_relu_kernel, col_grid(N, M, 64), dim3(64), 0, st, (const float*)XH, ldo, M, N,
                       (const float*)mean, (const float*)m2, (float)M, eps, gamma, beta, XH, H, ldo, invstd, rmean, rvar,
                       momentum, nupd);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_apply_relu(const float* Z, int ldz, int M, int N, const float* mean, const float* m2, float count, float eps,
                      const float* gamma, const float* beta, float* XH, float* H, int ldo, float* invstd, float* rmean,
                      float* rvar, float momentum, int nupd, void* stream) {
    if (!Z || !mean || !m2 || !gamma || !beta || !XH || !H || !invstd) return WFD_ERR_ARG;
    hipLaunchKernelGGL(bn_apply_relu_kernel, col_grid(N, M, 64), dim3(64), 0, (hipStream_t)stream, Z, ldz, M, N, mean, m2,
                       count, eps, gamma, beta, XH, H, ldo, invstd, rmean, rvar, momentum, nupd);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_running_update(const float* mean, const float* m2, float count, int N, float* rmean, float* rvar,
                          float momentum, int nupd, void* stream) {
    if (!mean || !m2 || !rmean || !rvar) return WFD_ERR_ARG;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, mean, m2, count,
                       N, rmean, rvar, momentum, nupd, 1);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_running_update_layers(const float* meanA, const float* m2A, int NA, float* rmeanA, float* rvarA,
                                 const float* meanB, const float* m2B, int NB, float* rmeanB, float* rvarB, float count,
                                 float momentum, int groups, int nupd_last, void* stream) {
    if (!meanA || !m2A || !rmeanA || !rvarA || !meanB || !m2B || !rmeanB || !rvarB || NA < 1 || NB < 1 || groups < 1 ||
        nupd_last < 0 || count <= 1.f)
        return WFD_ERR_ARG;
    BnLayer a{meanA, m2A, rmeanA, rvarA, NA}, b{meanB, m2B, rmeanB, rvarB, NB};
    hipLaunchKernelGGL(bn_running_update_layers_kernel, dim3(cdiv(NA > NB ? NA : NB, 256), 2), dim3(256), 0,
                       (hipStream_t)stream, a, b, count, momentum, groups, nupd_last);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// Batched generator forwards: `groups` independent BatchNorm batches of M rows each, stacked along the rows of P
// (the heads of the iterations that share generator parameters).  Statistics go to mean / m2 / invstd [groups][N];
// the running statistics are then updated once per group, in order.
int wfd_bn_train_relu_grouped(const float* P, int ns, long ss, int ldp, int M, int groups, int N, const float* bias,
                              float eps, const float* gamma, const float* beta, float* XH, float* H, int ldo, float* mean,
                              float* m2, float* invstd, float* rmean, float* rvar, float momentum, void* stream) {
    if (!P || !gamma || !beta || !XH || !H || !mean || !m2 || !invstd || M < 2 || groups < 1 || M > RG * 8)
        return WFD_ERR_ARG;
    constexpr int CTS = 32;          // a block reads whole 128-byte lines of its rows (8 columns: 4x the bytes, 1 % of the B=64 step)
    hipStream_t st = (hipStream_t)stream;
    if (M <= RG * 4)
        hipLaunchKernelGGL((bn_train_relu_small_kernel<CTS, 4>), dim3(cdiv(N, CTS), groups), dim3(CTS, RG), 0, st, P, ns, ss,
                           ldp, M, N, bias, eps, gamma, beta, XH, H, ldo, mean, m2, invstd, (float*)nullptr,
                           (float*)nullptr, momentum, 0);
    else
        hipLaunchKernelGGL((bn_train_relu_small_kernel<CTS, 8>), dim3(cdiv(N, CTS), groups), dim3(CTS, RG), 0, st, P, ns, ss,
                           ldp, M, N, bias, eps, gamma, beta, XH, H, ldo, mean, m2, invstd, (float*)nullptr,
                           (float*)nullptr, momentum, 0);
    WFD_CHECK_LAUNCH();
    if (rmean) {
        hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, mean, m2, (float)M, N, rmean,
                           rvar, momentum, 1, groups);
        WFD_CHECK_LAUNCH();
    }
    return WFD_OK;
}

int wfd_bn_bwd(const float* P, int ns, long ss, int ldp, int M, int N, const float* H, const float* XH, int ldh,
               const float* gamma, const float* invstd, float count, float* sums, float* DZ, int lddz, float* dgamma,
               float* dbeta, float* dbias, int phase, void* stream) {
    if (!P || !H || !XH || !gamma || !invstd || !sums) return WFD_ERR_ARG;
    if (!DZ || ((phase & 2) && (!dgamma || !dbeta || !dbias))) return WFD_ERR_ARG;
    if (phase == 3 && M <= RG * 8) {
        constexpr int CTS = 32;          // a block reads whole 128-byte lines of its rows (8 columns: 4x the bytes, 1 % of the B=64 step)
        if (M <= RG * 4)
            hipLaunchKernelGGL((bn_bwd_small_kernel<CTS, 4>), dim3(cdiv(N, CTS)), dim3(CTS, RG), 0, (hipStream_t)stream, P,
                               ns, ss, ldp, M, N, H, XH, ldh, gamma, invstd, count, sums, DZ, lddz, dgamma, dbeta, dbias);
        else
            hipLaunchKernelGGL((bn_bwd_small_kernel<CTS, 8>), dim3(cdiv(N, CTS)), dim3(CTS, RG), 0, (hipStream_t)stream, P,
                               ns, ss, ldp, M, N, H, XH, ldh, gamma, invstd, count, sums, DZ, lddz, dgamma, dbeta, dbias);
        WFD_CHECK_LAUNCH();
        return WFD_OK;
    }
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(cdiv(N, CT)), dim3(CT, RG), 0, (hipStream_t)stream, P, ns, ss, ldp, M, N, H, XH,
                       ldh, gamma, invstd, count, sums, DZ, lddz, dgamma, dbeta, dbias, phase);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// wfd_bn_bwd (phase 3) for a tall batch: `part` = 2 * ceil(M / 128) * N floats of scratch
int wfd_bn_bwd_tall(const float* P, int ns, long ss, int ldp, int M, int N, const float* H, const float* XH, int ldh,
                    const float* gamma, const float* invstd, float count, float* sums, float* DZ, int lddz, float* dgamma,
                    float* dbeta, float* dbias, float* part, void* stream) {
    if (!P || !H || !XH || !gamma || !invstd || !sums || !DZ || !dgamma || !dbeta || !dbias || !part || M < 2) return WFD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = cdiv(M, TALL_ROWS);
    const dim3 grid(cdiv(N, TALL_C), chunks), block(TALL_C, TALL_R);
    hipLaunchKernelGGL(bn_tallb_sums_kernel, grid, block, 0, st, P, ns, ss, ldp, M, N, H, XH, ldh, DZ, lddz, part);
    // sums = [sum dy ; sum dy * xhat]; dbeta = sum dy, dgamma = sum dy * xhat
    hipLaunchKernelGGL(bn_tallb_combine_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)part, chunks, 2, N, sums,
                       sums + N, dbeta, dgamma);
    hipLaunchKernelGGL(bn_tallb_dz_kernel, grid, block, 0, st, M, N, XH, ldh, gamma, invstd, count, (const float*)sums, DZ, lddz,
                       part);
    hipLaunchKernelGGL(bn_tallb_combine_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, (const float*)part, chunks, 1, N, dbias,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// many rows of an unsplit, 16-byte aligned product: the four-columns-per-thread form
static bool lenmask_v4(const float* P, int ns, int ldp, int M, const float* bias, const float* out, int ldo, long slot_stride) {
    const uintptr_t a = (uintptr_t)P | (uintptr_t)bias | (uintptr_t)out;
    return ns == 1 && M >= 1024 && (a & 15) == 0 && (ldp & 3) == 0 && (ldo & 3) == 0 && (slot_stride & 3) == 0;
}

int wfd_ep_sigmoid_lenmask(const float* P, int ns, long ss, int ldp, int M, int S, const float* bias, float range,
                           float smin, float* out, int ldo, int* idx, int* err, void* stream) {
    if (!P || !bias || !out || !idx || !err) return WFD_ERR_ARG;
    if (lenmask_v4(P, ns, ldp, M, bias, out, ldo, 0))
        hipLaunchKernelGGL(ep_sigmoid_lenmask_v4_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, P, ldp, S, bias, range, smin,
                           out, ldo, idx, err, 0, 0l, (unsigned long long*)nullptr, 0ull, 0ull);
    else
        hipLaunchKernelGGL(ep_sigmoid_lenmask_kernel, dim3(M), dim3(512), 0, (hipStream_t)stream, P, ns, ss, ldp, S, bias,
                           range, smin, out, ldo, idx, err, 0, 0l, (unsigned long long*)nullptr, 0ull, 0ull);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// the same for stacked generator forwards: rows [s*rows_per_slot, (s+1)*rows_per_slot) go to out + s*slot_stride
int wfd_ep_sigmoid_lenmask_slots(const float* P, int ns, long ss, int ldp, int M, int S, const float* bias, float range,
                                 float smin, float* out, int ldo, int rows_per_slot, long slot_stride, int* idx, int* err,
                                 void* stream) {
    if (!P || !bias || !out || !idx || !err || rows_per_slot <= 0 || M % rows_per_slot) return WFD_ERR_ARG;
    hipLaunchKernelGGL(ep_sigmoid_lenmask_kernel, dim3(M), dim3(512), 0, (hipStream_t)stream, P, ns, ss, ldp, S, bias,
                       range, smin, out, ldo, idx, err, rows_per_slot, slot_stride, (unsigned long long*)nullptr, 0ull, 0ull);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

// the same, and the launch also advances the device iteration state (wfd_iter_advance folded in: one launch fewer at the
// end of a group's stacked generator forward)
int wfd_ep_sigmoid_lenmask_slots_adv(const float* P, int ns, long ss, int ldp, int M, int S, const float* bias, float range,
                                     float smin, float* out, int ldo, int rows_per_slot, long slot_stride, int* idx, int* err,
                                     unsigned long long* state, unsigned long long inc0, unsigned long long inc1,
                                     void* stream) {
    if (!P || !bias || !out || !idx || !err || !state || rows_per_slot <= 0 || M % rows_per_slot) return WFD_ERR_ARG;
    hipLaunchKernelGGL(ep_sigmoid_lenmask_kernel, dim3(M), dim3(512), 0, (hipStream_t)stream, P, ns, ss, ldp, S, bias,
                       range, smin, out, ldo, idx, err, rows_per_slot, slot_stride, state, inc0, inc1);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_g_out_bwd(const float* P, int ns, long ss, int ldp, int M, int S, float dscale, const float* dDF, int lddf,
                  const float* fake, int ldf, const int* idx, float* DZ, int lddz, float* dbias, void* stream) {
    if (!P || !fake || !idx || !DZ) return WFD_ERR_ARG;
    if (!dbias) {            // dz only, fully parallel (tall batches; the bias gradient = wfd_colsum_tall of DZ)
        hipLaunchKernelGGL(g_out_dz_kernel, dim3(cdiv(S, 256), M < 1024 ? M : 1024), dim3(256), 0, (hipStream_t)stream, P, ns, ss,
                           ldp, M, S, dscale, dDF, lddf, fake, ldf, idx, DZ, lddz);
        WFD_CHECK_LAUNCH();
        return WFD_OK;
    }
    hipLaunchKernelGGL(g_out_bwd_kernel, dim3(cdiv(S, CT)), dim3(CT, RG), 0, (hipStream_t)stream, P, ns, ss, ldp, M, S,
                       dscale, dDF, lddf, fake, ldf, idx, DZ, lddz, dbias);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_d_loss(const float* v, const float* pen, int B, int npen, float invB, float pen_weight, float* out, float* acc,
               void* stream) {
    if (!v || !pen || !out) return WFD_ERR_ARG;
    hipLaunchKernelGGL(d_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, pen, B, npen, invB, pen_weight, out,
                       acc);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_g_head(const float* fv, int B, float invB, float* out, void* stream) {
    if (!fv || !out) return WFD_ERR_ARG;
    hipLaunchKernelGGL(g_head_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fv, B, invB, out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

}  // extern "C"
