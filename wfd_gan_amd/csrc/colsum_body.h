// Column-sum pieces shared by epilogues.hip (the column-reduction kernels) and gemm.hip (the fused weight-gradient launch
// that also carries a step's column sums).  Included inside each file's anonymous namespace.
#pragma once

// Column-reduction kernels use 32 columns x 16 row groups per workgroup (coalesced 128-B row segments, LDS
// reduce over the row groups) so that a 64..128-row batch still spreads over N/32 workgroups.
constexpr int CT = 32, RG = 16;

// sum over this thread's rows m = m0, m0 + RG, ... of X[m][n] * rs[m], eight independent loads in flight per batch
// (same add order as the plain loop)
__device__ __forceinline__ float col_partial(const float* __restrict__ X, int ldx, const float* __restrict__ rs, int M,
                                             int n, int m0, float a) {
    int m = m0;
    for (; m + 7 * RG < M; m += 8 * RG) {
        float x[8], r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x[k] = X[(long)(m + k * RG) * ldx + n];
            r[k] = rs ? rs[m + k * RG] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a += x[k] * r[k];
    }
    for (; m < M; m += RG) a += X[(long)m * ldx + n] * (rs ? rs[m] : 1.f);
    return a;
}

// OPT: an output that lies inside the flat gradient buffer [gbase, gbase + numel) is a parameter gradient: its parameter
// (same offset in pbase / sqbase) takes its RMSprop step right where the sum is produced
struct ColsumOpt { const float* gbase; float* pbase; float* sqbase; long numel; float lr, alpha, eps;
                   unsigned long long* state; unsigned long long inc0, inc1; };

// colsum_multi's segment work for ONE 32-column block bx on a 256-thread workgroup (the weight-gradient launch's shape):
// thread (tx = tid & 31, ry = tid >> 5) forms the partial sums of row groups ry and ry + 8 exactly as the (32, 16)-thread
// kernel's threads (tx, ry) and (tx, ry + 8) do, and the 16 partials are added in the same order: same bits.
// red: CT * RG floats of LDS.  desc: 11 words per segment (see colsum_multi_kernel).
template <bool OPT>
__device__ __forceinline__ void colsum_multi_body256(const long long* __restrict__ desc, int nseg, const ColsumOpt& o, int bx,
                                                     float* __restrict__ red) {
    const int tx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    if constexpr (OPT) {
        if (o.state && bx == 0 && threadIdx.x == 0) { o.state[0] += o.inc0; o.state[1] += o.inc1; }
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
    const int n = (bx - (int)d[9]) * CT + tx;
    float a0 = 0.f, a1 = 0.f;
    if (n < N) {
        a0 = col_partial(X, ldx, rs, M, n, ry, a0);
        if (X2) a0 += col_partial(X2, ldx2, rs2, M, n, ry, 0.f);
        a1 = col_partial(X, ldx, rs, M, n, ry + 8, a1);
        if (X2) a1 += col_partial(X2, ldx2, rs2, M, n, ry + 8, 0.f);
    }
    __syncthreads();
    red[ry * CT + tx] = a0;
    red[(ry + 8) * CT + tx] = a1;
    __syncthreads();
    if (n < N && ry == 0) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) a += red[g * CT + tx];
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

// ---- the other small bodies a step's last launch may carry (wfd_g_end, wfd_gemm_f32_rmsprop_multi_end) -------------------
// RMSprop on up to 8 element ranges [beg, end) of the flat buffers; i = element index over the concatenated ranges
struct RmsRanges { long beg[8], end[8]; int n; };
__device__ __forceinline__ void rmsprop_ranges_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                    const RmsRanges& r, float lr, float alpha, float eps, long i) {
    for (int k = 0; k < r.n; ++k) {
        const long len = r.end[k] - r.beg[k];
        if (i < len) {
            const long j = r.beg[k] + i;
            float pj = p[j], sj = sq[j];
            rmsprop_update(pj, sj, g[j], lr, alpha, eps);
            sq[j] = sj;
            p[j] = pj;
            return;
        }
        i -= len;
    }
}

// running statistics of one BatchNorm layer over `groups` stacked batches (see bn_running_update_layers_kernel)
struct BnLayer { const float* mean; const float* m2; float* rmean; float* rvar; int N; };
__device__ __forceinline__ void bn_running_layer_body(const BnLayer& l, int n, float count, float momentum, int groups,
                                                      int nupd_last) {
    if (n >= l.N) return;
    float rm = l.rmean[n], rv = l.rvar[n];
    for (int g = 0; g < groups; ++g) {
        const float mu = l.mean[(long)g * l.N + n], uv = l.m2[(long)g * l.N + n] / (count - 1.f);
        const int nu = g == groups - 1 ? nupd_last : 1;
        for (int u = 0; u < nu; ++u) {
            rm = (1.f - momentum) * rm + momentum * mu;
            rv = (1.f - momentum) * rv + momentum * uv;
        }
    }
    l.rmean[n] = rm;
    l.rvar[n] = rv;
}
struct BnRun { BnLayer a, b; float count, momentum; int groups, nupd_last, bn_blocks; };
