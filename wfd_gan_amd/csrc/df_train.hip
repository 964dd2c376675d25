// Kernels of the DF observer's TRAINING step (src/train_df.py:111-131 around src/model.py:135-206 in train mode):
// BatchNorm1d over channels-last conv activations (statistics over all B*L rows of a channel), its backward, the
// weight gradient of the Cin = 1 first conv, the tap-flipped transpose that the input-gradient conv reads, dropout
// as a mask multiply, tall column sums (conv bias gradients) and Adamax.  The contractions (convs, their input and
// weight gradients, the FC layers) run on the MFMA GEMM core in gemm.hip.
//
// Tall reductions are two-stage and deterministic: a grid of row chunks writes per-chunk partials, a second tiny
// kernel combines them in chunk order (Chan's pairwise update for mean / M2).
#include "common.h"
#include <mutex>

namespace {

constexpr int TC = 32, TR = 8;          // 32 channels x 8 row groups per workgroup

__device__ __forceinline__ float tr_reduce(float v, float (*red)[TC]) {
    __syncthreads();
    red[threadIdx.y][threadIdx.x] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int g = 0; g < TR; ++g) r += red[g][threadIdx.x];
    return r;
}

// Z[m][n] = sum of slabs + bias[n]  (optional materialisation of the conv output), then per-chunk statistics
//   part[chunk][0][n] = mean of the chunk's rows, part[chunk][1][n] = sum (z - mean)^2 over the chunk
__global__ void bn_tall_partial_kernel(const float* __restrict__ Z, int ldz, int M, int N, int rows_per_chunk,
                                       float* __restrict__ part) {
    __shared__ float red[TR][TC];
    const int n = blockIdx.x * TC + threadIdx.x, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const bool ok = n < N;
    float s = 0.f;
    if (ok)
        for (int m = r0 + threadIdx.y; m < r1; m += TR) s += Z[(long)m * ldz + n];
    const float mu = tr_reduce(s, red) / (float)(r1 - r0);
    float q = 0.f;
    if (ok)
        for (int m = r0 + threadIdx.y; m < r1; m += TR) {
            const float d = Z[(long)m * ldz + n] - mu;
            q = __fmaf_rn(d, d, q);
        }
    q = tr_reduce(q, red);
    if (ok && threadIdx.y == 0) {
        part[((long)chunk * 2 + 0) * N + n] = mu;
        part[((long)chunk * 2 + 1) * N + n] = q;
    }
}

// combine the chunk statistics (Chan et al.): 32 columns x 16 groups per workgroup; group g folds chunks g, g+16, ...
// in order (loads batched eight at a time), then the 16 group results are folded in order by group 0 -- a fixed
// association, so the result is deterministic.  mean, invstd = 1/sqrt(M2/M + eps); running stats with the unbiased
// variance, as nn.BatchNorm1d does in training mode.
constexpr int CG = 16;
__device__ __forceinline__ void chan_fold(float& mu, float& m2, float& cnt, float mb, float qb, float cb) {
    if (cb <= 0.f) return;
    const float tot = cnt + cb, delta = mb - mu;
    mu += delta * (cb / tot);
    m2 += qb + delta * delta * (cnt * cb / tot);
    cnt = tot;
}
__global__ void bn_tall_combine_kernel(const float* __restrict__ part, int nchunk, int rows_per_chunk, int M, int N,
                                       float eps, float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                       float* __restrict__ rmean, float* __restrict__ rvar) {
    __shared__ float sh[3][CG][TC];
    const int n = blockIdx.x * TC + threadIdx.x, g = threadIdx.y;
    const bool ok = n < N;
    float mu = 0.f, m2 = 0.f, cnt = 0.f;
    if (ok)
        for (int c0 = g; c0 < nchunk; c0 += 8 * CG) {
            float mb[8], qb[8], cb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k * CG;
                const bool in = c < nchunk;
                const int cc = in ? c : g;
                mb[k] = part[((long)cc * 2 + 0) * N + n];
                qb[k] = part[((long)cc * 2 + 1) * N + n];
                cb[k] = in ? (float)min(rows_per_chunk, M - c * rows_per_chunk) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) chan_fold(mu, m2, cnt, mb[k], qb[k], cb[k]);
        }
    sh[0][g][threadIdx.x] = mu; sh[1][g][threadIdx.x] = m2; sh[2][g][threadIdx.x] = cnt;
    __syncthreads();
    if (!ok || g != 0) return;
    for (int k = 1; k < CG; ++k) chan_fold(mu, m2, cnt, sh[0][k][threadIdx.x], sh[1][k][threadIdx.x], sh[2][k][threadIdx.x]);
    mean[n] = mu;
    invstd[n] = 1.0f / sqrtf(m2 / (float)M + eps);
    if (rmean) {
        rmean[n] = (1.f - momentum) * rmean[n] + momentum * mu;
        rvar[n] = (1.f - momentum) * rvar[n] + momentum * (m2 / ((float)M - 1.f));
    }
}

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : elu_expm1(v);
    return v;
}

// Y = act(gamma * (Z - mean) * invstd + beta), flat float4 (N % 4 == 0)
__global__ void bn_tall_apply_kernel(const float* __restrict__ Z, long total4, int N, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int act, float* __restrict__ Y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)((i * 4) % N);
        const float4 z = reinterpret_cast<const float4*>(Z)[i];
        const float zz[4] = {z.x, z.y, z.z, z.w};
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            y[k] = act_f(__fmaf_rn((zz[k] - mean[n + k]) * invstd[n + k], gamma[n + k], beta[n + k]), act);
        reinterpret_cast<float4*>(Y)[i] = make_float4(y[0], y[1], y[2], y[3]);
    }
}

// backward partials over a chunk of rows: part[chunk][0][n] = sum dA, part[chunk][1][n] = sum dA * xhat
__global__ void bn_tall_bwd_partial_kernel(const float* __restrict__ dA, const float* __restrict__ Z, int ld, int M, int N,
                                           int rows_per_chunk, const float* __restrict__ mean,
                                           const float* __restrict__ invstd, float* __restrict__ part) {
    __shared__ float red[TR][TC];
    const int n = blockIdx.x * TC + threadIdx.x, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const bool ok = n < N;
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
        const float mu = mean[n], is = invstd[n];
        for (int m = r0 + threadIdx.y; m < r1; m += TR) {
            const float d = dA[(long)m * ld + n];
            s0 += d;
            s1 = __fmaf_rn(d, (Z[(long)m * ld + n] - mu) * is, s1);
        }
    }
    s0 = tr_reduce(s0, red);
    s1 = tr_reduce(s1, red);
    if (ok && threadIdx.y == 0) {
        part[((long)chunk * 2 + 0) * N + n] = s0;
        part[((long)chunk * 2 + 1) * N + n] = s1;
    }
}

// out[j] = sum over chunks of part[chunk][j]   (j < width): 32 columns x 16 groups per workgroup, group g adds chunks
// g, g+16, ... in order (eight loads in flight), the groups are added in order through LDS (deterministic)
__global__ void sum_partials_kernel(const float* __restrict__ part, int nchunk, int width, float* __restrict__ out,
                                    float* __restrict__ out2, int split) {
    __shared__ float sh[CG][TC];
    const int j = blockIdx.x * TC + threadIdx.x, g = threadIdx.y;
    float a = 0.f;
    if (j < width)
        for (int c0 = g; c0 < nchunk; c0 += 8 * CG) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + k * CG;
                v[k] = part[(long)(c < nchunk ? c : g) * width + j];
                if (c >= nchunk) v[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) a += v[k];
        }
    sh[g][threadIdx.x] = a;
    __syncthreads();
    if (j >= width || g != 0) return;
    for (int k = 1; k < CG; ++k) a += sh[k][threadIdx.x];
    out[j] = a;
    // optional copy of the first `split` entries to a second destination (dbeta = sums[0][:])
    if (out2 && j < split) out2[j] = a;
}

// dZ = gamma * invstd * (dA - s0/M - xhat * s1/M), flat float4; sums = [s0[N] ; s1[N]]
__global__ void bn_tall_bwd_apply_kernel(const float* __restrict__ dA, const float* __restrict__ Z, long total4, int N,
                                         float invM, const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ gamma, const float* __restrict__ sums,
                                         float* __restrict__ dZ) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)((i * 4) % N);
        const float4 d = reinterpret_cast<const float4*>(dA)[i], z = reinterpret_cast<const float4*>(Z)[i];
        const float dd[4] = {d.x, d.y, d.z, d.w}, zz[4] = {z.x, z.y, z.z, z.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float is = invstd[n + k], xh = (zz[k] - mean[n + k]) * is;
            o[k] = gamma[n + k] * is * (dd[k] - sums[n + k] * invM - xh * sums[N + n + k] * invM);
        }
        reinterpret_cast<float4*>(dZ)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// column sums of a tall matrix, per chunk: part[chunk][n] = sum_{rows of chunk} X[m][n]
__global__ void colsum_tall_partial_kernel(const float* __restrict__ X, int ld, int M, int N, int rows_per_chunk,
                                           float* __restrict__ part) {
    __shared__ float red[TR][TC];
    const int n = blockIdx.x * TC + threadIdx.x, chunk = blockIdx.y;
    const int r0 = chunk * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    if (n < N)
        for (int m = r0 + threadIdx.y; m < r1; m += TR) s += X[(long)m * ld + n];
    s = tr_reduce(s, red);
    if (n < N && threadIdx.y == 0) part[(long)chunk * N + n] = s;
}

// first conv (Cin = 1) weight gradient partials: block (b, 128-position chunk) ->
//   part[blk][co*8 + t] = sum_l dZ[b][l][co] * x[b][l + t - 3]
// thread = (co, t) pair (Cout * 8 <= 1024 threads); the dZ chunk and the x window are staged in LDS with coalesced
// loads, the products run out of LDS
constexpr int W1CH = 128;
__global__ void conv1_wgrad_partial_kernel(const float* __restrict__ dZ, const float* __restrict__ X, int ldx, int xoff,
                                           int L, int Cout, float* __restrict__ part) {
    extern __shared__ float sm[];
    float* xs = sm;                        // x[l0 - 3 .. l0 + W1CH + 4)
    float* ds = sm + W1CH + 8;             // dZ[b][l0 .. l0 + W1CH)[Cout], row stride Cout + 1 (bank spread over t)
    const int b = blockIdx.y, l0 = blockIdx.x * W1CH, l1 = min(L, l0 + W1CH), nl = l1 - l0;
    for (int i = threadIdx.x; i < W1CH + 7; i += blockDim.x) {
        const int l = l0 - 3 + i;
        xs[i] = (l >= 0 && l < L) ? X[(long)b * ldx + xoff + l] : 0.f;
    }
    for (int i = threadIdx.x; i < nl * Cout; i += blockDim.x) {
        const int r = i / Cout, co = i - r * Cout;
        ds[r * (Cout + 1) + co] = dZ[((long)b * L + l0) * Cout + i];
    }
    __syncthreads();
    const int co = threadIdx.x >> 3, t = threadIdx.x & 7;
    float a = 0.f;
    if (co < Cout)
        for (int r = 0; r < nl; ++r) a = __fmaf_rn(ds[r * (Cout + 1) + co], xs[r + t], a);
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    if (co < Cout) part[blk * (Cout * 8) + threadIdx.x] = a;
}

// WT[ci][t'*Cout + co] = W[co][(7 - t')*Cin + ci]   (tap-flipped transpose read by the input-gradient conv)
__global__ void conv_wflip_kernel(const float* __restrict__ W, int Cout, int Cin, float* __restrict__ WT) {
    const long total = (long)Cout * Cin * 8;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int co = (int)(o % Cout);
        const long r = o / Cout;
        const int tp = (int)(r % 8), ci = (int)(r / 8);
        WT[o] = W[(long)co * (8 * Cin) + (7 - tp) * Cin + ci];
    }
}

__global__ void mul_mask_kernel(const float* __restrict__ X, const float* __restrict__ mask, long n,
                                float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = X[i] * mask[i];
}

// torch.optim.Adamax defaults (lr 2e-3, betas (.9,.999), eps 1e-8, no weight decay), one flat pass:
//   m = b1 m + (1-b1) g ; u = max(b2 u, |g| + eps) ; p -= lr / (1 - b1^t) * m / u        (t = step[0] + 1)
// the step counter lives on the device so that a captured graph advances it on replay (block 0 increments it last)
__global__ void adamax_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                              float* __restrict__ u, long n, float lr, float b1, float b2, float eps,
                              const long long* __restrict__ step) {
    const float t = (float)(step[0] + 1);
    const float clr = lr / (1.f - powf(b1, t));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i];
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float uu = fmaxf(b2 * u[i], fabsf(gg) + eps);
        m[i] = mm;
        u[i] = uu;
        p[i] -= clr * mm / uu;
    }
}

__global__ void step_inc_kernel(long long* step) { if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1; }

// argmax over C logits per row -> correct[0] += (argmax == label)   (accuracy passes, src/train_df.py:136-157)
__global__ void argmax_correct_kernel(const float* __restrict__ logits, int ldl, int M, int C,
                                      const int* __restrict__ labels, int* __restrict__ correct) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* row = logits + (long)m * ldl;
    int best = 0;
    float bv = row[0];
    for (int c = 1; c < C; ++c)
        if (row[c] > bv) { bv = row[c]; best = c; }       // first maximum, as torch.max
    if (best == labels[m]) atomicAdd(correct, 1);
}

inline unsigned flat_blocks(long n) {
    long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

// ---- wfd_colsum_tall for callers that cannot allocate (HIP-graph capture, serving loops) -------------------------------
// The chunk partials live in a file-scope device array instead of a caller workspace.  `scratch` below is the HOST-side
// stand-in of that array: only hipGetSymbolAddress turns it into this device's address, once per device (every GPU holds
// its own copy at its own address), and from then on host code uses the looked-up pointer, never the array's name.
constexpr long SCRATCH_FLOATS = 1L << 20;          // 1 Mi floats = 4 MiB on every GPU the library runs on, held until process exit
static __device__ __align__(256) float scratch[SCRATCH_FLOATS];
constexpr int SCRATCH_MAX_DEVICES = 64;
static float* g_scratch_ptr[SCRATCH_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

// device pointer of the CURRENT device's copy of `scratch` (looked up on the first call on that device)
static int scratch_of_current_device(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCRATCH_MAX_DEVICES) return WFD_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return WFD_ERR_LAUNCH;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return WFD_OK;
}

extern "C" {

int wfd_bn_tall_stats(const float* Z, int ldz, int M, int N, int rows_per_chunk, float* part, float eps, float momentum,
                      float* mean, float* invstd, float* rmean, float* rvar, void* stream) {
    if (!Z || !part || !mean || !invstd || M < 2 || N <= 0 || rows_per_chunk <= 0) return WFD_ERR_ARG;
    const int nchunk = cdiv(M, rows_per_chunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_tall_partial_kernel, dim3(cdiv(N, TC), nchunk), dim3(TC, TR), 0, st, Z, ldz, M, N,
                       rows_per_chunk, part);
    WFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_tall_combine_kernel, dim3(cdiv(N, TC)), dim3(TC, CG), 0, st, part, nchunk, rows_per_chunk, M, N,
                       eps, momentum, mean, invstd, rmean, rvar);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_tall_apply(const float* Z, int M, int N, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, int act, float* Y, void* stream) {
    if (!Z || !mean || !invstd || !gamma || !beta || !Y || M <= 0 || N <= 0 || (N & 3)) return WFD_ERR_ARG;
    const long total4 = (long)M * N / 4;
    hipLaunchKernelGGL(bn_tall_apply_kernel, dim3(flat_blocks(total4)), dim3(256), 0, (hipStream_t)stream, Z, total4, N,
                       mean, invstd, gamma, beta, act, Y);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_bn_tall_bwd(const float* dA, const float* Z, int M, int N, int rows_per_chunk, float* part, const float* mean,
                    const float* invstd, const float* gamma, float* sums, float* dgamma, float* dbeta, float* dZ,
                    void* stream) {
    if (!dA || !Z || !part || !mean || !invstd || !gamma || !sums || !dgamma || !dbeta || !dZ) return WFD_ERR_ARG;
    if (M <= 0 || N <= 0 || (N & 3) || rows_per_chunk <= 0) return WFD_ERR_ARG;
    const int nchunk = cdiv(M, rows_per_chunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_tall_bwd_partial_kernel, dim3(cdiv(N, TC), nchunk), dim3(TC, TR), 0, st, dA, Z, N, M, N,
                       rows_per_chunk, mean, invstd, part);
    WFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(2 * N, TC)), dim3(TC, CG), 0, st, part, nchunk, 2 * N, sums, dbeta, N);
    WFD_CHECK_LAUNCH();
    // dgamma = sums[N .. 2N): a device-to-device copy node keeps the call graph-capturable
    if (hipMemcpyAsync(dgamma, sums + N, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return WFD_ERR_LAUNCH;
    const long total4 = (long)M * N / 4;
    hipLaunchKernelGGL(bn_tall_bwd_apply_kernel, dim3(flat_blocks(total4)), dim3(256), 0, st, dA, Z, total4, N,
                       1.0f / (float)M, mean, invstd, gamma, sums, dZ);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_colsum_tall(const float* X, int ld, int M, int N, int rows_per_chunk, float* part, float* out, void* stream) {
    if (!X || !part || !out || M <= 0 || N <= 0 || rows_per_chunk <= 0) return WFD_ERR_ARG;
    const int nchunk = cdiv(M, rows_per_chunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_tall_partial_kernel, dim3(cdiv(N, TC), nchunk), dim3(TC, TR), 0, st, X, ld, M, N,
                       rows_per_chunk, part);
    WFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(N, TC)), dim3(TC, CG), 0, st, part, nchunk, N, out, (float*)nullptr, 0);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_static_scratch_floats(void) { return (int)SCRATCH_FLOATS; }

int wfd_colsum_tall_static(const float* X, int ld, int M, int N, int rows_per_chunk, float* out, void* stream) {
    if (!X || !out || M <= 0 || N <= 0 || rows_per_chunk <= 0 || ld < N) return WFD_ERR_ARG;
    const long need = (long)cdiv(M, rows_per_chunk) * N;
    if (need > SCRATCH_FLOATS) return WFD_ERR_WORKSPACE;          // beyond the static array: use wfd_colsum_tall
    // `scratch` exists once per GPU: the kernels must run on the device whose copy is handed to them.  A non-null stream
    // names its device; it has to be the current one (the null stream always is).
    if (stream) {
        int cur = -1;
        hipDevice_t sdev = -1;
        if (hipGetDevice(&cur) != hipSuccess || hipStreamGetDevice((hipStream_t)stream, &sdev) != hipSuccess) return WFD_ERR_LAUNCH;
        if ((int)sdev != cur) return WFD_ERR_ARG;
    }
    float* part = nullptr;
    const int rc = scratch_of_current_device(&part);
    if (rc != WFD_OK) return rc;
    // every partial the second kernel reads is written by the first one: nothing has to start at zero
    const int nchunk = cdiv(M, rows_per_chunk);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_tall_partial_kernel, dim3(cdiv(N, TC), nchunk), dim3(TC, TR), 0, st, X, ld, M, N,
                       rows_per_chunk, part);
    WFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(N, TC)), dim3(TC, CG), 0, st, part, nchunk, N, out, (float*)nullptr, 0);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_conv1_wgrad(const float* dZ, const float* X, int ldx, int xoff, int Bn, int L, int Cout, float* part, float* dW,
                    void* stream) {
    if (!dZ || !X || !part || !dW || Bn <= 0 || L <= 0 || Cout <= 0 || Cout * 8 > 1024) return WFD_ERR_ARG;
    const int nlc = cdiv(L, W1CH);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(W1CH + 8 + W1CH * (Cout + 1)) * sizeof(float);
    if (lds > 60000) return WFD_ERR_ARG;
    hipLaunchKernelGGL(conv1_wgrad_partial_kernel, dim3(nlc, Bn), dim3(Cout * 8), lds, st, dZ, X, ldx, xoff, L, Cout,
                       part);
    WFD_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(Cout * 8, TC)), dim3(TC, CG), 0, st, part, nlc * Bn, Cout * 8, dW,
                       (float*)nullptr, 0);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_conv_wflip(const float* W, int Cout, int Cin, float* WT, void* stream) {
    if (!W || !WT || Cout <= 0 || Cin <= 0) return WFD_ERR_ARG;
    hipLaunchKernelGGL(conv_wflip_kernel, dim3(flat_blocks((long)Cout * Cin * 8)), dim3(256), 0, (hipStream_t)stream, W,
                       Cout, Cin, WT);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_mul_mask(const float* X, const float* mask, long n, float* out, void* stream) {
    if (!X || !mask || !out || n <= 0) return WFD_ERR_ARG;
    hipLaunchKernelGGL(mul_mask_kernel, dim3(flat_blocks(n)), dim3(256), 0, (hipStream_t)stream, X, mask, n, out);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_adamax(float* p, const float* g, float* m, float* u, long n, float lr, float beta1, float beta2, float eps,
               long long* step, int advance, void* stream) {
    if (!p || !g || !m || !u || !step || n <= 0) return WFD_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adamax_kernel, dim3(flat_blocks(n)), dim3(256), 0, st, p, g, m, u, n, lr, beta1, beta2, eps, step);
    WFD_CHECK_LAUNCH();
    if (advance) {
        hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, st, step);
        WFD_CHECK_LAUNCH();
    }
    return WFD_OK;
}

int wfd_argmax_correct(const float* logits, int ldl, int M, int C, const int* labels, int* correct, void* stream) {
    if (!logits || !labels || !correct || M <= 0 || C <= 0 || ldl < C) return WFD_ERR_ARG;
    hipLaunchKernelGGL(argmax_correct_kernel, dim3(cdiv(M, 128)), dim3(128), 0, (hipStream_t)stream, logits, ldl, M, C,
                       labels, correct);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

}  // extern "C"
