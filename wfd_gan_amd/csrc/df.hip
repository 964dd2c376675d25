// Frozen DF observer (src/model.py:135-212) pieces that are not GEMM-shaped: first conv (Cin = 1), zero-padded
// max-pool (window 8, stride 8), masked cross-entropy, and the matching input-gradient kernels.
// Activations are channels-last [B][L][C], so the im2col'd convs in gemm.hip read/write them as plain
// row-major matrices; eval-mode BatchNorm is folded to a per-channel (scale, shift).
#include "common.h"

namespace {

// Y[b][l][co] = act((sum_t w[co][t] * x[b][l+t-3]) * scale[co] + shift[co]);  x row stride ldx, offset xoff.
// block = 256 threads handles 64 positions x 32 channels of one sample.
__global__ void conv1_fwd_kernel(const float* __restrict__ X, int ldx, int xoff, int L, const float* __restrict__ W,
                                 const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                 float* __restrict__ Y, int Cout) {
    __shared__ float xs[64 + 8];
    __shared__ __attribute__((aligned(16))) float ws[64 * 8];
    const int b = blockIdx.y, l0 = blockIdx.x * 64, tid = threadIdx.x;
    if (tid < 71) {
        const int l = l0 + tid - 3;
        xs[tid] = (l >= 0 && l < L) ? X[(long)b * ldx + xoff + l] : 0.f;
    }
    // taps stored [t][co]: the lanes of a wave differ in co, so a tap is read from consecutive banks
    for (int i = tid; i < Cout * 8; i += blockDim.x) ws[(i & 7) * Cout + (i >> 3)] = W[i];
    __syncthreads();
    // a thread produces 4 consecutive channels of one position: 16-byte stores, Cout / 4 threads per position
    const int q4 = Cout >> 2;
    for (int o = tid; o < 64 * q4; o += blockDim.x) {
        const int li = o / q4, c0 = 4 * (o - li * q4);
        if (l0 + li >= L) break;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float x = xs[li + t];
            const float4 w = *reinterpret_cast<const float4*>(ws + t * Cout + c0);
            a[0] = __fmaf_rn(w.x, x, a[0]); a[1] = __fmaf_rn(w.y, x, a[1]);
            a[2] = __fmaf_rn(w.z, x, a[2]); a[3] = __fmaf_rn(w.w, x, a[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = __fmaf_rn(a[k], scale[c0 + k], shift[c0 + k]);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = v > 0.f ? v : elu_expm1(v);
            a[k] = v;
        }
        *reinterpret_cast<float4*>(Y + ((long)b * L + l0 + li) * Cout + c0) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// PO[b][j][c] = max over l in [8j-3, 8j+4] of (0 <= l < L ? Y[b][l][c] : 0)   (zero padding takes part)
__global__ void pool8_fwd_kernel(const float* __restrict__ Y, int Bn, int L, int C, int Lout, float* __restrict__ PO) {
    const long total = (long)Bn * Lout * C;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        const long bj = o / C;
        const int j = (int)(bj % Lout), b = (int)(bj / Lout);
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int l = 8 * j - 3 + t;
            const float v = (l >= 0 && l < L) ? Y[((long)b * L + l) * C + c] : 0.f;
            mx = v > mx ? v : mx;
        }
        PO[o] = mx;
    }
}

// Backward of pool8 + activation + BN scale in one pass:
//   dC[b][l][c] = (l == argmax_j ? dPO[b][j][c] : 0) * act'(Y[b][l][c]) * scale[c]
// argmax = first maximum of the padded window (torch's max_pool backward); a winning pad element drops the
// gradient.  act: 1 relu (Y > 0), 2 elu (Y > 0 ? 1 : Y + 1).  Positions beyond the last window get 0.
__global__ void pool8_bwd_act_kernel(const float* __restrict__ dPO, const float* __restrict__ Y, int Bn, int L, int C,
                                     int Lout, const float* __restrict__ scale, int act, float* __restrict__ dC) {
    const long total = (long)Bn * Lout * C;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        const long bj = o / C;
        const int j = (int)(bj % Lout), b = (int)(bj / Lout);
        float mx = -INFINITY;
        int am = 0;
        float yv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int l = 8 * j - 3 + t;
            yv[t] = (l >= 0 && l < L) ? Y[((long)b * L + l) * C + c] : 0.f;
            if (yv[t] > mx) { mx = yv[t]; am = t; }
        }
        const float g = dPO[o] * scale[c];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int l = 8 * j - 3 + t;
            if (l >= 0 && l < L) {
                float d = 0.f;
                if (t == am) d = (act == 2) ? (yv[t] > 0.f ? g : g * (yv[t] + 1.f)) : (yv[t] > 0.f ? g : 0.f);
                dC[((long)b * L + l) * C + c] = d;
            }
        }
        if (j == Lout - 1)
            for (int l = 8 * j + 5; l < L; ++l) dC[((long)b * L + l) * C + c] = 0.f;
    }
}

// Masked cross-entropy over the rows the critic scored > 0 (src/mlp_df_wgan_train.py:251-263), static-shape form:
//   ce[m] = logsumexp(logits[m]) - logits[m][label];  w = sel[m] / max(1, nsel)
//   dlogits[m][c] = alpha * w * (softmax - onehot)        (alpha = f-loss weight)
// One wave per row.  nsel is read from device memory (global count over all ranks in data-parallel runs).
__global__ void masked_ce_kernel(const float* __restrict__ logits, int ldl, int C, const int* __restrict__ labels,
                                 const float* __restrict__ fv, const float* __restrict__ nsel_ptr, float alpha,
                                 float* __restrict__ ce, float* __restrict__ dlogits, int lddl) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const float* row = logits + (long)m * ldl;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(row[c] - mx);
    s = wave_sum(s);
    const float lse = mx + logf(s);
    const int lab = labels[m];
    const float sel = fv[m] > 0.f ? 1.f : 0.f;
    const float nsel = nsel_ptr[0];
    const float w = nsel > 0.f ? alpha * sel / nsel : 0.f;
    if (lane == 0) ce[m] = sel * (lse - row[lab]);
    for (int c = lane; c < C; c += 64) {
        const float p = expf(row[c] - lse);
        dlogits[(long)m * lddl + c] = w * (p - (c == lab ? 1.f : 0.f));
    }
}

// G-step scalars after the observer: f_loss = sum ce / max(1, nsel) (0 when nsel == 0, :264-267);
// out: [0] g_loss [1] f_loss [2] combined [3] nsel ; acc[0..3] += g_loss, f_loss, combined, 1
__global__ void g_loss_kernel(const float* __restrict__ ce, int B, const float* __restrict__ head, float nsel_global_scale,
                              float alpha, float* __restrict__ out, float* __restrict__ acc) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += ce[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float nsel = head[1];
        const float fl = nsel > 0.f ? s / nsel : 0.f;
        const float gl = head[0];
        out[0] = gl; out[1] = fl; out[2] = gl + alpha * fl; out[3] = nsel;
        if (acc) { acc[0] += gl; acc[1] += fl; acc[2] += gl + alpha * fl; acc[3] += 1.f; }
    }
    (void)nsel_global_scale;
}

// The three launches above plus g_head (epilogues.hip) as ONE workgroup, for the batches of the B = 64 step (<= 256 rows):
// head = [-sum fv / B, #{fv > 0}], masked cross-entropy per row (one wave per row, 16 waves), then the G-step scalars.
// Per-row arithmetic is masked_ce_kernel's; only the order of the two batch sums differs (wave partials in wave order).
__global__ __launch_bounds__(1024) void g_tail_kernel(const float* __restrict__ logits, int ldl, int B, int C,
                                                      const int* __restrict__ labels, const float* __restrict__ fv,
                                                      float invB, float alpha, float* __restrict__ ce,
                                                      float* __restrict__ dlogits, int lddl, float* __restrict__ head,
                                                      float* __restrict__ out, float* __restrict__ acc) {
    __shared__ float red[16];
    __shared__ float hd[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // every global load of the launch is issued up front (one latency, not one per dependent step): the wave's rows of logits
    // (C <= 128: two per lane and row; up to 16 rows per wave at B = 256), their labels and critic scores
    constexpr int RMAX = 16;
    float x0[RMAX], x1[RMAX], fvr[RMAX];
    int lab[RMAX];
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        const int m = wave + nw * j;
        const bool ok = m < B;
        const float* row = logits + (long)(ok ? m : 0) * ldl;
        x0[j] = (ok && lane < C) ? row[lane] : -INFINITY;
        x1[j] = (ok && lane + 64 < C) ? row[lane + 64] : -INFINITY;
        lab[j] = ok ? labels[m] : 0;
        fvr[j] = ok ? fv[m] : 0.f;
    }
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { s += fv[i]; c += fv[i] > 0.f ? 1.f : 0.f; }
    s = block_sum(s, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) { hd[0] = -s * invB; hd[1] = c; head[0] = hd[0]; head[1] = c; }
    __syncthreads();
    const float nsel = hd[1];
    float cesum = 0.f;
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        const int m = wave + nw * j;
        if (m >= B) break;                                  // uniform over the wave
        const float mx = wave_max(fmaxf(x0[j], x1[j]));
        const float e0 = lane < C ? expf(x0[j] - mx) : 0.f, e1 = lane + 64 < C ? expf(x1[j] - mx) : 0.f;
        const float lse = mx + logf(wave_sum(e0 + e1));
        const float sel = fvr[j] > 0.f ? 1.f : 0.f;
        const float w = nsel > 0.f ? alpha * sel / nsel : 0.f;
        // logit of the label: held by lane lab % 64 in x0 (lab < 64) or x1
        const float mine = lab[j] < 64 ? x0[j] : x1[j];
        const float xl = __shfl(mine, lab[j] & 63, 64);
        const float cem = sel * (lse - xl);
        if (lane == 0) { ce[m] = cem; cesum += cem; }
        if (lane < C) dlogits[(long)m * lddl + lane] = w * (expf(x0[j] - lse) - (lane == lab[j] ? 1.f : 0.f));
        if (lane + 64 < C) dlogits[(long)m * lddl + lane + 64] = w * (expf(x1[j] - lse) - (lane + 64 == lab[j] ? 1.f : 0.f));
    }
    cesum = block_sum(cesum, red);
    if (threadIdx.x == 0) {
        const float fl = nsel > 0.f ? cesum / nsel : 0.f;
        const float gl = hd[0];
        out[0] = gl; out[1] = fl; out[2] = gl + alpha * fl; out[3] = nsel;
        if (acc) { acc[0] += gl; acc[1] += fl; acc[2] += gl + alpha * fl; acc[3] += 1.f; }
    }
}

// dX[b][l] = sum_{co,t} W[co][t] * dC[b][l - t + 3][co]   (input gradient of the Cin = 1 conv)
// block: one sample x 64 positions; LDS holds dC rows l0-4 .. l0+66.
__global__ void conv1_bwd_kernel(const float* __restrict__ dC, int L, int Cout, const float* __restrict__ W,
                                 float* __restrict__ dX, int lddx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int CS = Cout + 4;           // padded row (Cout % 4 == 0): 16-byte aligned for ds_write_b128, and the 16 positions x
                                       // 4 channel phases a wave reads at a time fall in 64 different banks (36 l + q mod 64)
    float* ds = sm;                    // [71][Cout + 4]
    float* ws = sm + 71 * CS;          // [Cout][8]
    const int b = blockIdx.y, l0 = blockIdx.x * 64, tid = threadIdx.x;
    const int q4 = Cout >> 2;
    for (int i = tid; i < 71 * q4; i += blockDim.x) {          // 16-byte loads, 16-byte stores into the padded rows
        const int r = i / q4, c0 = 4 * (i - r * q4);
        const int l = l0 - 4 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l >= 0 && l < L) v = *reinterpret_cast<const float4*>(dC + ((long)b * L + l) * Cout + c0);
        *reinterpret_cast<float4*>(ds + r * CS + c0) = v;
    }
    for (int i = tid; i < Cout * 8; i += blockDim.x) ws[i] = W[i];
    __syncthreads();
    // 4 lanes per position, each a quarter of the channels
    const int li = tid >> 2, q = tid & 3;
    float a = 0.f;
    for (int co = q; co < Cout; co += 4) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // source row l - t + 3  ->  local row (li - t + 3) + 4
            a = __fmaf_rn(ws[co * 8 + t], ds[(li - t + 7) * CS + co], a);
        }
    }
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (q == 0 && l0 + li < L) dX[(long)b * lddx + l0 + li] = a;
}

}  // namespace

extern "C" {

int wfd_df_conv1_fwd(const float* X, int ldx, int xoff, int Bn, int L, const float* W, const float* scale,
                     const float* shift, int act, float* Y, int Cout, void* stream) {
    if (!X || !W || !scale || !shift || !Y || Cout > 64 || Cout <= 0 || (Cout & 3)) return WFD_ERR_ARG;
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3(cdiv(L, 64), Bn), dim3(256), 0, (hipStream_t)stream, X, ldx, xoff, L, W,
                       scale, shift, act, Y, Cout);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_df_pool8_fwd(const float* Y, int Bn, int L, int C, float* PO, void* stream) {
    if (!Y || !PO) return WFD_ERR_ARG;
    const int Lout = (L + 7) / 8;
    const long total = (long)Bn * Lout * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pool8_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, Y, Bn, L, C, Lout, PO);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_df_pool8_bwd_act(const float* dPO, const float* Y, int Bn, int L, int C, const float* scale, int act,
                         float* dC, void* stream) {
    if (!dPO || !Y || !scale || !dC) return WFD_ERR_ARG;
    const int Lout = (L + 7) / 8;
    const long total = (long)Bn * Lout * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pool8_bwd_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dPO, Y, Bn, L, C, Lout,
                       scale, act, dC);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_masked_ce(const float* logits, int ldl, int M, int C, const int* labels, const float* fv, const float* nsel,
                  float alpha, float* ce, float* dlogits, int lddl, void* stream) {
    if (!logits || !labels || !fv || !nsel || !ce || !dlogits) return WFD_ERR_ARG;
    hipLaunchKernelGGL(masked_ce_kernel, dim3(M), dim3(64), 0, (hipStream_t)stream, logits, ldl, C, labels, fv, nsel,
                       alpha, ce, dlogits, lddl);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_g_tail(const float* logits, int ldl, int B, int C, const int* labels, const float* fv, float invB, float alpha,
               float* ce, float* dlogits, int lddl, float* head, float* out, float* acc, void* stream) {
    if (!logits || !labels || !fv || !ce || !dlogits || !head || !out || B < 1 || B > 256 || C < 1 || C > 128) return WFD_ERR_ARG;
    hipLaunchKernelGGL(g_tail_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, ldl, B, C, labels, fv, invB,
                       alpha, ce, dlogits, lddl, head, out, acc);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_g_loss(const float* ce, int B, const float* head, float alpha, float* out, float* acc, void* stream) {
    if (!ce || !head || !out) return WFD_ERR_ARG;
    hipLaunchKernelGGL(g_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ce, B, head, 1.f, alpha, out, acc);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_df_conv1_bwd(const float* dC, int Bn, int L, int Cout, const float* W, float* dX, int lddx, void* stream) {
    if (!dC || !W || !dX || Cout > 64 || (Cout & 3)) return WFD_ERR_ARG;
    const size_t sm = (size_t)(71 * (Cout + 4) + 8 * Cout) * sizeof(float);
    hipLaunchKernelGGL(conv1_bwd_kernel, dim3(cdiv(L, 64), Bn), dim3(256), sm, (hipStream_t)stream, dC, L, Cout, W, dX,
                       lddx);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

}  // extern "C"
