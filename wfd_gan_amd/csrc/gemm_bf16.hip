// bf16-operand MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the large-batch forward path of the
// generator (bulk sampling) -- the "bf16 operands" half of north_star's MFMA clause.  NT form only:
//   C[m][n] = sum_k A[m][k] * B[n][k],  A (M,K) and B (N,K) bf16 row-major, K % 8 == 0, rows 16-byte aligned
// (activations x nn.Linear weights, src/model.py:28-32).  Results differ from the fp32 path at bf16 rounding level
// (~1e-2 relative on activations); the fp32 kernels in gemm.hip remain the parity path.
//
// Tile 128x128x64 per 256-thread workgroup, 2x2 waves of 64x64 (2x2 accumulators of 32x32).  Operand tiles are
// staged global -> registers -> LDS, double buffered, rows padded to 72 bf16 (144 B: 9 mod 16 sixteen-byte slots, so
// the ds_read_b128 fragment reads of 16 consecutive rows are conflict-free).  Lane (i = lane & 31, h = lane >> 5)
// of k-step s holds A[row i][k = 16 s + 8 h + 0..7] and B[col i][same k].  Workgroup ids are remapped so that the
// workgroups of one XCD walk neighbouring tiles (shared operand panels stay in that XCD's L2).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TM = 128, TN = 128, TK = 64, NTH = 256;
constexpr int LDS_ROW = TK + 8;                 // 72 bf16 = 144 B
constexpr int TILE_E = 128 * LDS_ROW;           // elements per operand tile

struct Bf16Args {
    const __bf16* A;
    const __bf16* B;
    int M, N, K, lda, ldb;
    float* P;          // raw fp32 output (EPI = 0)
    int ldp;
    __bf16* O;         // bf16 output (EPI = 1): act(acc * scale[n] + shift[n])
    int ldo;
    const float* scale;
    const float* shift;
    int act;
    int gx, gy;        // tiles along N / M
};

__device__ __forceinline__ void load_rows(uint4 (&r)[4], const __bf16* __restrict__ G, int ld, int row0, int nrows,
                                          int k0, int K, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + NTH * j, rr = c >> 3, kc = c & 7;
        const int gr = row0 + rr, gk = k0 + 8 * kc;
        r[j] = make_uint4(0u, 0u, 0u, 0u);
        if (gr < nrows && gk < K) r[j] = *reinterpret_cast<const uint4*>(G + (long)gr * ld + gk);
    }
}

__device__ __forceinline__ void store_rows(const uint4 (&r)[4], __bf16* __restrict__ S, int tid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + NTH * j, rr = c >> 3, kc = c & 7;
        *reinterpret_cast<uint4*>(S + rr * LDS_ROW + 8 * kc) = r[j];
    }
}

template <int EPI>
__global__ __launch_bounds__(NTH) void gemm_bf16_nt_kernel(Bf16Args g) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];   // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
    // XCD-aware order: ids b and b+8 share an XCD, so give each XCD a contiguous run of tiles
    int bid = blockIdx.x;
    const int nb = g.gx * g.gy;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    const int m0 = (bid / g.gx) * TM, n0 = (bid % g.gx) * TN;
    const int nk = (g.K + TK - 1) / TK;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    uint4 ra[4], rb[4];
    load_rows(ra, g.A, g.lda, m0, g.M, 0, g.K, tid);
    load_rows(rb, g.B, g.ldb, n0, g.N, 0, g.K, tid);
    store_rows(ra, smem, tid);
    store_rows(rb, smem + TILE_E, tid);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_rows(ra, g.A, g.lda, m0, g.M, (kt + 1) * TK, g.K, tid);
            load_rows(rb, g.B, g.ldb, n0, g.N, (kt + 1) * TK, g.K, tid);
        }
        const __bf16* a_s = smem + cur * 2 * TILE_E;
        const __bf16* b_s = a_s + TILE_E;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const bf16x8*>(a_s + (wm * 64 + t * 32 + i) * LDS_ROW + 16 * s + 8 * h);
                bf[t] = *reinterpret_cast<const bf16x8*>(b_s + (wn * 64 + t * 32 + i) * LDS_ROW + 16 * s + 8 * h);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) {
            store_rows(ra, smem + (cur ^ 1) * 2 * TILE_E, tid);
            store_rows(rb, smem + (cur ^ 1) * 2 * TILE_E + TILE_E, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // C/D map: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + b * 32 + i;
            if (col >= g.N) continue;
            float sc = 1.f, sh = 0.f;
            if (EPI == 1) { sc = g.scale ? g.scale[col] : 1.f; sh = g.shift ? g.shift[col] : 0.f; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.M) continue;
                float v = acc[a][b][r];
                if (EPI == 1) {
                    v = __fmaf_rn(v, sc, sh);
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 3) v = 1.0f / (1.0f + expf(-v));
                    g.O[(long)row * g.ldo + col] = (__bf16)v;
                } else {
                    g.P[(long)row * g.ldp + col] = v;
                }
            }
        }
}

// fp32 -> bf16 (round to nearest even) with zero padding of columns [N, Npad)
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, int ldi, __bf16* __restrict__ out, int ldo, int M, int N,
                                   int Npad) {
    const long total = (long)M * Npad;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int m = (int)(o / Npad), n = (int)(o - (long)m * Npad);
        out[(long)m * ldo + n] = (__bf16)(n < N ? in[(long)m * ldi + n] : 0.f);
    }
}

}  // namespace

extern "C" {

int wfd_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* P, int ldp, void* O,
                     int ldo, const float* scale, const float* shift, int act, void* stream) {
    if (!A || !B || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) return WFD_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return WFD_ERR_ARG;
    if ((!P && !O) || (P && ldp < N) || (O && ldo < N)) return WFD_ERR_ARG;
    Bf16Args g{};
    g.A = (const __bf16*)A; g.B = (const __bf16*)B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.P = P; g.ldp = ldp; g.O = (__bf16*)O; g.ldo = ldo; g.scale = scale; g.shift = shift; g.act = act;
    g.gx = cdiv(N, TN); g.gy = cdiv(M, TM);
    const size_t lds = (size_t)4 * TILE_E * sizeof(__bf16);
    hipStream_t st = (hipStream_t)stream;
    if (O) {
        WFD_LDS_ATTR((gemm_bf16_nt_kernel<1>), lds);
        hipLaunchKernelGGL(gemm_bf16_nt_kernel<1>, dim3(g.gx * g.gy), dim3(NTH), lds, st, g);
    } else {
        WFD_LDS_ATTR((gemm_bf16_nt_kernel<0>), lds);
        hipLaunchKernelGGL(gemm_bf16_nt_kernel<0>, dim3(g.gx * g.gy), dim3(NTH), lds, st, g);
    }
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int wfd_f32_to_bf16(const float* in, int ldi, void* out, int ldo, int M, int N, int Npad, void* stream) {
    if (!in || !out || M <= 0 || N <= 0 || Npad < N || ldo < Npad || ldi < N) return WFD_ERR_ARG;
    long blocks = ((long)M * Npad + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ldi,
                       (__bf16*)out, ldo, M, N, Npad);
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

}  // extern "C"
