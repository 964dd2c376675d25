/* wfdgan.h -- C ABI of libwfdgan_hip.so: MI355X (gfx950) kernels for the wfd-gan training / sampling hot path.
 *
 * The reference (websitefingerprinting/wfd-gan) is pure Python: it has no FFI of its own, its "operator
 * interface" for this path is the ATen ops that src/model.py and src/mlp_df_wgan_train.py dispatch.  Each entry
 * point below replaces the cited op sequence.  Citations are file:line under the reference's root.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller; nothing here allocates.
 *   - `stream` is a hipStream_t (0 = default stream); all calls are asynchronous and graph-capturable.
 *   - return value: 0 ok, 1 bad argument, 2 launch failure.
 *   - matrices are row-major fp32 with an explicit leading dimension (floats).
 *   - "slabs": a GEMM writes `ns` partial products P[z][M][ldp] (z < ns, `ss` floats apart); the epilogue
 *     entry points take (P, ns, ss, ldp) and sum the slabs in z order while applying their math.
 *
 * The Python binding (wfd_gan_amd/_lib.py) parses THIS file to build its ctypes signatures: keep one
 * declaration per statement, types from {int, long, float, double, unsigned long long, T*}.
 */
#ifndef WFDGAN_H
#define WFDGAN_H
#ifdef __cplusplus
extern "C" {
#endif

int wfd_version(void);

/* ---- dense contractions (fp32 MFMA, exact f32 products) ------------------------------------------------ */

/* Operand mode of every contraction entry point below (per HOST THREAD, so threads driving plans of different modes do
 * not race; default 0 in every thread; read when a launch is issued or captured):
 *   0  f32-input MFMA, exact f32 products -- the parity path and the default;
 *   1  "bf16":   fp32 data converted to bf16 on its way into LDS, bf16 MFMA, fp32 accumulate (BASELINE config 4);
 *   2  "bf16x3": x = hi + lo split into two bf16, product = hi*hi + hi*lo + lo*hi on the bf16 pipe (products carry
 *                ~1e-5 relative error instead of 6e-8, at 16/3 of the f32 MFMA rate).
 * Buffers in HBM stay fp32 in every mode.  No counterpart in the reference (torch's default fp32 matmul = mode 0). */
int wfd_set_gemm_mode(int mode);
int wfd_get_gemm_mode(void);

/* Split count the library recommends for an MxNxK product (fills 256 CUs when the batch dimension is tiny). */
int wfd_gemm_pick_splits(int M, int N, int K);
/* ... for a given operand mode (the bf16-pipe core of modes 1 / 2 uses 128x128 tiles on large products) */
int wfd_gemm_pick_splits_mode(int M, int N, int K, int mode);

/* layout 0 (NT): P = A[M][K] * B[N][K]^T   -- nn.Linear forward (torch addmm), src/model.py:28-32,57-66,187-196
 * layout 1 (NN): P = A[M][K] * B[K][N]     -- input-gradient of nn.Linear: autograd.grad / backward,
 *                                             src/mlp_df_wgan_train.py:204-206,210-212,229,273
 * layout 2 (TN): P = A[K][M]^T * B[K][N]   -- weight-gradient of nn.Linear (reduction over batch rows), :229,:273 */
int wfd_gemm_f32(int layout, const float* A, int lda, const float* B, int ldb, float* P, int ldp, long slab_stride,
                 int M, int N, int K, int splits, void* stream);

/* Two INDEPENDENT products of the backward in one launch: problem 1 = a weight gradient (layout 2, TN, written in full to
 * P1), problem 2 = the next activation-gradient product of the chain (layout2 0 = NT or 1 = NN, into splits2 slabs) --
 * the pairs of src/mlp_df_wgan_train.py:229,273 that autograd also runs independently.  f32 operand mode only, 16-byte
 * aligned operands, leading dimensions multiples of 4, N > 32; returns 1 otherwise (launch them one by one then). */
int wfd_gemm_f32_pair(const float* A1, int lda1, const float* B1, int ldb1, float* P1, int ldp1, int M1, int N1, int K1,
                      int layout2, const float* A2, int lda2, const float* B2, int ldb2, float* P2, int ldp2,
                      long slab_stride2, int M2, int N2, int K2, int splits2, void* stream);

/* k=8 stride-1 "SAME" Conv1d over a channels-last activation X[B][L][Cin] as an implicit GEMM,
 * P[b*L+l][co] = sum_{t,ci} X[b][l+t-pad][ci] * Wp[co][t*Cin+ci]; pad 3 = forward (src/model.py:92-105, F.pad
 * left 3 / right 4 + nn.Conv1d), pad 4 with tap-flipped transposed weights = its input gradient (:273). */
int wfd_conv8_gemm_f32(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, float* P, int ldp,
                       long slab_stride, int splits, void* stream);

/* 1 when the conv entry points below run an (M = Bn*L, Cin, Cout) conv with the LDS-resident kernel (Cin = 32 on a long
 * sequence: input rows and the weight tile stay in LDS, no K split): take the fused single-launch form then */
int wfd_conv8_direct(int M, int Cin, int Cout);
/* ... for a given operand mode (modes 1 / 2: Cin, Cout in {32, 64} with at least 256 chunks of 128 rows) */
int wfd_conv8_direct_mode(int M, int Cin, int Cout, int mode);

/* The same contractions with the consumer's element-wise math fused into the store (single split: no slab round
 * trip, one launch less): epi 1 = wfd_ep_affine_act (scale/shift/act), epi 2 = wfd_ep_mul (D/dmode/scale, no second
 * copy).  Used where the product alone fills the chip (DF convs over B*L rows, large-batch Linear layers). */
int wfd_gemm_f32_fused(int layout, const float* A, int lda, const float* B, int ldb, float* out, int ldo, int M, int N,
                       int K, int epi, const float* scale, const float* shift, int act, const float* D, int ldd,
                       int dmode, void* stream);
int wfd_conv8_gemm_f32_fused(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, float* out,
                             int ldo, int epi, const float* scale, const float* shift, int act, const float* D, int ldd,
                             int dmode, void* stream);
/* wfd_conv8_gemm_f32_fused (epi 1: conv + folded eval BatchNorm + activation, src/model.py:91-104,140-176) whose launch also
 * writes the following MyMaxPool1dPadSame(8, 8) (src/model.py:107-131) of its output to PO[Bn][L/8][Cout] -- the values of
 * wfd_df_pool8_fwd.  Only for Cin = 32, pad = 3, L % 8 == 0, f32 operand mode and a product large enough for the
 * LDS-resident conv form (wfd_conv8_direct); WFD_ERR_ARG otherwise (issue the two launches). */
int wfd_conv8_gemm_f32_fused_pool(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, float* out,
                                  int ldo, const float* scale, const float* shift, int act, float* PO, void* stream);

/* Small-batch form of the same contractions + consumers: the K split runs INSIDE the workgroup (one 32x32 or 32x16 output
 * tile per workgroup, its 2-8 waves each contract a k-range with exact-f32 MFMAs and fold through LDS in wave order), so
 * the consumer's math is applied in the store -- no slabs, no second launch.  layout 0 NT / 1 NN as wfd_gemm_f32.
 *   epi 0: out = product (read by row / column reducing consumers as ONE slab);
 *   epi 1: wfd_ep_affine_act (nn.Linear bias + ReLU, eval BatchNorm + ELU / ReLU, src/model.py:20-34,140-196);
 *   epi 2: wfd_ep_mul, out2 = out * rowscale[m] when out2 != NULL (autograd.grad chain, src/mlp_df_wgan_train.py:204-213,229,273);
 *   epi 3: wfd_ep_lrelu_drop with bias = shift, H = out, D = out2 (nn.LeakyReLU + nn.Dropout, src/model.py:58-65).
 * Operands 16-byte aligned, lda / ldb multiples of 4.  cfg 0 = the library's choice (wfd_gemm_wk_pick), else 1..5.
 * Results agree with the two-launch form to fp32 summation order (tests/test_gpu_gemm_wk.py). */
int wfd_gemm_f32_wk(int layout, const float* A, int lda, const float* B, int ldb, int M, int N, int K, int epi, float* out,
                    int ldo, float* out2, int ldo2, const float* scale, const float* shift, int act, const float* D, int ldd,
                    int dmode, const float* rowscale, const float* mask, int ldmask, float slope, int cfg, void* stream);

/* Up to 3 INDEPENDENT such products of one layout and configuration (3 or 4) in one launch: in the G-step
 * (src/mlp_df_wgan_train.py:246-263) the critic's layer k on the generated rows and the DF observer's FC layer k read
 * nothing of each other, and neither do the two input-gradient chains (:273).  desc (HOST memory): per product 24 64-bit
 * words = the arguments of wfd_gemm_f32_wk in order (slope as the float's bit pattern).  Same bits as separate launches.
 * Returns 1 when the products do not share layout / configuration: launch them one by one then. */
int wfd_gemm_f32_wk_multi(const long long* desc, int n, void* stream);
/* the k=8 "SAME" conv of wfd_conv8_gemm_f32 (src/model.py:92-105) in that form; Cin a multiple of 32; epi 0..2 */
int wfd_conv8_f32_wk(const float* X, int Bn, int L, int Cin, int pad, const float* Wp, int Cout, int epi, float* out, int ldo,
                     const float* scale, const float* shift, int act, const float* D, int ldd, int dmode, int cfg,
                     void* stream);
/* configuration the library picks for a product (layout 0 NT, 1 NN, 3 conv), 0 = use the slab form of wfd_gemm_f32 */
int wfd_gemm_wk_pick(int layout, int M, int N, int K);

/* bf16-operand MFMA (fp32 accumulate) NT product for the large-batch forward path (bulk sampling, BASELINE config
 * 4/5): A (M,K) and B (N,K) bf16 row-major, K and both leading dimensions multiples of 8, 16-byte aligned.
 * O != NULL: O = bf16(act(acc*scale[n]+shift[n])) (act 0 none, 1 ReLU, 3 sigmoid); else raw fp32 into P.
 * nn.Linear (+ folded eval BatchNorm + ReLU) of src/model.py:28-32 at bf16 operand precision -- NOT the parity
 * path (fp32 kernels above); tolerance stated in tests/test_gpu_bf16.py. */
int wfd_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* P, int ldp, void* O,
                     int ldo, const float* scale, const float* shift, int act, void* stream);

/* fp32 -> bf16 (round-to-nearest-even) with zero padding of columns [N, Npad) */
int wfd_f32_to_bf16(const float* in, int ldi, void* out, int ldo, int M, int N, int Npad, void* stream);

/* ---- MLP epilogues ------------------------------------------------------------------------------------ */

/* out = act(sumP * scale[n] + shift[n]); act 0 none, 1 ReLU, 2 ELU, 3 sigmoid; scale/shift may be NULL.
 * bias + nn.ReLU (src/model.py:21,24), eval-mode BatchNorm folded to scale/shift + ELU/ReLU (:143-183,188-193). */
int wfd_ep_affine_act(const float* P, int ns, long ss, int ldp, int M, int N, const float* scale, const float* shift,
                      int act, float* out, int ldo, void* stream);

/* Discriminator hidden layer: z = sumP + bias; H = LeakyReLU(z, slope) * mask; D = mask * (z > 0 ? 1 : slope).
 * nn.LeakyReLU(0.2) + nn.Dropout(0.2), src/model.py:58-65; mask holds {0, 1/(1-p)} (explicit for parity, from
 * wfd_rng_dropout_mask otherwise).  D is the layer's local derivative, kept for the input-gradient chain. */
int wfd_ep_lrelu_drop(const float* P, int ns, long ss, int ldp, int M, int N, const float* bias, const float* mask,
                      int ldmask, float slope, float* H, int ldh, float* D, int ldd, void* stream);

/* Discriminator head nn.Linear(512,1) (src/model.py:66) as a wave-reduced GEMV, fused with the first step of
 * d validity / d input:  v = <H3,w4> + b4;  delta = w4 o D3;  delta_s = rowscale[m] * delta (may be NULL). */
int wfd_d_head(const float* H3, int ldh, const float* D3, int ldd, const float* w4, const float* b4, int M, int K,
               const float* rowscale, float* v, float* delta, float* delta_s, int ldo, void* stream);

/* wfd_ep_lrelu_drop on the slabs of the last hidden layer (src/model.py:63-65) and wfd_d_head (:66) in one launch: same
 * outputs (H3, D3, v, delta, delta_s), same operations in the same order, one dependent launch less per critic pass. */
int wfd_d_head_lrelu(const float* P, int ns, long ss, int ldp, const float* bias, const float* mask, int ldmask,
                     float slope, float* H3, int ldh, float* D3, int ldd, const float* w4, const float* b4, int M, int K,
                     const float* rowscale, float* v, float* delta, float* delta_s, int ldo, void* stream);

/* out = sumP * f(D) * colscale[n]; dmode 0 none, 1 D, 2 [D > 0] (ReLU'), 3 ELU' from its output; optional
 * out2 = out * rowscale[m].  One step of autograd.grad(validity, traces, create_graph=True) in closed form
 * (src/mlp_df_wgan_train.py:204-206) and of backward() through ReLU/ELU + folded BN (:273). */
int wfd_ep_mul(const float* P, int ns, long ss, int ldp, int M, int N, const float* D, int ldd, int dmode,
               const float* colscale, const float* rowscale, float* out, float* out2, int ldo, void* stream);

/* Row-wise gradient penalty on g = sumP[m][0..S): nsq = ||g||^2;
 * mode 0 W-div (k=2,p=6): pen = nsq^3 (src/mlp_df_wgan_train.py:207,213); mode 1 WGAN-GP: pen = (||g||-1)^2
 * (src/utils.py:82-83).  GH = d(mean penalty)/dg, the seed of the double-backward (d_loss.backward(), :229).
 * rowflag (may be NULL): rows with flag 0 carry no penalty (pen = 0, GH = 0), e.g. the real / fake rows under
 * WGAN-GP, where only the interpolates are penalised. */
int wfd_ep_penalty_row(const float* P, int ns, long ss, int ldp, int M, int S, int mode, float invB, float lambda,
                       const float* rowflag, float* nsq, float* pen, float* GH, int ldg, void* stream);

/* out[n] (+)= sum_m X[m][n] * rowscale[m]  -- bias gradients (backward(), :229,:273) */
int wfd_colsum(const float* X, int ldx, int M, int N, const float* rowscale, float* out, int accumulate,
               void* stream);

/* All column sums of a step in one launch.  desc (DEVICE memory) holds 11 64-bit words per segment:
 * [X, ldx, rowscale, X2, ldx2, rowscale2, M, N, out, first_block, acc]; out[n] = sum_m X[m][n]*rs[m] (+ the same
 * over X2), acc[n] += out[n]; pointers may be 0 (rowscale = 1, no second source, no accumulator).  Blocks cover 32
 * columns each.  With N = 1 segments this also produces the step's loss scalars (:218,226-227) on device. */
int wfd_colsum_multi(const long long* desc, int nseg, int total_blocks, void* stream);

/* wfd_colsum_multi whose outputs are parameter gradients consumed on the spot: an output that lies inside the flat
 * gradient buffer [gbase, gbase + numel) is still written, and its parameter -- same offset in pbase / sqbase -- takes
 * its RMSprop step (torch defaults, as wfd_rmsprop) in the same launch: d_loss.backward() + optimizer_D.step() for the
 * biases and the critic's head (src/mlp_df_wgan_train.py:229-230), whose gradients ARE column sums.  Outputs elsewhere
 * (loss scalars) are plain.  Bookkeeping as wfd_rmsprop: state[0] += inc0, state[1] += inc1 (state may be NULL). */
int wfd_colsum_multi_rmsprop(const long long* desc, int nseg, int total_blocks, const float* gbase, float* pbase,
                             float* sqbase, long numel, float lr, float alpha, float eps, unsigned long long* state,
                             unsigned long long inc0, unsigned long long inc1, void* stream);

/* torch.optim.RMSprop.step with torch defaults (alpha .99, eps 1e-8), src/mlp_df_wgan_train.py:163-164,230,275,
 * over one flat parameter buffer; g is multiplied by gscale first (1/world_size after an all-reduce).  As the
 * last kernel of a step it can also do the step's bookkeeping: state[0] += inc0, state[1] += inc1 (state may be
 * NULL), see wfd_iter_advance. */
int wfd_rmsprop(float* p, const float* g, float* sq, long n, float lr, float alpha, float eps, float gscale,
                unsigned long long* state, unsigned long long inc0, unsigned long long inc1, void* stream);

/* The same update on up to 8 element ranges [ranges[2k], ranges[2k+1]) (HOST array) of the flat buffers: the vectors
 * (biases, BatchNorm affine, the critic's head) of a network whose weight matrices were updated inside their
 * gradient's store by wfd_gemm_f32_rmsprop.  Bookkeeping as wfd_rmsprop. */
int wfd_rmsprop_ranges(float* p, const float* g, float* sq, const long* ranges, int nranges, float lr, float alpha, float eps,
                       unsigned long long* state, unsigned long long inc0, unsigned long long inc1, void* stream);
/* the small work that ends a G-step whose matrices were updated in their gradients' stores, as ONE launch:
 * wfd_colsum_multi_rmsprop (desc, nseg, cs_blocks: the first layer's bias gradient and its step; also does the step's
 * bookkeeping) + wfd_rmsprop_ranges on `ranges` (the remaining vectors; optimizer_G.step(), src/mlp_df_wgan_train.py:275)
 * + wfd_bn_running_update_layers (meanA ... nupd_last; meanA = NULL leaves that part out).  Same values as the three. */
int wfd_g_end(const long long* desc, int nseg, int cs_blocks, const float* gbase, float* pbase, float* sqbase, long numel,
              const long* ranges, int nranges, float lr, float alpha, float eps, unsigned long long* state,
              unsigned long long inc0, unsigned long long inc1, const float* meanA, const float* m2A, int NA, float* rmeanA,
              float* rvarA, const float* meanB, const float* m2B, int NB, float* rmeanB, float* rvarB, float count,
              float momentum, int groups, int nupd_last, void* stream);

/* loss.backward() + optimizer.step() for ONE weight matrix (src/mlp_df_wgan_train.py:229-230, 273-275): the gradient
 * g = (M,N,K) product of the given layout (2 = TN: dW = dZ^T H over the batch rows) is consumed in the store,
 * SQ = alpha SQ + (1-alpha) g^2;  P -= lr g / (sqrt(SQ) + eps)  (torch.optim.RMSprop defaults), on the parameter
 * matrix P[M][ldp] and its square average SQ of the same layout; the gradient never reaches memory.  Same bits as
 * wfd_gemm_f32 into a gradient buffer followed by wfd_rmsprop.  Unsplit contraction (K = rows of the batch);
 * single-GPU path (data parallel all-reduces the gradient first). */
int wfd_gemm_f32_rmsprop(int layout, const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* P,
                         int ldp, float* SQ, float lr, float alpha, float eps, void* stream);

/* Up to 4 such weight matrices in ONE launch (the critic's three layers at the end of optimizer_D's step, the generator's
 * four at the end of optimizer_G's, src/mlp_df_wgan_train.py:229-230, 273-275): desc (HOST memory) holds, per matrix, 10
 * 64-bit words [A, lda, B, ldb, M, N, K, P, ldp, SQ] of a TN product as in wfd_gemm_f32_rmsprop(layout = 2).  The
 * products must be independent of each other (none reads a P of the list).  Returns 1 when a problem does not fit the
 * fast form (operands not 16-byte aligned, N <= 32, operand mode != f32): launch them one by one then.  Same bits. */
int wfd_gemm_f32_rmsprop_multi(const long long* desc, int n, float lr, float alpha, float eps, void* stream);
/* wfd_gemm_f32_rmsprop_multi whose launch also carries wfd_colsum_multi_rmsprop(cs_desc, cs_nseg, cs_blocks, gbase, pbase,
 * sqbase, numel, lr, alpha, eps, state, inc0, inc1) in its first cs_blocks workgroups: the whole end of a critic step
 * (d_loss.backward()'s weight, bias and head gradients with optimizer_D.step(), src/mlp_df_wgan_train.py:229-230, the loss
 * scalars of :215-227 and the step's bookkeeping) as ONE launch.  The column sums must not read the parameter matrices. */
int wfd_gemm_f32_rmsprop_multi_colsum(const long long* desc, int n, float lr, float alpha, float eps, const long long* cs_desc,
                                      int cs_nseg, int cs_blocks, const float* gbase, float* pbase, float* sqbase, long numel,
                                      unsigned long long* state, unsigned long long inc0, unsigned long long inc1,
                                      void* stream);
/* ... and also the other two parts of wfd_g_end (RMSprop on `ranges`; the BatchNorm running statistics, meanA = NULL leaves
 * them out): the whole end of a generator step (loss.backward()'s weight / bias gradients with optimizer_G.step(),
 * src/mlp_df_wgan_train.py:273-275, and the second forward's running-statistics update, :240) as ONE launch */
int wfd_gemm_f32_rmsprop_multi_end(const long long* desc, int n, float lr, float alpha, float eps, const long long* cs_desc,
                                   int cs_nseg, int cs_blocks, const float* gbase, float* pbase, float* sqbase, long numel,
                                   const long* ranges, int nranges, unsigned long long* state, unsigned long long inc0,
                                   unsigned long long inc1, const float* meanA, const float* m2A, int NA, float* rmeanA,
                                   float* rvarA, const float* meanB, const float* m2B, int NB, float* rmeanB, float* rvarB,
                                   float count, float momentum, int groups, int nupd_last, void* stream);
/* The data-parallel form of wfd_gemm_f32_rmsprop_multi_colsum: the same weight gradients as ONE launch, but STORED (desc
 * words 7 / 8 = the gradient matrix and its row stride, word 9 unused) -- the ranks' gradients have to meet before the
 * optimizer may see them (loss.backward() of src/mlp_df_wgan_train.py:229,274 without :230,275) -- optionally carrying
 * wfd_colsum_multi(cs_desc, cs_nseg, cs_blocks) in its first workgroups (cs_desc = NULL: none).  Returns 1 when the fast
 * form does not apply (fewer than 8 tiles, unaligned operands, N <= 32): launch the products one by one then. */
int wfd_gemm_f32_multi_grad(const long long* desc, int n, const long long* cs_desc, int cs_nseg, int cs_blocks, void* stream);

/* nn.BatchNorm1d(n, eps=0.8) training forward (src/model.py:23), in two halves so that data-parallel ranks can
 * combine (mean, M2) in between: stats writes Z = sumP + bias and the per-column mean / sum of squared deviations;
 * apply writes xhat, relu(gamma*xhat+beta), invstd and updates running_mean / running_var (unbiased) nupd times.
 * wfd_bn_train_relu is both halves in one launch (single rank). */
int wfd_bn_stats(const float* P, int ns, long ss, int ldp, int M, int N, const float* bias, float* Z, int ldz,
                 float* mean, float* m2, void* stream);
int wfd_bn_train_relu(const float* P, int ns, long ss, int ldp, int M, int N, const float* bias, float eps,
                      const float* gamma, const float* beta, float* XH, float* H, int ldo, float* mean, float* m2,
                      float* invstd, float* rmean, float* rvar, float momentum, int nupd, void* stream);
int wfd_bn_apply_relu(const float* Z, int ldz, int M, int N, const float* mean, const float* m2, float count, float eps,
                      const float* gamma, const float* beta, float* XH, float* H, int ldo, float* invstd, float* rmean,
                      float* rvar, float momentum, int nupd, void* stream);
int wfd_bn_running_update(const float* mean, const float* m2, float count, int N, float* rmean, float* rvar,
                          float momentum, int nupd, void* stream);

/* The running-statistics updates of BOTH BatchNorm layers of the generator (src/model.py:30-31) for `groups` stacked
 * training-mode forwards in one launch: per layer, batches g = 0..groups-1 (mean / m2 [groups][N]) are applied in
 * order, the last one nupd_last times (the G-step's second generator(z, c) on the same batch,
 * src/mlp_df_wgan_train.py:196,240).  Same operations, element for element, as wfd_bn_running_update batch by batch. */
int wfd_bn_running_update_layers(const float* meanA, const float* m2A, int NA, float* rmeanA, float* rvarA,
                                 const float* meanB, const float* m2B, int NB, float* rmeanB, float* rvarB, float count,
                                 float momentum, int groups, int nupd_last, void* stream);

/* Backward of ReLU + BatchNorm1d (training) (:273). phase bit 0: local column sums (sums[0..N) = sum dy,
 * sums[N..2N) = sum dy*xhat); bit 1: dz, dgamma, dbeta and the preceding Linear's bias gradient. */
int wfd_bn_bwd(const float* P, int ns, long ss, int ldp, int M, int N, const float* H, const float* XH, int ldh,
               const float* gamma, const float* invstd, float count, float* sums, float* DZ, int lddz, float* dgamma,
               float* dbeta, float* dbias, int phase, void* stream);

/* wfd_bn_bwd with phase = 3 for a tall batch (large-batch steps): row chunks of 128 in parallel, partial sums combined
 * in chunk order (deterministic); part = scratch of 2 * ceil(M / 128) * N floats */
int wfd_bn_bwd_tall(const float* P, int ns, long ss, int ldp, int M, int N, const float* H, const float* XH, int ldh,
                    const float* gamma, const float* invstd, float count, float* sums, float* DZ, int lddz, float* dgamma,
                    float* dbeta, float* dbias, float* part, void* stream);

/* Generator output: nn.Sigmoid + learned-length tail mask, src/model.py:33,42-47 (replaces zeros_like / index
 * scatter / cumsum / mul).  idx[m] = trunc(y0*range + smin) + 1; *err |= 1 where the reference would raise
 * IndexError (idx >= S). */
int wfd_ep_sigmoid_lenmask(const float* P, int ns, long ss, int ldp, int M, int S, const float* bias, float range,
                           float smin, float* out, int ldo, int* idx, int* err, void* stream);

/* Backward through the mask and the sigmoid into the generator's last pre-activation (:273):
 * dz = [j < idx] * (dscale * sumP + dDF[j-1]) * y(1-y); dbias = column sums.  dbias may be NULL: dz only, one thread per
 * element (tall batches; take the column sums of DZ with wfd_colsum_tall). */
int wfd_g_out_bwd(const float* P, int ns, long ss, int ldp, int M, int S, float dscale, const float* dDF, int lddf,
                  const float* fake, int ldf, const int* idx, float* DZ, int lddz, float* dbias, void* stream);

/* d_loss = -mean(rv) + mean(fv) + pen_weight * sum(pen)/B (src/mlp_df_wgan_train.py:215,218,224,226-227).
 * out[5] = {d_loss, penalty, w_dist, mean_rv, mean_fv}; acc[3] += {d_loss, w_dist, 1} (device epoch sums). */
int wfd_d_loss(const float* v, const float* pen, int B, int npen, float invB, float pen_weight, float* out, float* acc,
               void* stream);

/* out[2] = {g_loss = -mean(fv), nsel = #{fv > 0}} (:247,251) */
int wfd_g_head(const float* fv, int B, float invB, float* out, void* stream);

/* ---- DF observer (src/model.py:135-212), eval mode ---------------------------------------------------- */

/* first conv (Cin = 1) + folded BN + activation; reads x = X[b*ldx + xoff + l] (fake[:,1:], :257-259) */
int wfd_df_conv1_fwd(const float* X, int ldx, int xoff, int Bn, int L, const float* W, const float* scale,
                     const float* shift, int act, float* Y, int Cout, void* stream);

/* MyMaxPool1dPadSame(8,1): zero pad (3,4) then MaxPool1d(8) stride 8, src/model.py:119-132; channels-last */
int wfd_df_pool8_fwd(const float* Y, int Bn, int L, int C, float* PO, void* stream);

/* backward of pool + activation (1 ReLU, 2 ELU) + folded-BN scale in one pass (:273) */
int wfd_df_pool8_bwd_act(const float* dPO, const float* Y, int Bn, int L, int C, const float* scale, int act,
                         float* dC, void* stream);

/* nn.CrossEntropyLoss over the rows with fv > 0 (:251-263) in static-shape form; dlogits carries alpha/nsel */
int wfd_masked_ce(const float* logits, int ldl, int M, int C, const int* labels, const float* fv, const float* nsel,
                  float alpha, float* ce, float* dlogits, int lddl, void* stream);

/* out[4] = {g_loss, f_loss, g_loss + alpha f_loss, nsel}; acc[4] += {g_loss, f_loss, combined, 1} (:263-271) */
int wfd_g_loss(const float* ce, int B, const float* head, float alpha, float* out, float* acc, void* stream);

/* wfd_g_head + wfd_masked_ce + wfd_g_loss in ONE launch (one workgroup; B <= 256 rows, C <= 128 classes: the
 * reference's configurations), src/mlp_df_wgan_train.py:247-271: head[2] = {-mean(fv), #{fv > 0}}, ce / dlogits as wfd_masked_ce,
 * out[4] / acc[4] as wfd_g_loss.  Single rank only (the exact data-parallel mode all-reduces head in between). */
int wfd_g_tail(const float* logits, int ldl, int B, int C, const int* labels, const float* fv, float invB, float alpha,
               float* ce, float* dlogits, int lddl, float* head, float* out, float* acc, void* stream);

/* input gradient of the first conv: dX[b][l] = sum_{co,t} W[co][t] dC[b][l-t+3][co] (:273) */
int wfd_df_conv1_bwd(const float* dC, int Bn, int L, int Cout, const float* W, float* dX, int lddx, void* stream);

/* ---- data side ---------------------------------------------------------------------------------------- */

/* nn.Dropout keep-mask values {0, 1/(1-p)} from Philox4x32-10 (fast mode; parity tests pass explicit masks).
 * The Philox counter is state[0] + base, with state in DEVICE memory (may be NULL = 0) so that a captured
 * graph draws fresh numbers on every replay; wfd_iter_advance moves it. */
int wfd_rng_dropout_mask(float* out, long n, float p, unsigned long long seed, const unsigned long long* state,
                         unsigned long long base, void* stream);

/* wfd_rng_dropout_mask for up to 8 buffers in one launch (the masks of all iterations of a group of n_critic iterations,
 * drawn ahead): `slots` is a HOST array of nslots x {device pointer, element count, seed, base}, read during the call;
 * buffer s holds what wfd_rng_dropout_mask(out_s, n_s, p, seed_s, state, base_s) would write. */
int wfd_rng_dropout_mask_slots(const unsigned long long* slots, int nslots, float p, const unsigned long long* state,
                               void* stream);

/* z ~ N(0,1) (np.random.normal, src/mlp_df_wgan_train.py:194; np.random.randn, src/generate.py:53) on device */
int wfd_rng_normal_rows(float* out, int ld, int M, int Z, unsigned long long seed, const unsigned long long* state,
                        unsigned long long base, void* stream);

/* z and the step's dropout keep-masks in one launch (wfd_rng_normal_rows + wfd_rng_dropout_mask, same streams) */
int wfd_rng_step_inputs(float* zout, int ld, int M, int Z, unsigned long long zseed, float* mout, long nmask, float p,
                        unsigned long long mseed, const unsigned long long* state, void* stream);

/* U[0,1) interpolation weights (np.random.random((B,1)), src/utils.py:67) */
int wfd_rng_uniform(float* out, long n, unsigned long long seed, const unsigned long long* state,
                    unsigned long long base, void* stream);

/* WGAN-GP interpolates alpha*real + (1-alpha)*fake on the first S columns, class columns [S,W) copied
 * (src/utils.py:69) */
int wfd_interpolate_rows(const float* real, const float* fake, const float* alpha, float* out, int B, int S, int W,
                         int ld, void* stream);

/* state[0] += inc0 (Philox counter), state[1] += inc1 (cursor into the epoch permutation) */
int wfd_iter_advance(unsigned long long* state, unsigned long long inc0, unsigned long long inc1, void* stream);

/* one training batch: rows idx[] of the resident dataset + one-hot labels into the real / fake / generator-input
 * buffers (DataLoader + np.eye(class_dim)[c], src/mlp_df_wgan_train.py:182-186); row b reads
 * idx[(state[1] + row_offset + b) mod n_idx], clamped to [0, n_rows); row_offset = rank * B picks this rank's rows
 * of a global batch in data-parallel runs */
int wfd_gather_batch(const float* X, int ldX, const int* y, const long long* idx, const unsigned long long* state,
                     long n_idx, long n_rows, long row_offset, int B, int S, int C, float* real, float* fake, int ldd,
                     float* zin, int ldz, int Z, int* labels_out, void* stream);

int wfd_onehot_rows(const int* labels, int M, int C, float* out, int ld, int col0, void* stream);

/* preprocessing.MinMaxScaler().fit on the float64 feature matrix (src/mlp_df_wgan_train.py:132-133): per-column
 * data_min_ / data_max_ (exact, any order); part: 2 * max_chunks * S doubles of scratch */
int wfd_colminmax_f64(const double* X, long N, int S, double* part, int max_chunks, double* mn, double* mx, void* stream);
/* MinMaxScaler.transform (X *= scale_; X += min_ in float64, two roundings) + the .float() of :184 */
int wfd_minmax_transform_f64(const double* X, long N, int S, const double* scale, const double* min_, float* out,
                             void* stream);

/* checkpoint dump, src/mlp_df_wgan_train.py:291-292: total_real = scaler.inverse_transform(scaler.transform(x)) in float64
 * (four in-place passes, one rounding each) and total_fake = scaler.inverse_transform(float32 generated rows) (each pass in
 * double, rounded to float32); bit-identical to sklearn (tests/test_gpu_tail.py) */
int wfd_minmax_roundtrip_f64(const double* X, long N, int S, const double* scale, const double* min_, double* out,
                             void* stream);
int wfd_minmax_inverse_f32(const float* Y, int ldy, long M, int S, const double* scale, const double* min_, float* out,
                           void* stream);

/* --clip: my_clip over every row, src/mlp_df_wgan_train.py:82-91,122 (float64, in place) */
int wfd_clip_bursts(double* X, long N, int S, double max_len, double max_out, double max_in, void* stream);

/* Burst extraction, the on-disk -> npz step in front of the trainer: extract.get_burst + extract.extract, src/extract.py:58-123,
 * for `ntraces` packet traces stored ragged (trace i = entries off[i] .. off[i+1] of `times` / `sizes`, float64, every
 * trace non-empty).  feat[i][0..S) = [number of bursts, |burst size sums| (divided by cell_norm when > 0: --norm), 0 ...];
 * bsum / btime (as long as times) receive trace i's signed burst sums / burst start times at off[i] + r; meta[i] =
 * {first kept packet, end of the kept range, number of bursts} (the host evaluates the reference's asserts :106-107
 * from them). */
int wfd_extract_bursts(const double* times, const double* sizes, const long long* off, int ntraces, int S, double cell_norm,
                       double* feat, double* bsum, double* btime, int* meta, void* stream);

/* sampler epilogue: MinMaxScaler.inverse_transform (float64 min_/scale_, fp32 data) + np.trim_zeros + np.round
 * (half-even) + astype(int) + 0 -> 1, src/generate.py:59-63.  out rows are zero-padded; len_out = lengths. */
int wfd_sample_post(const float* Y, int ldy, int M, int S, const double* min_, const double* scale_, int* out, int ldo,
                    int* len_out, void* stream);

/* the same epilogue for the generator's other consumers.  mode 0: generate.py:59-63 (= wfd_sample_post);
 * 1: WTGAN.get_ref_trace, src/gan_wt.py:79-87 (count from element 0, made even, rint, 0 -> 1; element 0 not emitted);
 * 2: query_from_gan, src/gan_glue.py:114-124 (truncating cast, trailing zeros trimmed, even length, <= 0 -> 1,
 *    alternating signs); 3: sample_trace, src/synthesize.py:86-90 (rint first, then trim zeros at both ends). */
int wfd_sample_post_mode(const float* Y, int ldy, int M, int S, const double* min_, const double* scale_, int mode, int* out,
                         int ldo, int* len_out, void* stream);

/* ---- stacked generator forwards (the heads of the iterations that share generator parameters run as ONE pass over
 *      slots * B rows; no counterpart in the reference, which runs them one by one: src/mlp_df_wgan_train.py:196) ---- */

/* wfd_gather_batch for `slots` consecutive batches: slot s reads the permutation slot_idx_stride entries after slot
 * s - 1, writes real / fake rows at real + s * slot_stride (floats), generator input and labels stacked contiguously */
int wfd_gather_batch_slots(const float* X, int ldX, const int* y, const long long* idx, const unsigned long long* state,
                           long n_idx, long n_rows, long row_offset, int B, int slots, long slot_idx_stride, int S, int C,
                           float* real, float* fake, int ldd, long slot_stride, float* zin, int ldz, int Z,
                           int* labels_out, void* stream);
/* wfd_bn_train_relu on `groups` independent batches of M rows stacked along the rows (statistics [groups][N]); the
 * running statistics take one update per group, in order, exactly as successive forward calls would leave them */
int wfd_bn_train_relu_grouped(const float* P, int ns, long ss, int ldp, int M, int groups, int N, const float* bias,
                              float eps, const float* gamma, const float* beta, float* XH, float* H, int ldo, float* mean,
                              float* m2, float* invstd, float* rmean, float* rvar, float momentum, void* stream);
/* wfd_ep_sigmoid_lenmask whose output rows go to `out + slot * slot_stride` (slot = row / rows_per_slot) */
int wfd_ep_sigmoid_lenmask_slots(const float* P, int ns, long ss, int ldp, int M, int S, const float* bias, float range,
                                 float smin, float* out, int ldo, int rows_per_slot, long slot_stride, int* idx, int* err,
                                 void* stream);

/* wfd_ep_sigmoid_lenmask_slots that also advances the device iteration state (wfd_iter_advance folded into the launch:
 * state[0] += inc0, state[1] += inc1; nothing in the launch reads the state) */
int wfd_ep_sigmoid_lenmask_slots_adv(const float* P, int ns, long ss, int ldp, int M, int S, const float* bias, float range,
                                     float smin, float* out, int ldo, int rows_per_slot, long slot_stride, int* idx, int* err,
                                     unsigned long long* state, unsigned long long inc0, unsigned long long inc1,
                                     void* stream);
/* everything a group of n_critic iterations draws before its first product as ONE launch: wfd_rng_dropout_mask_slots
 * (mslots, nslots, p, mask_state) + wfd_gather_batch_slots (X ... labels_out; the DataLoader + one-hot of
 * src/mlp_df_wgan_train.py:184-186 for `slots` batches) + wfd_rng_normal_rows on the B * slots rows of zin, columns
 * [0, Z), Philox key zseed at counter state[0] (the z of src/mlp_df_wgan_train.py:194).  Same values as the three. */
int wfd_group_inputs(const unsigned long long* mslots, int nslots, float p, const unsigned long long* mask_state,
                     const float* X, int ldX, const int* y, const long long* idx, const unsigned long long* state,
                     long n_idx, long n_rows, long row_offset, int B, int slots, long slot_idx_stride, int S, int C,
                     float* real, float* fake, int ldd, long slot_stride, float* zin, int ldz, int Z, int* labels_out,
                     unsigned long long zseed, void* stream);

/* ---- DF observer training (src/train_df.py:111-131 around src/model.py:135-206 in train mode) ---------------------- */

/* weight gradient of the k=8 "SAME" conv (autograd of MyConv1dPadSame, src/model.py:91-104):
 * P[co][t*Cin+ci] = sum_{b,l} dY[b*L+l][co] * X[b][l+t-3][ci]; `splits` slabs of Cout x ldp floats */
int wfd_conv8_wgrad_f32(const float* dY, const float* X, int Bn, int L, int Cin, int Cout, float* P, int ldp,
                        long slab_stride, int splits, void* stream);

/* nn.BatchNorm1d on a channels-last conv activation Z[M = B*L][N] in training mode (src/model.py:142,146,...):
 * statistics over all M rows (two-stage, deterministic), running stats with the unbiased variance.
 * part: scratch of ceil(M / rows_per_chunk) * 2 * N floats */
int wfd_bn_tall_stats(const float* Z, int ldz, int M, int N, int rows_per_chunk, float* part, float eps, float momentum,
                      float* mean, float* invstd, float* rmean, float* rvar, void* stream);
/* Y = act(gamma * (Z - mean) * invstd + beta); act 0 none, 1 ReLU, 2 ELU */
int wfd_bn_tall_apply(const float* Z, int M, int N, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, int act, float* Y, void* stream);
/* backward of that BatchNorm given dA = d loss / d (BN output): sums = [sum dA ; sum dA*xhat], dgamma, dbeta,
 * dZ = gamma * invstd * (dA - s0/M - xhat * s1/M)   (dZ may alias dA) */
int wfd_bn_tall_bwd(const float* dA, const float* Z, int M, int N, int rows_per_chunk, float* part, const float* mean,
                    const float* invstd, const float* gamma, float* sums, float* dgamma, float* dbeta, float* dZ,
                    void* stream);
/* out[n] = sum_m X[m][n] over a tall matrix (conv bias gradients); part: ceil(M / rows_per_chunk) * N floats */
int wfd_colsum_tall(const float* X, int ld, int M, int N, int rows_per_chunk, float* part, float* out, void* stream);
/* wfd_colsum_tall for callers that cannot allocate or pass a workspace (code under HIP-graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  The chunk partials live in a static device
 * array inside the library (wfd_static_scratch_floats() floats = 4 MiB, resident on every GPU the library runs on until the
 * process ends); the call allocates nothing.  Returns 3 when ceil(M / rows_per_chunk) * N exceeds that size (use
 * wfd_colsum_tall with a workspace then).  Same kernels, same summation order, bit-identical results.  All calls on one
 * GPU share the array: calls that overlap on one GPU (different streams) race -- serialise them.  The first call on a
 * device looks the array's address up (hipGetSymbolAddress): make it outside stream capture.  Every GPU has its own copy
 * of the array: a non-null `stream` must belong to the CURRENT device (hipSetDevice first); a stream of another device is
 * rejected with code 1 instead of handing its kernels the wrong GPU's array. */
int wfd_colsum_tall_static(const float* X, int ld, int M, int N, int rows_per_chunk, float* out, void* stream);
int wfd_static_scratch_floats(void);
/* weight gradient of the first conv (Cin = 1): dW[co][t] = sum_{b,l} dZ[b][l][co] * x[b][l+t-3], x[b][l] = X[b*ldx+xoff+l];
 * part: ceil(L / 128) * Bn * Cout * 8 floats */
int wfd_conv1_wgrad(const float* dZ, const float* X, int ldx, int xoff, int Bn, int L, int Cout, float* part, float* dW,
                    void* stream);
/* WT[ci][t'*Cout+co] = W[co][(7-t')*Cin+ci]: the operand of the input-gradient conv, rebuilt after every update */
int wfd_conv_wflip(const float* W, int Cout, int Cin, float* WT, void* stream);
/* nn.Dropout as a mask multiply (mask values {0, 1/(1-p)} from wfd_rng_dropout_mask); out may alias X */
int wfd_mul_mask(const float* X, const float* mask, long n, float* out, void* stream);
/* torch.optim.Adamax step with torch's defaults passed in (src/train_df.py:115); step[0] = steps taken so far, kept on
 * the device and incremented when advance != 0 */
int wfd_adamax(float* p, const float* g, float* m, float* u, long n, float lr, float beta1, float beta2, float eps,
               long long* step, int advance, void* stream);
/* correct[0] += #{rows whose first-maximum logit index equals the label} (src/train_df.py:141-143,153-155) */
int wfd_argmax_correct(const float* logits, int ldl, int M, int C, const int* labels, int* correct, void* stream);

/* ---- peer-to-peer ("mesh") gradient exchange over xGMI (SURVEY section 8 e) ------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE (src/mlp_df_wgan_train.py:98-100 picks one device); it replaces what a DistributedDataParallel
 * wrapper's all-reduce + src/mlp_df_wgan_train.py:230,275 (optimizer.step()) would do.  One process per GPU.  Every rank
 * allocates an exchange REGION (gradients and, for the fused form, parameters) and a flag block with wfd_mesh_alloc, hands
 * the 64-byte IPC handles to its peers over any host channel (the Python layer uses torch.distributed's object gather),
 * maps theirs with wfd_mesh_open and builds a mesh from the W region / flag pointers in rank order (its own entries are
 * its own allocations).  After that an exchange is ONE kernel launch on the caller's stream (capturable in a HIP graph):
 * start barrier -> every rank sums its 1/W shard from all ranks in rank order and stores the result to all ranks -> end
 * barrier (wfd_gan_amd/csrc/mesh.hip).  All ranks must issue the same sequence of exchanges, one at a time per mesh.  Waits
 * are bounded (20 s by default, wfd_mesh_set_timeout_ms): a peer that never arrives is counted in wfd_mesh_status' third word
 * instead of hanging the GPU -- the exchange's result is then undefined, so check the status at a point where the host
 * synchronises anyway (the Python layer does at every epoch end), and keep ranks from running far ahead of a rank that is
 * busy on the host (a host barrier after rank-0-only work).
 * kind 0: region (hipMalloc), 1: flag block (wfd_mesh_flag_bytes() bytes, uncached); both zeroed; handle: 64 bytes out. */
int wfd_mesh_flag_bytes(void);
int wfd_mesh_alloc(long bytes, int kind, void** ptr, void* ipc_handle);
int wfd_mesh_open(const void* ipc_handle, void** ptr);
int wfd_mesh_close(void* ptr);
int wfd_mesh_free(void* ptr);
/* data[p] / flags[p]: rank p's region / flag block as mapped into THIS process, p = 0 .. world-1 (world <= 8); the device
 * that is current now is the mesh's device for good */
int wfd_mesh_create(int rank, int world, void* const* data, void* const* flags, long region_floats, void** mesh);
int wfd_mesh_destroy(void* mesh);
/* workgroups (1024 threads) an exchange may use, 1 .. 256 (default 128, one per second CU).  Every workgroup of every rank waits
 * for the peers, so all of them must be resident at once: ranks that SHARE a GPU (test rigs) must split its CUs. */
int wfd_mesh_set_max_blocks(void* mesh, int max_blocks);
/* bound of every in-kernel wait for a peer, 1 ms .. 10 min (default 20 s) */
int wfd_mesh_set_timeout_ms(void* mesh, long ms);
/* region[off, off + n) <- sum over ranks, in place on every rank, identical bits everywhere (off, n multiples of 4) */
int wfd_mesh_allreduce_f32(void* mesh, long off, long n, void* stream);
/* the data-parallel optimizer step as ONE launch: g = gscale * sum over ranks of region[grad_off + i] (rank order), RMSprop
 * (wfd_rmsprop's arithmetic, same bits) on the OWNER's region[param_off + i] / sq[i], and the updated parameter stored to
 * every rank's region[param_off + i]; i in [0, n), each element owned by one rank.  sq: this rank's square averages of
 * the same n elements (local memory, only the owner's shard is touched).  state / inc0 / inc1 as wfd_rmsprop. */
int wfd_mesh_reduce_rmsprop(void* mesh, long grad_off, long param_off, float* sq, long n, float lr, float alpha, float eps,
                            float gscale, unsigned long long* state, unsigned long long inc0, unsigned long long inc1,
                            void* stream);
/* the small exchanges of the exact (global-batch, SyncBN) mode, one-shot (every rank reads all ranks' vectors and forms the
 * result itself, in rank order; in place; n <= 1024 * max_blocks):
 *   region[off, off + n) <- sum over ranks                                       (BatchNorm backward sums, selected-row count)
 *   (mean, M2) of equal row shards -> the pair over all rows, Chan et al.:  mean = sum_p mean_p / W,
 *   M2 = sum_p M2_p + rows_per_rank * sum_p (mean_p - mean)^2              (train-mode BatchNorm1d over the global batch,
 *                                                                            src/model.py:23 under a batch split) */
int wfd_mesh_allreduce_small(void* mesh, long off, int n, void* stream);
int wfd_mesh_bn_combine(void* mesh, long mean_off, long m2_off, int n, float rows_per_rank, void* stream);
/* synchronises the device; out3 = {exchanges completed, 0, waits that timed out} */
int wfd_mesh_status(void* mesh, long* out3);

#ifdef __cplusplus
}
#endif
#endif /* WFDGAN_H */
