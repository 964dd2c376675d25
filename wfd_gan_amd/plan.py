"""Launch plans: a step of the hot path is a fixed list of C-ABI calls on preallocated device buffers.

All pointers and sizes are resolved once when the plan is built; running a plan is a tight loop of ctypes calls
that only appends the stream, so the same plan can be run eagerly or captured into a HIP graph
(torch.cuda.CUDAGraph) and replayed with one launch.
"""
from __future__ import annotations

import os
from collections import namedtuple
from typing import List, Optional, Tuple

import torch

from ._lib import lib

Slabs = namedtuple("Slabs", "ptr ns ss ldp M N")

NT, NN, TN = 0, 1, 2
HOST = object()          # marker: plan entry executed on the host (e.g. a torch.distributed collective)
FORK = object()          # marker: the side stream waits for everything launched on the main stream so far
JOIN = object()          # marker: the main stream waits for the side stream
ACT_NONE, ACT_RELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3
DM_NONE, DM_MUL, DM_RELU, DM_ELU = 0, 1, 2, 3


def ptr(t) -> int:
    if t is None:
        return 0
    if isinstance(t, int):
        return t
    return t.data_ptr()


def rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# Tuning hook (tools/wk_tune.py): {(layout, M, N, K): cfg} consulted before the library's own choice of the
# split-inside-the-workgroup configuration (layout 3 = conv: M = rows, N = Cout, K = 8 Cin); cfg 0 = the slab form.  Empty in
# production: the measured choices live in wfd_gemm_wk_pick.
WK_OVERRIDE = {}


class Plan:
    """Accumulates (function, args, lane) triples; ``run()`` launches them in order.

    Lane 0 is the caller's current stream, lane 1 a side stream: independent branches of a step (weight-gradient
    GEMMs next to the backward chain, the DF observer next to the critic pass) are issued on the side lane between
    ``fork()`` / ``join()`` markers, so that small kernels which cannot fill 256 CUs on their own overlap.  Under
    HIP-graph capture the same markers become graph dependencies.  Each lane has its own slab workspace."""

    _side_streams = {}

    def __init__(self, workspace: torch.Tensor, side_workspace: Optional[torch.Tensor] = None, overlap: bool = True,
                 gemm_mode: int = 0, wk: bool = True):
        self.L = lib()
        # wk: small-batch products run with the K split inside the workgroup and their consumer fused into the store
        # (wfd_gemm_f32_wk: one launch, no slabs) wherever the library picks that form, in every operand mode
        self.wk = bool(wk)
        # operand mode of the plan's contractions (include/wfdgan.h, wfd_set_gemm_mode): 0 f32-input MFMA (parity path),
        # 1 bf16 operands, 2 split-bf16 x3.  Set around run(), so plans of different modes can coexist in a process.
        self.gemm_mode = int(gemm_mode)
        self.two = side_workspace is not None          # side lane available (own slab workspace)
        self.ws_lane = [workspace, side_workspace if self.two else workspace]
        self.calls: List[Tuple] = []
        # device tensors that only this plan's launches reference by raw pointer (descriptor tables): they live as long as
        # the plan (and any plan it is concatenated into, and the graphs captured from it, which keep the plan alive)
        self.keep: List[torch.Tensor] = []
        # thunks that reset host-side collective state (parallel.GradSync.pending ...) when a capture of this plan is abandoned
        self.abort_hooks: List = []
        self._batch = None                             # wk products being collected for one launch (batch())
        self.lane = 0                                  # stream lane of the launches being added
        self.wsl = 0                                   # lane whose slab workspace they use
        self.overlap = self.two and overlap            # side lane on its own stream (else: same stream, in order)

    @property
    def ws(self):
        return self.ws_lane[self.wsl]

    @property
    def ws_floats(self):
        return self.ws_lane[self.wsl].numel()

    def pick_splits(self, M: int, N: int, K: int) -> int:
        return self.L.gemm_pick_splits_mode(M, N, K, self.gemm_mode)

    # -- generic ------------------------------------------------------------------------------------------
    def add(self, fn, *args):
        self.calls.append((fn, args, self.lane))

    def extend(self, other: "Plan"):
        if other.gemm_mode != self.gemm_mode:
            raise ValueError("plans of different gemm modes cannot be concatenated")
        self.calls += other.calls
        self.keep += other.keep
        self.abort_hooks += [h for h in other.abort_hooks if h not in self.abort_hooks]

    def host(self, thunk, capturable: bool = False):
        """A host-side action between launches (a collective).  ``capturable``: the action only enqueues device work on
        (or ordered after) the current stream and may therefore run INSIDE a HIP-graph capture -- a torch.distributed
        collective on the "nccl" (RCCL) backend -- so the whole step stays one graph; any other action splits the plan into
        separately captured stages with the action run eagerly in between."""
        self.calls.append((HOST, (thunk, bool(capturable)), 0))

    def fork(self):
        """Following side-lane launches may start once everything issued on the main lane so far is done."""
        if self.two:
            self.calls.append((FORK, (), 1))

    def join(self):
        if self.two:
            self.calls.append((JOIN, (), 1))

    class _Side:
        def __init__(self, plan, stream):
            self.plan, self.stream = plan, stream

        def __enter__(self):
            if self.plan.two:
                self.plan.wsl = 1
                if self.stream:
                    self.plan.lane = 1

        def __exit__(self, *exc):
            self.plan.lane = 0
            self.plan.wsl = 0

    def side(self, stream: bool = True):
        """``with plan.side(): ...`` issues the enclosed launches on the side lane with the side lane's slab workspace (the
        stream part is a no-op without overlap).  stream = False: the side WORKSPACE only (slabs that must outlive later
        main-lane products), launches stay on the main lane."""
        return Plan._Side(self, stream)

    def run(self, serial: bool = False):
        prev = self.L.get_gemm_mode()
        if prev != self.gemm_mode:
            self.L.set_gemm_mode(self.gemm_mode)
        try:
            self._run(serial)
        finally:
            if prev != self.gemm_mode:
                self.L.set_gemm_mode(prev)

    def _run(self, serial: bool = False):
        main = torch.cuda.current_stream()
        side = None
        if self.overlap and not serial:
            key = main.device
            side = Plan._side_streams.get(key)
            if side is None:
                side = Plan._side_streams[key] = torch.cuda.Stream(device=main.device)
        ms = main.cuda_stream
        ss = side.cuda_stream if side is not None else ms
        for fn, a, lane in self.calls:
            if fn is FORK:
                if side is not None:
                    side.wait_stream(main)
            elif fn is JOIN:
                if side is not None:
                    main.wait_stream(side)
            elif fn is HOST:
                a[0]()
            else:
                fn(*a, ss if lane else ms)

    def stages(self, split_all: bool = False):
        """[Plan | thunk, ...]: maximal runs of launches separated by host actions (capturable ones stay inside their run
        unless ``split_all``)."""
        mk = lambda: Plan(self.ws_lane[0], self.ws_lane[1] if self.two else None, self.overlap, self.gemm_mode, self.wk)
        out, cur = [], mk()
        cur.keep = self.keep
        for fn, a, lane in self.calls:
            if fn is HOST and (split_all or not a[1]):
                if len(cur):
                    out.append(cur)
                out.append(a[0])
                cur = mk()
                cur.keep = self.keep
            else:
                cur.calls.append((fn, a, lane))
        if len(cur):
            out.append(cur)
        return out

    def __len__(self):
        return len(self.calls)

    @property
    def has_collectives(self) -> bool:
        """Capturable host actions inside: capture with capture_error_mode="thread_local" (the collective backend's
        watchdog thread issues HIP calls of its own while this thread captures)."""
        return any(fn is HOST for fn, _, _ in self.calls)

    # -- contractions -------------------------------------------------------------------------------------
    def wk_cfg(self, layout: int, A, lda: int, B, ldb: int, M: int, N: int, K: int) -> int:
        """Configuration of the split-inside-the-workgroup form for this product, 0 = use the slab form."""
        if not self.wk or layout not in (NT, NN):
            return 0
        if ptr(A) % 16 or ptr(B) % 16 or lda % 4 or ldb % 4:
            return 0
        if (layout, M, N, K) in WK_OVERRIDE:
            return WK_OVERRIDE[(layout, M, N, K)]
        return self.L.gemm_wk_pick(layout, M, N, K)

    def _wk(self, layout, A, lda, B, ldb, M, N, K, cfg, epi, out, ldo, out2=None, ldo2=0, scale=None, shift=None, act=0,
            D=None, ldd=0, dmode=0, rowscale=None, mask=None, ldmask=0, slope=0.0):
        args = (layout, ptr(A), lda, ptr(B), ldb, M, N, K, epi, ptr(out), ldo, ptr(out2), ldo2, ptr(scale), ptr(shift), act,
                ptr(D), ldd, dmode, ptr(rowscale), ptr(mask), ldmask, slope, cfg)
        if self._batch is not None:
            self._batch.append(args)
        else:
            self.add(self.L.gemm_f32_wk, *args)

    class _Batch:
        def __init__(self, plan):
            self.plan = plan

        def __enter__(self):
            self.plan._batch = []

        def __exit__(self, *exc):
            calls, self.plan._batch = self.plan._batch, None
            self.plan._flush_batch(calls)

    def batch(self):
        """``with plan.batch(): ...``: the split-inside-the-workgroup products built inside are INDEPENDENT of each other (the
        caller's promise) and go out as one launch (wfd_gemm_f32_wk_multi) when they share layout and configuration;
        otherwise, and for everything else built inside, one launch each as usual."""
        return Plan._Batch(self)

    def _flush_batch(self, calls):
        import ctypes
        import struct
        same = len({(c[0], c[23]) for c in calls}) == 1
        if len(calls) < 2 or len(calls) > 3 or not same or calls[0][23] not in (3, 4) or \
                os.environ.get("WFD_WK_MULTI", "1") == "0":      # (the switch: same-session A/B)
            for c in calls:
                self.add(self.L.gemm_f32_wk, *c)
            return
        words = []
        for c in calls:
            w = list(c)
            w[22] = struct.unpack("<I", struct.pack("<f", float(c[22])))[0]      # slope: the float's bits
            words += [int(v) for v in w]
        arr = (ctypes.c_longlong * len(words))(*words)
        self.keep.append(arr)
        self.add(self.L.gemm_f32_wk_multi, arr, len(calls))

    def gemm(self, layout: int, A, lda: int, B, ldb: int, M: int, N: int, K: int, out=None, ldp: Optional[int] = None,
             splits: Optional[int] = None) -> Slabs:
        """Raw product into the slab workspace (or straight into ``out`` with splits = 1)."""
        cfg = self.wk_cfg(layout, A, lda, B, ldb, M, N, K) if splits is None else 0
        if cfg:
            # one launch, one "slab": the row / column reducing consumers (BatchNorm, head, penalty) read it as such
            if out is None:
                ldp = rup(N, 4)
                if M * ldp > self.ws_floats:
                    raise RuntimeError("slab workspace too small")
                self._wk(layout, A, lda, B, ldb, M, N, K, cfg, 0, self.ws, ldp)
                return Slabs(ptr(self.ws), 1, M * ldp, ldp, M, N)
            self._wk(layout, A, lda, B, ldb, M, N, K, cfg, 0, out, ldp)
            return Slabs(ptr(out), 1, 0, ldp, M, N)
        if out is not None:
            # a long contraction that alone cannot fill the chip (weight gradients of a large batch: K = rows of the
            # batch) is split over K into slabs and summed into `out`; short ones go straight into `out`
            ns = self.pick_splits(M, N, K) if K >= 1024 else 1
            lds = rup(N, 4)
            while ns > 1 and ns * M * lds > self.ws_floats:
                ns -= 1
            if ns > 1:
                self.add(self.L.gemm_f32, layout, ptr(A), lda, ptr(B), ldb, ptr(self.ws), lds, M * lds, M, N, K, ns)
                self.affine_act(Slabs(ptr(self.ws), ns, M * lds, lds, M, N), None, None, ACT_NONE, out, ldp)
            else:
                self.add(self.L.gemm_f32, layout, ptr(A), lda, ptr(B), ldb, ptr(out), ldp, 0, M, N, K, 1)
            return Slabs(ptr(out), 1, 0, ldp, M, N)
        ldp = rup(N, 4)
        ns = splits or self.pick_splits(M, N, K)
        ss = M * ldp
        while ns > 1 and ns * ss > self.ws_floats:
            ns -= 1
        if ns * ss > self.ws_floats:
            raise RuntimeError(f"slab workspace too small: need {ns * ss} floats, have {self.ws_floats}")
        self.add(self.L.gemm_f32, layout, ptr(A), lda, ptr(B), ldb, ptr(self.ws), ldp, ss, M, N, K, ns)
        return Slabs(ptr(self.ws), ns, ss, ldp, M, N)

    def gemm_rms(self, layout: int, A, lda: int, B, ldb: int, M: int, N: int, K: int, P, ldp: int, SQ, lr: float,
                 alpha: float = 0.99, eps: float = 1e-8):
        """A weight gradient consumed in its own store: RMSprop on the parameter matrix ``P`` / square average ``SQ``
        (wfd_gemm_f32_rmsprop); the gradient is never written.  Unsplit contraction only."""
        self.add(self.L.gemm_f32_rmsprop, layout, ptr(A), lda, ptr(B), ldb, M, N, K, ptr(P), ldp, ptr(SQ), lr, alpha, eps)

    def gemm_rms_multi(self, problems, lr: float, alpha: float = 0.99, eps: float = 1e-8, colsum=None) -> bool:
        """Several independent weight gradients consumed in their stores as ONE launch (wfd_gemm_f32_rmsprop_multi);
        problems: [(A, lda, B, ldb, M, N, K, P, ldp, SQ)] of TN products, at most 4 per launch.  Falls back to one launch
        each where the fast form does not apply (narrow or unaligned matrices).
        colsum = (desc_ptr, nseg, blocks, gbase, pbase, sqbase, numel, state, inc0, inc1): the launch also carries that
        wfd_colsum_multi_rmsprop (wfd_gemm_f32_rmsprop_multi_colsum); returns True when it did (else the caller launches it)."""
        import ctypes
        # (every operand mode: the multi launch is the f32-operand kernel -- contractions over the batch rows only)
        ok = all(N > 32 and ptr(A) % 16 == 0 and ptr(B) % 16 == 0 and lda % 4 == 0 and ldb % 4 == 0
                 for A, lda, B, ldb, M, N, K, P, ldp, SQ in problems)
        if not ok or len(problems) < 2 or os.environ.get("WFD_RMS_MULTI", "1") == "0":      # (the switch: same-session A/B)
            for A, lda, B, ldb, M, N, K, P, ldp, SQ in problems:
                self.gemm_rms(TN, A, lda, B, ldb, M, N, K, P, ldp, SQ, lr, alpha, eps)
            return False
        if colsum is not None and len(problems) <= 4:
            words = []
            for A, lda, B, ldb, M, N, K, P, ldp, SQ in problems:
                words += [ptr(A), lda, ptr(B), ldb, M, N, K, ptr(P), ldp, ptr(SQ)]
            arr = (ctypes.c_longlong * len(words))(*words)
            self.keep.append(arr)
            if len(colsum) == 10:
                desc, nseg, blk, gbase, pbase, sqbase, numel, state, inc0, inc1 = colsum
                self.add(self.L.gemm_f32_rmsprop_multi_colsum, arr, len(problems), lr, alpha, eps, desc, nseg, blk, gbase, pbase,
                         sqbase, numel, state, inc0, inc1)
            else:       # + (ranges, nranges) + the 14 arguments of wfd_bn_running_update_layers: wfd_g_end's work as well
                desc, nseg, blk, gbase, pbase, sqbase, numel, state, inc0, inc1, ranges, nranges = colsum[:12]
                self.add(self.L.gemm_f32_rmsprop_multi_end, arr, len(problems), lr, alpha, eps, desc, nseg, blk, gbase, pbase,
                         sqbase, numel, ranges, nranges, state, inc0, inc1, *colsum[12:])
            return True
        for a in range(0, len(problems), 4):
            chunk = problems[a:a + 4]
            words = []
            for A, lda, B, ldb, M, N, K, P, ldp, SQ in chunk:
                words += [ptr(A), lda, ptr(B), ldb, M, N, K, ptr(P), ldp, ptr(SQ)]
            arr = (ctypes.c_longlong * len(words))(*words)
            self.keep.append(arr)
            self.add(self.L.gemm_f32_rmsprop_multi, arr, len(chunk), lr, alpha, eps)
        return False

    def gemm_multi_grad(self, problems, colsum=None) -> bool:
        """Several independent weight gradients STORED by ONE launch (wfd_gemm_f32_multi_grad; the data-parallel form, whose
        gradients have to meet the other ranks' before the optimizer sees them); problems: [(A, lda, B, ldb, M, N, K, G, ldg)]
        of TN products, at most 4.  colsum = (desc_ptr, nseg, blocks): the launch also carries that wfd_colsum_multi.  Returns
        True when it did; where the fast form does not apply the products go out one by one and the caller launches the
        column sums."""
        import ctypes
        ok = all(N > 32 and ptr(A) % 16 == 0 and ptr(B) % 16 == 0 and lda % 4 == 0 and ldb % 4 == 0
                 for A, lda, B, ldb, M, N, K, G, ldg in problems)
        tiles = sum(-(-M // 64) * -(-N // 64) for A, lda, B, ldb, M, N, K, G, ldg in problems)
        if not ok or not 2 <= len(problems) <= 4 or tiles < 8 or os.environ.get("WFD_RMS_MULTI", "1") == "0":
            for A, lda, B, ldb, M, N, K, G, ldg in problems:
                self.gemm(TN, A, lda, B, ldb, M, N, K, out=G, ldp=ldg)
            return False
        words = []
        for A, lda, B, ldb, M, N, K, G, ldg in problems:
            words += [ptr(A), lda, ptr(B), ldb, M, N, K, ptr(G), ldg, 0]
        arr = (ctypes.c_longlong * len(words))(*words)
        self.keep.append(arr)
        cs = colsum if colsum is not None else (0, 0, 0)
        self.add(self.L.gemm_f32_multi_grad, arr, len(problems), *cs)
        return colsum is not None

    def gemm_pair(self, tn, other, pair: bool = True) -> Slabs:
        """A weight gradient and the next product of the backward chain, which do not depend on each other, as ONE launch
        (wfd_gemm_f32_pair): tn = (A, lda, B, ldb, M, N, K, out, ldp) is the TN product written straight to ``out``;
        other = (layout, A, lda, B, ldb, M, N, K) goes to slabs, which are returned.  Falls back to two launches where
        the pair kernel does not apply (bf16-pipe modes, unaligned operands, narrow N, a K-split weight gradient)."""
        A1, lda1, B1, ldb1, M1, N1, K1, out, ldp1 = tn
        lay, A2, lda2, B2, ldb2, M2, N2, K2 = other
        al = lambda x, ld: ptr(x) % 16 == 0 and ld % 4 == 0
        ok = (pair and self.gemm_mode == 0 and lay in (NT, NN) and N1 > 32 and N2 > 32 and K1 < 1024 and al(A1, lda1)
              and al(B1, ldb1) and al(A2, lda2) and al(B2, ldb2))
        if not ok:
            self.gemm(TN, A1, lda1, B1, ldb1, M1, N1, K1, out=out, ldp=ldp1)
            return self.gemm(lay, A2, lda2, B2, ldb2, M2, N2, K2)
        ldp2 = rup(N2, 4)
        ns = self.pick_splits(M2, N2, K2)
        ss = M2 * ldp2
        while ns > 1 and ns * ss > self.ws_floats:
            ns -= 1
        if ns * ss > self.ws_floats:
            raise RuntimeError(f"slab workspace too small: need {ns * ss} floats, have {self.ws_floats}")
        self.add(self.L.gemm_f32_pair, ptr(A1), lda1, ptr(B1), ldb1, ptr(out), ldp1, M1, N1, K1, lay, ptr(A2), lda2, ptr(B2),
                 ldb2, ptr(self.ws), ldp2, ss, M2, N2, K2, ns)
        return Slabs(ptr(self.ws), ns, ss, ldp2, M2, N2)

    def conv8(self, X, Bn: int, L: int, Cin: int, pad: int, Wp, Cout: int) -> Slabs:
        M, K = Bn * L, 8 * Cin
        ldp = rup(Cout, 4)
        ns = self.pick_splits(M, Cout, K)
        ss = M * ldp
        while ns > 1 and ns * ss > self.ws_floats:
            ns -= 1
        if ns * ss > self.ws_floats:
            raise RuntimeError("slab workspace too small for conv")
        self.add(self.L.conv8_gemm_f32, ptr(X), Bn, L, Cin, pad, ptr(Wp), Cout, ptr(self.ws), ldp, ss, ns)
        return Slabs(ptr(self.ws), ns, ss, ldp, M, Cout)

    # -- contraction + element-wise consumer: fused into one launch when a single split fills the chip ------------
    def gemm_affine(self, layout, A, lda, B, ldb, M, N, K, scale, shift, act, out, ldo):
        cfg = self.wk_cfg(layout, A, lda, B, ldb, M, N, K)
        if cfg:
            self._wk(layout, A, lda, B, ldb, M, N, K, cfg, 1, out, ldo, scale=scale, shift=shift, act=act)
        elif self.pick_splits(M, N, K) == 1:
            self.add(self.L.gemm_f32_fused, layout, ptr(A), lda, ptr(B), ldb, ptr(out), ldo, M, N, K, 1, ptr(scale),
                     ptr(shift), act, 0, 0, 0)
        else:
            self.affine_act(self.gemm(layout, A, lda, B, ldb, M, N, K), scale, shift, act, out, ldo)

    def gemm_mul(self, layout, A, lda, B, ldb, M, N, K, D, ldd, dmode, colscale, out, ldo, rowscale=None, out2=None):
        """out = product * f(D) * colscale[n]; out2 = out * rowscale[m] (wfd_ep_mul's math) in as few launches as the shape
        allows."""
        cfg = self.wk_cfg(layout, A, lda, B, ldb, M, N, K)
        if cfg:
            self._wk(layout, A, lda, B, ldb, M, N, K, cfg, 2, out, ldo, out2=out2, ldo2=ldo, scale=colscale, D=D, ldd=ldd,
                     dmode=dmode, rowscale=rowscale if out2 is not None else None)
        elif self.pick_splits(M, N, K) == 1 and out2 is None:
            self.add(self.L.gemm_f32_fused, layout, ptr(A), lda, ptr(B), ldb, ptr(out), ldo, M, N, K, 2, ptr(colscale),
                     0, 0, ptr(D), ldd, dmode)
        else:
            self.mul(self.gemm(layout, A, lda, B, ldb, M, N, K), D, ldd, dmode, colscale, rowscale, out, out2, ldo)

    def gemm_lrelu_drop(self, layout, A, lda, B, ldb, M, N, K, bias, mask, ldmask, H, ldh, D, ldd, slope: float = 0.2):
        """A critic hidden layer (product + wfd_ep_lrelu_drop's math)."""
        cfg = self.wk_cfg(layout, A, lda, B, ldb, M, N, K)
        if cfg:
            self._wk(layout, A, lda, B, ldb, M, N, K, cfg, 3, H, ldh, out2=D, ldo2=ldd, shift=bias, mask=mask, ldmask=ldmask,
                     slope=slope)
        else:
            self.lrelu_drop(self.gemm(layout, A, lda, B, ldb, M, N, K), bias, mask, ldmask, H, ldh, D, ldd, slope)

    def gemm_pair_mul(self, tn, other, D, ldd, dmode, out, ldo, pair: bool = True):
        """A weight gradient (tn, written in full) and the next product of the backward chain with its `o d` factor applied:
        TN launch + one fused launch where the split-inside-the-workgroup form applies, else the pair launch (or two
        launches) + wfd_ep_mul."""
        lay, A2, lda2, B2, ldb2, M2, N2, K2 = other
        cfg = self.wk_cfg(lay, A2, lda2, B2, ldb2, M2, N2, K2)
        if cfg:
            A1, lda1, B1, ldb1, M1, N1, K1, o1, ldp1 = tn
            self.gemm(TN, A1, lda1, B1, ldb1, M1, N1, K1, out=o1, ldp=ldp1)
            self._wk(lay, A2, lda2, B2, ldb2, M2, N2, K2, cfg, 2, out, ldo, D=D, ldd=ldd, dmode=dmode)
        else:
            self.mul(self.gemm_pair(tn, other, pair=pair), D, ldd, dmode, None, None, out, None, ldo)

    def conv_wk_cfg(self, X, Wp, M: int, Cin: int, Cout: int) -> int:
        if not self.wk or Cin % 32 or ptr(X) % 16 or ptr(Wp) % 16 or self.L.conv8_direct_mode(M, Cin, Cout, self.gemm_mode):
            return 0
        if (3, M, Cout, 8 * Cin) in WK_OVERRIDE:
            return WK_OVERRIDE[(3, M, Cout, 8 * Cin)]
        return self.L.gemm_wk_pick(3, M, Cout, 8 * Cin)

    def conv8_affine(self, X, Bn, L, Cin, pad, Wp, Cout, scale, shift, act, out):
        cfg = self.conv_wk_cfg(X, Wp, Bn * L, Cin, Cout)
        if cfg:
            self.add(self.L.conv8_f32_wk, ptr(X), Bn, L, Cin, pad, ptr(Wp), Cout, 1, ptr(out), Cout, ptr(scale), ptr(shift), act,
                     0, 0, 0, cfg)
        elif self.pick_splits(Bn * L, Cout, 8 * Cin) == 1 or self.L.conv8_direct_mode(Bn * L, Cin, Cout, self.gemm_mode):
            self.add(self.L.conv8_gemm_f32_fused, ptr(X), Bn, L, Cin, pad, ptr(Wp), Cout, ptr(out), Cout, 1, ptr(scale),
                     ptr(shift), act, 0, 0, 0)
        else:
            self.affine_act(self.conv8(X, Bn, L, Cin, pad, Wp, Cout), scale, shift, act, out, Cout)

    def conv8_affine_pool(self, X, Bn, L, Cin, pad, Wp, Cout, scale, shift, act, out, PO):
        """conv8_affine followed by the k = 8 / stride 8 max-pool of its output; one launch where the LDS-resident Cin = 32
        form applies (wfd_conv8_gemm_f32_fused_pool), else the two launches."""
        fuse = (self.gemm_mode == 0 and Cin == 32 and pad == 3 and L % 8 == 0 and scale is not None and shift is not None
                and not self.conv_wk_cfg(X, Wp, Bn * L, Cin, Cout) and self.L.conv8_direct_mode(Bn * L, Cin, Cout, 0)
                and os.environ.get("WFD_POOL_FUSE", "1") != "0")          # (the switch: same-session A/B)
        if fuse:
            self.add(self.L.conv8_gemm_f32_fused_pool, ptr(X), Bn, L, Cin, pad, ptr(Wp), Cout, ptr(out), Cout, ptr(scale),
                     ptr(shift), act, ptr(PO))
        else:
            self.conv8_affine(X, Bn, L, Cin, pad, Wp, Cout, scale, shift, act, out)
            self.add(self.L.df_pool8_fwd, ptr(out), Bn, L, Cout, ptr(PO))

    def conv8_mul(self, X, Bn, L, Cin, pad, Wp, Cout, D, ldd, dmode, colscale, out):
        cfg = self.conv_wk_cfg(X, Wp, Bn * L, Cin, Cout)
        if cfg:
            self.add(self.L.conv8_f32_wk, ptr(X), Bn, L, Cin, pad, ptr(Wp), Cout, 2, ptr(out), Cout, ptr(colscale), 0, 0,
                     ptr(D), ldd, dmode, cfg)
        elif self.pick_splits(Bn * L, Cout, 8 * Cin) == 1 or self.L.conv8_direct_mode(Bn * L, Cin, Cout, self.gemm_mode):
            self.add(self.L.conv8_gemm_f32_fused, ptr(X), Bn, L, Cin, pad, ptr(Wp), Cout, ptr(out), Cout, 2,
                     ptr(colscale), 0, 0, ptr(D), ldd, dmode)
        else:
            self.mul(self.conv8(X, Bn, L, Cin, pad, Wp, Cout), D, ldd, dmode, colscale, None, out, None, Cout)

    # -- epilogues ----------------------------------------------------------------------------------------
    def affine_act(self, s: Slabs, scale, shift, act: int, out, ldo: int):
        self.add(self.L.ep_affine_act, s.ptr, s.ns, s.ss, s.ldp, s.M, s.N, ptr(scale), ptr(shift), act, ptr(out), ldo)

    def lrelu_drop(self, s: Slabs, bias, mask, ldmask: int, H, ldh: int, D, ldd: int, slope: float = 0.2):
        self.add(self.L.ep_lrelu_drop, s.ptr, s.ns, s.ss, s.ldp, s.M, s.N, ptr(bias), ptr(mask), ldmask, slope,
                 ptr(H), ldh, ptr(D), ldd)

    def mul(self, s: Slabs, D, ldd: int, dmode: int, colscale, rowscale, out, out2, ldo: int):
        self.add(self.L.ep_mul, s.ptr, s.ns, s.ss, s.ldp, s.M, s.N, ptr(D), ldd, dmode, ptr(colscale), ptr(rowscale),
                 ptr(out), ptr(out2), ldo)
