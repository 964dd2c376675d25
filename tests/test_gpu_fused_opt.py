"""GPU: the single-GPU optimizer path -- RMSprop applied inside each weight gradient's store (wfd_gemm_f32_rmsprop) plus one
launch over the vectors (wfd_rmsprop_ranges) -- against (a) the two-launch form it replaces (gradient buffer + flat
wfd_rmsprop pass): bit-identical by construction (same product kernel, one shared update function), and (b) the reference
fixtures (loss.backward() + optimizer.step(), src/mlp_df_wgan_train.py:229-230, 273-275) with test_gpu_step's tolerances."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import wfdgan_oracle as O
from tests.test_gpu_step import check_tensors, exact_params
from tests.util import iter_inputs, load_iter_fixture, rebuild_state, unpack_masks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _L():
    from wfd_gan_amd._lib import lib
    return lib(), torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2048, 1501, 256), (512, 600, 64), (1401, 2048, 64), (1, 40, 24), (70, 31, 100)])
def test_gemm_rmsprop_equals_gemm_then_rmsprop(shape, mode):
    """TN product (dW = dZ^T H) with the update in the store == wfd_gemm_f32 into a gradient buffer + wfd_rmsprop, in all three
    operand modes, over two steps (the second starts from a non-zero square average); leading dimension padded to 4 floats
    as the flat parameter buffers are, pad columns untouched."""
    L, st = _L()
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    ldp = (N + 3) // 4 * 4
    lda, ldb = (M + 3) // 4 * 4, ldp
    P0 = torch.randn(M, ldp, generator=g)
    P0[:, N:] = 7.0                                           # sentinel in the pad columns
    Pa, Pb = P0.to(DEV), P0.to(DEV)
    SQa, SQb = torch.zeros(M, ldp, device=DEV), torch.zeros(M, ldp, device=DEV)
    G = torch.zeros(M, ldp, device=DEV)
    L.set_gemm_mode(mode)
    try:
        for step in range(2):
            A = torch.randn(K, lda, generator=g).to(DEV)
            B = torch.randn(K, ldb, generator=g).to(DEV)
            L.gemm_f32_rmsprop(2, A.data_ptr(), lda, B.data_ptr(), ldb, M, N, K, Pa.data_ptr(), ldp, SQa.data_ptr(), 2e-4, 0.99,
                               1e-8, st)
            L.gemm_f32(2, A.data_ptr(), lda, B.data_ptr(), ldb, G.data_ptr(), ldp, 0, M, N, K, 1, st)
            Gm = G.clone()                                    # the flat pass (pad columns carry no gradient)
            Gm[:, N:] = 0.0
            L.rmsprop(Pb.data_ptr(), Gm.data_ptr(), SQb.data_ptr(), M * ldp, 2e-4, 0.99, 1e-8, 1.0, 0, 0, 0, st)
            torch.cuda.synchronize()
            assert torch.equal(Pa[:, :N], Pb[:, :N]), f"step {step}: parameters"
            assert torch.equal(SQa[:, :N], SQb[:, :N]), f"step {step}: square average"
            assert bool((Pa[:, N:] == 7.0).all()) and bool((SQa[:, N:] == 0.0).all()), "pad columns written"
        # the product itself against fp64
        want = A[:, :M].double().t() @ B[:, :N].double()
        tol = {0: 1e-5, 1: 3e-2, 2: 1e-4}[mode]
        assert float((G[:, :N].double() - want).abs().max()) <= tol * float(want.abs().max())
    finally:
        L.set_gemm_mode(0)


def test_rmsprop_ranges_equals_flat_pass_on_those_ranges():
    L, st = _L()
    g = torch.Generator().manual_seed(3)
    n = 5000
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ranges = [(4, 260), (1000, 1003), (2048, 4999)]
    pa, pb = p0.to(DEV), p0.to(DEV)
    sqa, sqb = torch.full((n,), 0.5, device=DEV), torch.full((n,), 0.5, device=DEV)
    gd = gr.to(DEV)
    state = torch.zeros(2, dtype=torch.int64, device=DEV)
    flat = [v for r in ranges for v in r]
    arr = (ctypes.c_long * len(flat))(*flat)
    L.rmsprop_ranges(pa.data_ptr(), gd.data_ptr(), sqa.data_ptr(), arr, len(ranges), 2e-4, 0.99, 1e-8, state.data_ptr(), 3, 9, st)
    L.rmsprop(pb.data_ptr(), gd.data_ptr(), sqb.data_ptr(), n, 2e-4, 0.99, 1e-8, 1.0, 0, 0, 0, st)
    torch.cuda.synchronize()
    inside = torch.zeros(n, dtype=torch.bool)
    for lo, hi in ranges:
        inside[lo:hi] = True
    inside = inside.to(DEV)
    assert torch.equal(pa[inside], pb[inside]) and torch.equal(sqa[inside], sqb[inside])
    assert torch.equal(pa[~inside], p0.to(DEV)[~inside]) and bool((sqa[~inside] == 0.5).all())
    assert state.tolist() == [3, 9]
    with pytest.raises(Exception):
        L.rmsprop_ranges(pa.data_ptr(), gd.data_ptr(), sqa.data_ptr(), arr, 9, 2e-4, 0.99, 1e-8, 0, 0, 0, st)


def _engine(st, meta, **kw):         # noqa: E302 (used by the tests above and below)
    from wfd_gan_amd.engine import TrainEngine
    return TrainEngine(meta["seq_len"], meta["class_dim"], meta["latent_dim"], meta["batch"], meta["scaler_min"],
                       meta["scaler_max"], {k: v.detach() for k, v in st.G.items()}, {k: v.detach() for k, v in st.D.items()},
                       st.DF, lr=meta["lr"], alpha=meta["alpha"], n_critic=meta["n_critic"], **kw)


def test_fused_optimizer_in_every_operand_mode():
    """Round 3: the bf16-pipe operand modes run the f32 mode's launch structure (weight gradients through the f32-operand
    wfd_gemm_f32_rmsprop_multi launch with RMSprop in its store, merged step ends), so bench.py's operand_modes compare the
    arithmetic of the forward / chain contractions only.  One iteration in bf16x3 mode, fused vs gradient buffers + flat
    pass: the weight gradients of the fused form are exact-f32 products, those of the flat form split-bf16 ones (~1e-5
    relative), so the parameters agree to that -- not to the bit."""
    fx, meta = load_iter_fixture("cfg1")
    B = meta["batch"]
    engs = []
    for fuse in (True, False):
        st = rebuild_state(meta, fx)
        e = _engine(st, meta, fuse_opt=fuse, operands="bf16x3")
        assert bool(e.fuse_d) == fuse and bool(e.fuse_g) == fuse
        mr, mf, mg = (unpack_masks(fx, "it0_" + k, B) for k in ("mreal", "mfake", "mg"))
        e.set_batch(*(torch.from_numpy(a) for a in iter_inputs(fx, meta, 0)))
        e.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
        before = (e.Dp.flat.clone(), e.Gp.flat.clone())
        e.d_step()
        e.g_step()
        torch.cuda.synchronize()
        engs.append((e.Dp.flat - before[0], e.Gp.flat - before[1]))
    for a, b in zip(*engs):
        # first RMSprop step: lr g / (0.1 |g| + eps) -- ~10 lr in the direction of g's sign for |g| >> 1e-7, proportional to g
        # below.  A weight-gradient element is a sum over the batch rows with cancellation, so the split-bf16 products' 1e-5
        # relative error is a much larger relative error of the small elements: measured 92 % of the updates within 1e-3 of
        # the step size.  Required: most updates agree closely, nearly all agree in direction, none is off by more than the
        # two-step distance a sign flip gives.
        step = 10 * meta["lr"]
        assert ((a - b).abs() <= 1e-3 * step).float().mean().item() >= 0.85
        moved = (a.abs() > 0.5 * step) & (b.abs() > 0.5 * step)
        assert (torch.sign(a[moved]) == torch.sign(b[moved])).float().mean().item() >= 0.995
        assert float((a - b).abs().max()) <= 2.001 * step


@pytest.mark.parametrize("operands", ["f32"])
@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg2"])
def test_fused_optimizer_step_equals_flat_pass_and_reference(name, operands):
    """Every iteration of the fixture on two engines: optimizer in the gradient stores vs gradient buffers + flat pass (pair
    launches off, so that both run the same product kernels).  Parameters and square averages must stay bit-identical; the
    fused engine's parameters are also checked against the reference fixture (f32 operands)."""
    torch.set_num_threads(8)
    fx, meta = load_iter_fixture(name)
    st = rebuild_state(meta, fx)
    st64 = rebuild_state(meta, fx, dtype=torch.float64)
    B = meta["batch"]
    fused = _engine(st, meta, fuse_opt=True, operands=operands)
    plain = _engine(st, meta, fuse_opt=False, pair=False, operands=operands)
    assert fused.fuse_d and fused.fuse_g and not plain.fuse_d
    for it in range(meta["n_iter"]):
        p = f"it{it}_"
        do_g = it % meta["n_critic"] == 0
        mr, mf, mg = unpack_masks(fx, p + "mreal", B), unpack_masks(fx, p + "mfake", B), unpack_masks(fx, p + "mg", B)
        traces, labels, z = iter_inputs(fx, meta, it)
        for e in (fused, plain):
            e.set_batch(torch.from_numpy(traces), torch.from_numpy(labels), torch.from_numpy(z))
            e.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
            e.d_step()
        torch.cuda.synchronize()
        assert torch.equal(fused.Dp.flat, plain.Dp.flat), f"it{it}: critic parameters"
        assert torch.equal(fused.Dp.sq, plain.Dp.sq), f"it{it}: critic square averages"
        assert torch.equal(fused.buf["dstats"], plain.buf["dstats"])
        if operands == "f32":
            m64 = lambda ms: None if ms is None else [t.double() for t in ms]
            O.iteration(st64, torch.from_numpy(traces).double(), torch.from_numpy(labels), torch.from_numpy(z).double(),
                        do_g, m64(mr), m64(mf), m64(mg), keep=True)
            O.iteration(st, torch.from_numpy(traces).double(), torch.from_numpy(labels), torch.from_numpy(z), do_g, mr, mf,
                        mg, keep=True)
        if do_g:
            for e in (fused, plain):
                e.g_step()
            torch.cuda.synchronize()
            assert torch.equal(fused.Gp.flat, plain.Gp.flat), f"it{it}: generator parameters"
            assert torch.equal(fused.Gp.sq, plain.Gp.sq), f"it{it}: generator square averages"
        if operands == "f32":
            check_tensors(fused.Dp, fused.Dp.p, fx, p + "d_param_norms", p + "d_param_slices", f"it{it} D param",
                          (exact_params(st.D, fused.Dp.specs), exact_params(st64.D, fused.Dp.specs)))
            check_tensors(fused.Gp, fused.Gp.p, fx, p + "g_param_norms", p + "g_param_slices", f"it{it} G param",
                          (exact_params(st.G, fused.Gp.specs), exact_params(st64.G, fused.Gp.specs)))


@pytest.mark.parametrize("graphs", [False, True])
def test_grouped_fast_mode_fused_equals_flat_pass(graphs):
    """Fast mode (device-resident data, Philox, stacked generator forwards, HIP graphs): the same seeds drive a fused and an
    unfused engine through three groups of n_critic iterations -- bit-identical parameters, BatchNorm buffers and loss sums."""
    from wfd_gan_amd.engine import GroupedTrainer
    S, C, Z, B = 101, 6, 24, 16
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    out = []
    for fuse in (True, False):
        from wfd_gan_amd.engine import TrainEngine
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        O.perturb_df_bn(st.DF, 4)
        eng = TrainEngine(S, C, Z, B, st.scaler_min, st.scaler_max, {k: v.detach() for k, v in st.G.items()},
                          {k: v.detach() for k, v in st.D.items()}, st.DF, device=DEV, seed=11, fuse_opt=fuse, pair=False)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        grp = GroupedTrainer(eng, graphs=graphs)
        for _ in range(3):
            grp.run_group()
        torch.cuda.synchronize()
        grp.check_length_index()
        out.append([eng.Dp.flat.clone(), eng.Dp.sq.clone(), eng.Gp.flat.clone(), eng.Gp.sq.clone(), eng.iter_state.clone(),
                    eng.buf["dacc"].clone(), eng.buf["gacc"].clone()] + [v.clone() for v in eng.Gbuf.values()])
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), f"tensor {k} differs between the fused and the flat optimizer path"
    assert torch.isfinite(out[0][0]).all() and float((out[0][0] != 0).float().mean()) > 0.5


@pytest.mark.parametrize("graphs", [False, True])
def test_grouped_observer_ahead_is_bit_identical(graphs):
    """GroupedTrainer(observer_ahead=True): the DF observer's forward of a group runs on a side lane beside the group's
    critic updates (it reads only the generated rows and the frozen observer).  Same kernels on the same inputs, only the
    schedule differs: every state tensor must be bit-identical to the single-lane group after three groups."""
    from wfd_gan_amd.engine import GroupedTrainer, TrainEngine
    S, C, Z, B = 101, 6, 24, 16
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    out = []
    for ahead in (True, False):
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        O.perturb_df_bn(st.DF, 4)
        eng = TrainEngine(S, C, Z, B, st.scaler_min, st.scaler_max, {k: v.detach() for k, v in st.G.items()},
                          {k: v.detach() for k, v in st.D.items()}, st.DF, device=DEV, seed=11, fuse_opt=True)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        grp = GroupedTrainer(eng, graphs=graphs, observer_ahead=ahead)
        for _ in range(3):
            grp.run_group()
        torch.cuda.synchronize()
        out.append([eng.Dp.flat.clone(), eng.Dp.sq.clone(), eng.Gp.flat.clone(), eng.Gp.sq.clone(), eng.iter_state.clone(),
                    eng.buf["dacc"].clone(), eng.buf["gacc"].clone(), grp.slot(2).buf["LOG"].clone(),
                    grp.slot(2).buf["DDF"].clone()] + [v.clone() for v in eng.Gbuf.values()])
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), f"tensor {k} differs between the two schedules"


@pytest.mark.parametrize("switch,part,fewer", [("WFD_HEAD_MERGE", "head", 3), ("WFD_G_END_MERGE", "gstep", 3),
                                               ("WFD_D_END_MERGE", "rest", 1)])
@pytest.mark.parametrize("graphs", [False, True])
def test_grouped_head_merged_launches_are_bit_identical(graphs, switch, part, fewer, monkeypatch):
    """Small independent launches of a group sharing one launch, against one launch each.  WFD_HEAD_MERGE: the head of a group with its three independent draws (dropout masks of every pass, gathered batches + one-hot
    columns, z) in one launch (wfd_group_inputs) and the state advance folded into the last launch
    (wfd_ep_sigmoid_lenmask_slots_adv): 3 launches fewer.  WFD_G_END_MERGE: the end of the fused G-step (first layer's bias
    gradient and its RMSprop step, the other vectors' steps, the BatchNorm running statistics) in the first workgroups of the weight-gradient launch (wfd_gemm_f32_rmsprop_multi_end): 3 fewer.
    WFD_D_END_MERGE: the column sums that end a fused D-step (bias / head gradients with their steps, loss scalars,
    bookkeeping) in the first workgroups of the weight-gradient launch (wfd_gemm_f32_rmsprop_multi_colsum): 1 fewer per D-step.
    Every state tensor bit-identical after three groups (the second and third group see the advanced counter / cursor)."""
    from wfd_gan_amd.engine import GroupedTrainer, TrainEngine
    S, C, Z, B = 101, 6, 42, 16          # Z + C > 32: the generator's first layer takes the multi-product launch too
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    out, launches = [], []
    for merge in ("1", "0"):
        monkeypatch.setenv(switch, merge)
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        O.perturb_df_bn(st.DF, 4)
        eng = TrainEngine(S, C, Z, B, st.scaler_min, st.scaler_max, {k: v.detach() for k, v in st.G.items()},
                          {k: v.detach() for k, v in st.D.items()}, st.DF, device=DEV, seed=11, fuse_opt=True)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        grp = GroupedTrainer(eng, graphs=graphs)
        plan = getattr(grp, part)
        launches.append(len((plan[0] if isinstance(plan, list) else plan).calls))
        for _ in range(3):
            grp.run_group()
        torch.cuda.synchronize()
        out.append([eng.Dp.flat.clone(), eng.Dp.sq.clone(), eng.Dp.grad.clone(), eng.Gp.flat.clone(), eng.Gp.sq.clone(),
                    eng.iter_state.clone(), grp.slot(1).buf["dstats"].clone(),
                    eng.gf_state.clone(), eng.buf["dacc"].clone(), eng.buf["gacc"].clone(), grp.stacked["ZIN"].clone(),
                    grp.stacked["labels"].clone(), grp.stacked["XC"].clone(), grp.slot(2).buf["LOG"].clone()]
                   + [e.buf["MKflat"].clone() for e in grp.e] + [v.clone() for v in eng.Gbuf.values()])
    assert launches[1] - launches[0] == fewer
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), f"tensor {k} differs between the merged and the one-launch-each form"


@pytest.mark.parametrize("graphs", [False, True])
def test_run_iterations_cuts_groups_anywhere(graphs):
    """GroupedTrainer.run_iterations(k): a caller that stops or starts inside a group of n_critic iterations (the driver's
    --steps 20) replays the group's per-iteration pieces -- the same launches in the same order, so nine iterations cut as
    2 + 5 + 2 leave every state tensor bit-identical to three whole groups."""
    from wfd_gan_amd.engine import GroupedTrainer, TrainEngine
    S, C, Z, B = 101, 6, 42, 16
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    out = []
    for cuts in ((2, 5, 2), None):
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        O.perturb_df_bn(st.DF, 4)
        eng = TrainEngine(S, C, Z, B, st.scaler_min, st.scaler_max, {k: v.detach() for k, v in st.G.items()},
                          {k: v.detach() for k, v in st.D.items()}, st.DF, device=DEV, seed=11, fuse_opt=True)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        grp = GroupedTrainer(eng, graphs=graphs)
        if cuts is None:
            for _ in range(3):
                grp.run_group()
        else:
            for k in cuts:
                grp.run_iterations(k)
            assert grp._phase == 0
        torch.cuda.synchronize()
        out.append([eng.Dp.flat.clone(), eng.Dp.sq.clone(), eng.Gp.flat.clone(), eng.Gp.sq.clone(), eng.iter_state.clone(),
                    eng.gf_state.clone(), eng.buf["dacc"].clone(), eng.buf["gacc"].clone(), grp.stacked["XC"].clone(),
                    torch.tensor([eng.nbt["model.3"], eng.nbt["model.6"], grp._pos])]
                   + [v.clone() for v in eng.Gbuf.values()])
    for k, (a, b) in enumerate(zip(*out)):
        assert torch.equal(a, b), f"tensor {k} differs between the cut and the whole groups"


@pytest.mark.parametrize("name", ["tiny", "cfg2"])
def test_g_step_merged_launches_are_bit_identical(name, monkeypatch):
    """The G-step with its independent products of equal depth sharing launches (critic layer k beside observer FC layer k,
    the two input-gradient chains side by side: Plan.batch -> wfd_gemm_f32_wk_multi) against one launch each: same kernels
    on the same operands, so every output must be bit-identical; at config 2 the six pairs really share launches."""
    fx, meta = load_iter_fixture(name)
    B = meta["batch"]
    engs = []
    for merge in ("1", "0"):
        monkeypatch.setenv("WFD_G_MERGE", merge)
        st = rebuild_state(meta, fx)
        with torch.no_grad():
            st.D["model.9.bias"] += 0.3                  # the critic likes some fakes: the observer term is live
        engs.append(_engine(st, meta, fuse_opt=True))
    merged, plain = engs
    assert merged.g_merge and not plain.g_merge
    L = merged.L
    n_multi = sum(fn is L.gemm_f32_wk_multi for fn, _, _ in merged.plan_g.calls)
    assert not any(fn is L.gemm_f32_wk_multi for fn, _, _ in plain.plan_g.calls)
    if name == "cfg2":
        from wfd_gan_amd.plan import FORK, JOIN
        launches = lambda e: sum(fn is not FORK and fn is not JOIN for fn, _, _ in e.plan_g.calls)
        assert n_multi == 6 and launches(merged) == launches(plain) - 6
    p = "it0_"
    mr, mf, mg = unpack_masks(fx, p + "mreal", B), unpack_masks(fx, p + "mfake", B), unpack_masks(fx, p + "mg", B)
    traces, labels, z = iter_inputs(fx, meta, 0)
    for e in engs:
        e.set_batch(torch.from_numpy(traces), torch.from_numpy(labels), torch.from_numpy(z))
        e.set_masks([torch.cat([a, b]) for a, b in zip(mr, mf)], mg)
        e.d_step()
        e.g_step()
    torch.cuda.synchronize()
    assert merged.g_stats()["n_sel"] > 0
    for k in ("LOG", "DLOG", "DDF", "GDZ4", "gV", "gDL1", "gstats"):
        assert torch.equal(merged.buf[k], plain.buf[k]), k
    assert torch.equal(merged.Gp.flat, plain.Gp.flat) and torch.equal(merged.Gp.sq, plain.Gp.sq)


def test_gemm_rmsprop_multi_equals_separate_launches():
    """wfd_gemm_f32_rmsprop_multi: the critic's three weight gradients (config-2 shapes) in one launch == three
    wfd_gemm_f32_rmsprop launches, bit for bit, over two steps; a narrow matrix is rejected (the caller falls back)."""
    L, st = _L()
    g = torch.Generator().manual_seed(11)
    shapes = [(2048, 1501, 256), (1024, 2048, 256), (512, 1024, 256)]
    P, SQ, A, B = [], [], [], []
    for M, N, K in shapes:
        ldp = (N + 3) // 4 * 4
        p0 = torch.randn(M, ldp, generator=g)
        P.append([p0.to(DEV), p0.to(DEV)])
        SQ.append([torch.zeros(M, ldp, device=DEV), torch.zeros(M, ldp, device=DEV)])
    for step in range(2):
        words = []
        ops = []
        for k, (M, N, K) in enumerate(shapes):
            ldp = (N + 3) // 4 * 4
            a, b = torch.randn(K, M, generator=g).to(DEV), torch.randn(K, ldp, generator=g).to(DEV)
            ops += [a, b]
            L.gemm_f32_rmsprop(2, a.data_ptr(), M, b.data_ptr(), ldp, M, N, K, P[k][0].data_ptr(), ldp, SQ[k][0].data_ptr(),
                               2e-4, 0.99, 1e-8, st)
            words += [a.data_ptr(), M, b.data_ptr(), ldp, M, N, K, P[k][1].data_ptr(), ldp, SQ[k][1].data_ptr()]
        arr = (ctypes.c_longlong * len(words))(*words)
        L.gemm_f32_rmsprop_multi(arr, 3, 2e-4, 0.99, 1e-8, st)
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(P[k][0], P[k][1]) and torch.equal(SQ[k][0], SQ[k][1]), (step, k)
    assert float((P[0][0] - P[0][0].clone().zero_()).abs().max()) > 0
    words[5] = 24                                                # N <= 32: not the fast form
    with pytest.raises(Exception):
        L.gemm_f32_rmsprop_multi((ctypes.c_longlong * len(words))(*words), 3, 2e-4, 0.99, 1e-8, st)
