"""GPU: the pipelined fast-mode driver (generator forward of iteration i+1 on a second stream while iteration i's
critic update runs).  Every pipelined iteration is replayed on a plain sequential engine from the same starting
parameters with the batch / z / dropout masks the pipelined step actually used: the updated parameters must be
bit-identical (same kernels, same inputs) -- i.e. overlapping changed nothing but the schedule."""
import pytest
import torch

from oracle import wfdgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(S, C, Z, B, st, **kw):
    from wfd_gan_amd.engine import TrainEngine
    return TrainEngine(S, C, Z, B, st.scaler_min, st.scaler_max, {k: v.detach() for k, v in st.G.items()},
                       {k: v.detach() for k, v in st.D.items()}, st.DF, device=DEV, **kw)


@pytest.mark.parametrize("graphs", [False, True])
@pytest.mark.parametrize("penalty", ["div", "gp"])
def test_pipelined_iterations_equal_sequential_replay(graphs, penalty):
    from wfd_gan_amd.engine import PipelinedTrainer
    S, C, Z, B = 101, 6, 24, 16
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
    O.perturb_df_bn(st.DF, 4)
    eng = _engine(S, C, Z, B, st, penalty=penalty, seed=11)
    eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
    pipe = PipelinedTrainer(eng, graphs=graphs)
    ref = _engine(S, C, Z, B, st, penalty=penalty)
    seen = []
    n_iter = 2 * eng.iters_per_epoch + 2                   # crosses two epoch boundaries (reshuffles)
    for i in range(n_iter):
        torch.cuda.synchronize()
        pre = dict(G=eng.Gp.flat.clone(), Gsq=eng.Gp.sq.clone(), D=eng.Dp.flat.clone(), Dsq=eng.Dp.sq.clone())
        do_g = pipe.iteration(i)
        torch.cuda.synchronize()
        s = pipe.slot(i)
        assert do_g == (i % 3 == 0)
        # replay on the sequential parity engine
        ref.Gp.flat.copy_(pre["G"]); ref.Gp.sq.copy_(pre["Gsq"]); ref.Dp.flat.copy_(pre["D"]); ref.Dp.sq.copy_(pre["Dsq"])
        ref.set_batch(s.real_rows[:, :S], s.buf["labels"], s.buf["ZIN"][:, :Z])
        if penalty == "gp":
            ref.set_gp_alpha(s.buf["GPA"])
        ref.set_masks([s.buf[f"MK{k}"] for k in (1, 2, 3)], [s.buf[f"gMK{k}"] for k in (1, 2, 3)] if do_g else None)
        ref.d_step()
        if do_g:
            ref.g_step()
        torch.cuda.synchronize()
        assert torch.equal(ref.fake_rows, s.fake_rows), f"it{i}: generated rows"
        assert torch.equal(ref.Dp.flat, eng.Dp.flat), f"it{i}: critic parameters"
        assert torch.equal(ref.Gp.flat, eng.Gp.flat), f"it{i}: generator parameters"
        assert torch.equal(ref.Dp.sq, eng.Dp.sq) and torch.equal(ref.Gp.sq, eng.Gp.sq)
        seen.append(s.buf["labels"].clone())
    pipe.check_length_index()
    assert torch.isfinite(eng.Dp.flat).all() and torch.isfinite(eng.Gp.flat).all()
    # every epoch is a permutation: the label histogram of one epoch's batches equals the dataset's (full batches)
    ipe = eng.iters_per_epoch
    ep0 = torch.cat(seen[:ipe]).cpu()
    assert ep0.numel() == ipe * B
    # BatchNorm bookkeeping: one update per D-step forward + one per G-step
    assert eng.nbt["model.3"] == n_iter + (n_iter + 2) // 3


def test_pipelined_matches_itself_graph_vs_eager():
    from wfd_gan_amd.engine import PipelinedTrainer
    S, C, Z, B = 61, 4, 16, 8
    X, y = O.synth_dataset(C, 20, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    outs = []
    for graphs in (False, True):
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        eng = _engine(S, C, Z, B, st, seed=3)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        pipe = PipelinedTrainer(eng, graphs=graphs)
        for i in range(13):
            pipe.iteration(i)
        torch.cuda.synchronize()
        outs.append((eng.Dp.flat.clone(), eng.Gp.flat.clone(), {k: v.clone() for k, v in eng.Gbuf.items()}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


def test_group_graph_equals_single_iterations():
    """iteration_group (one graph replay per n_critic iterations) == the same iterations one by one, across an
    epoch boundary (where it falls back to single iterations)."""
    S, C, Z, B = 61, 4, 16, 8
    X, y = O.synth_dataset(C, 20, S - 1, 10, 13, 122, seed=1)          # 80 rows -> 10 iterations per epoch
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    outs = []
    for grouped in (False, True):
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        eng = _engine(S, C, Z, B, st, seed=3)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        eng.capture_graphs()
        assert eng.graph_group is not None
        i = 0
        while i < 24:
            if grouped:
                i += eng.iteration_group(i)
            else:
                eng.iteration(i)
                i += 1
        torch.cuda.synchronize()
        outs.append((eng.Dp.flat.clone(), eng.Gp.flat.clone(), dict(eng.nbt), eng.iter_state.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and torch.equal(outs[0][3], outs[1][3])


@pytest.mark.parametrize("penalty", ["div", "gp"])
def test_grouped_heads_equal_sequential_replay(penalty):
    """GroupedTrainer: the generator forwards of the three iterations of a group run stacked (3*B rows, BatchNorm per
    batch).  Every iteration of two groups is replayed on a sequential engine from the same parameters with the
    batch / z / masks the grouped run used.  Not bit-identical (a 3*B-row GEMM splits K differently), so: generated
    rows and gradients to 1e-5 of their scale, updated parameters to 1e-3 of their scale (RMSprop's first steps,
    lr g / (0.1 |g| + 1e-8), turn the rounding difference of a near-zero gradient into a visible fraction of the 10 lr
    step; a wrong sign would be 4e-2), BatchNorm running statistics to 1e-5."""
    from wfd_gan_amd.engine import GroupedTrainer
    S, C, Z, B = 101, 6, 24, 16
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
    O.perturb_df_bn(st.DF, 4)
    eng = _engine(S, C, Z, B, st, penalty=penalty, seed=11)
    eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
    grp = GroupedTrainer(eng, graphs=False)
    ref = _engine(S, C, Z, B, st, penalty=penalty)
    perm = eng.perm.clone()
    for gi in range(2):
        bn0 = {k: v.clone() for k, v in eng.Gbuf.items()}
        for k, v in bn0.items():
            ref.Gbuf[k].copy_(v)
        grp.head.run()
        torch.cuda.synchronize()
        for k in range(3):
            s = grp.slot(k)
            # the stacked gather took batch 3*gi + k of the permutation
            want_idx = perm[(3 * gi + k) * B:(3 * gi + k + 1) * B]
            assert torch.equal(s.buf["labels"].long(), eng.data_y[want_idx].long())
            assert torch.equal(s.real_rows[:, :S], eng.data_X[want_idx])
            pre = dict(G=eng.Gp.flat.clone(), Gsq=eng.Gp.sq.clone(), D=eng.Dp.flat.clone(), Dsq=eng.Dp.sq.clone())
            grp.rest[k].run()
            if k == 2:
                grp.gstep.run()
            torch.cuda.synchronize()
            ref.Gp.flat.copy_(pre["G"]); ref.Gp.sq.copy_(pre["Gsq"]); ref.Dp.flat.copy_(pre["D"]); ref.Dp.sq.copy_(pre["Dsq"])
            ref.set_batch(s.real_rows[:, :S], s.buf["labels"], s.buf["ZIN"][:, :Z])
            if penalty == "gp":
                ref.set_gp_alpha(s.buf["GPA"])
            ref.set_masks([s.buf[f"MK{j}"] for j in (1, 2, 3)], [s.buf[f"gMK{j}"] for j in (1, 2, 3)] if k == 2 else None)
            ref.d_step()
            if k == 2:
                ref.g_step()
            torch.cuda.synchronize()
            assert float((ref.fake_rows - s.fake_rows).abs().max()) < 1e-5, f"group {gi} slot {k}: generated rows"
            assert torch.equal(ref.buf["idx"], s.buf["idx"])
            for a, bb, what in ((ref.Dp.grad, eng.Dp.grad, "critic gradients"),) + \
                    (((ref.Gp.grad, eng.Gp.grad, "generator gradients"),) if k == 2 else ()):
                assert float((a - bb).abs().max()) <= 1e-5 * float(a.abs().max()), f"group {gi} slot {k}: {what}"
            for a, bb, what in ((ref.Dp.flat, eng.Dp.flat, "critic"), (ref.Gp.flat, eng.Gp.flat, "generator")):
                assert float((a - bb).abs().max()) <= 1e-3 * float(a.abs().max()), f"group {gi} slot {k}: {what}"
        for k, v in eng.Gbuf.items():       # 3 forward updates + the G-step's, in order
            assert float((ref.Gbuf[k] - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k
    assert not int(eng.buf["err"].item())


def test_parity_plan_survives_rebuilds_of_the_step_plans():
    """Every plan owns the descriptor table its merged column-sum launch reads (Plan.keep).  Building further D plans on
    the same engine (attach_dataset, the pipelined / grouped drivers) must leave the parity-mode plan intact: d_step() after
    those rebuilds equals d_step() on a fresh engine, bit for bit."""
    from wfd_gan_amd.engine import GroupedTrainer, PipelinedTrainer
    S, C, Z, B = 61, 4, 16, 8
    X, y = O.synth_dataset(C, 20, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    g = torch.Generator().manual_seed(0)
    z = torch.randn(B, Z, generator=g)
    masks = [(torch.rand(2 * B, n, generator=g) > 0.2).float() * 1.25 for n in (2048, 1024, 512)]
    outs = []
    for rebuild in (False, True):
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        eng = _engine(S, C, Z, B, st, seed=3)
        if rebuild:
            eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)      # builds fast_d / fast_g
            PipelinedTrainer(eng, graphs=False)                              # builds "gf" / "rest" plans on eng
            GroupedTrainer(eng, graphs=False)
            junk = [torch.full((64,), -1, dtype=torch.int64, device=eng.dev) for _ in range(64)]   # reuse freed blocks
            del junk
        eng.set_batch(Xs[:B], torch.from_numpy(y[:B]), z)
        eng.set_masks(masks, None)
        eng.d_step()
        torch.cuda.synchronize()
        outs.append((eng.Dp.flat.clone(), eng.Dp.grad.clone(), eng.buf["dstats"].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_grouped_head_noise_differs_from_single_iteration_noise():
    """The trainer runs iteration 0 of an epoch on the single-iteration plan and iterations 1.. on the stacked heads; both
    z streams count from 0, so they must use different Philox keys: iteration 1 must not repeat iteration 0's noise."""
    from wfd_gan_amd.engine import GroupedTrainer
    S, C, Z, B = 61, 4, 16, 8
    X, y = O.synth_dataset(C, 20, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
    eng = _engine(S, C, Z, B, st, seed=3)
    eng.attach_dataset(torch.from_numpy(X * scale + min_).float(), torch.from_numpy(y), shuffle_seed=5)
    grp = GroupedTrainer(eng, graphs=False)
    eng.iteration(0)
    torch.cuda.synchronize()
    z0 = eng.buf["ZIN"][:, :Z].clone()
    grp.run_group()
    torch.cuda.synchronize()
    zs = [grp.slot(k).buf["ZIN"][:, :Z].clone() for k in range(3)]
    for k, zk in enumerate(zs):
        assert not torch.equal(z0, zk), f"slot {k} repeats iteration 0's noise"
        assert abs(float(zk.mean())) < 0.5 and 0.5 < float(zk.std()) < 1.5
    assert not torch.equal(zs[0], zs[1]) and not torch.equal(zs[1], zs[2])


def test_grouped_graph_equals_eager():
    from wfd_gan_amd.engine import GroupedTrainer
    S, C, Z, B = 61, 4, 16, 8
    X, y = O.synth_dataset(C, 20, S - 1, 10, 13, 122, seed=1)          # 80 rows: 10 iterations per epoch
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    outs = []
    for graphs in (False, True):
        st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
        eng = _engine(S, C, Z, B, st, seed=3)
        eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
        grp = GroupedTrainer(eng, graphs=graphs)
        for _ in range(7):                                              # crosses epoch boundaries
            assert grp.run_group() == 3
        torch.cuda.synchronize()
        outs.append((eng.Dp.flat.clone(), eng.Gp.flat.clone(), {k: v.clone() for k, v in eng.Gbuf.items()}, dict(eng.nbt)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k
    assert outs[0][3] == outs[1][3] and outs[0][3]["model.3"] == 28
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()


def test_grouped_trainer_cursor_handover():
    """The trainer's epoch schedule: iteration 0 on the plain engine, groups from batch 1 (cursor of the stacked heads
    set by hand), the leftover full batches on the plain engine again -- every batch of the permutation exactly once."""
    from wfd_gan_amd.engine import GroupedTrainer
    S, C, Z, B = 61, 4, 16, 8
    X, y = O.synth_dataset(C, 20, S - 1, 10, 13, 122, seed=1)          # 80 rows: 10 batches
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
    eng = _engine(S, C, Z, B, st, seed=3)
    eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
    eng.auto_reshuffle = False
    eng.capture_graphs()
    grp = GroupedTrainer(eng, graphs=True)
    grp.auto_reshuffle = False
    perm = eng.perm.clone()
    used = []
    eng.iteration(0)
    used.append(eng.real_rows[:, :S].clone())
    eng.gf_state[1] = B
    i = 1
    while i + 3 <= 10:
        grp.run_group()
        used += [grp.slot(k).real_rows[:, :S].clone() for k in range(3)]
        i += 3
    eng.iter_state[1] = i * B
    while i < 10:
        eng.iteration(i)
        used.append(eng.real_rows[:, :S].clone())
        i += 1
    torch.cuda.synchronize()
    got = torch.cat(used)
    assert torch.equal(got, eng.data_X[perm])


def test_grouped_mask_draw_equals_per_iteration_draws():
    """GroupedTrainer draws the dropout keep-masks of its n critic passes and of its G-step in ONE launch at the head of
    the group (wfd_rng_dropout_mask_slots): the same numbers each iteration would draw itself at its own Philox counter
    (the counter advances once per optimizer step)."""
    from wfd_gan_amd.engine import GroupedTrainer
    from wfd_gan_amd._lib import lib
    L = lib()
    S, C, Z, B = 101, 6, 24, 16
    X, y = O.synth_dataset(C, 30, S - 1, 10, 13, 122, seed=1)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_).float()
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=2, df_seed=3)
    eng = _engine(S, C, Z, B, st, seed=11)
    eng.attach_dataset(Xs, torch.from_numpy(y), shuffle_seed=5)
    grp = GroupedTrainer(eng, graphs=False)
    assert grp.merged_masks
    assert not any(fn is L.rng_dropout_mask for p in grp.rest + [grp.gstep] for fn, _, _ in p.calls)
    M64, inc = 2**64 - 1, eng._rng_inc()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        s0 = int(eng.iter_state[0])
        grp.run_group()
        torch.cuda.synchronize()
        for k in range(3):
            e = grp.slot(k)
            want = torch.empty_like(e.buf["MKflat"])
            L.rng_dropout_mask(want.data_ptr(), want.numel(), 0.2, (e.seed + eng.GOLD) & M64, 0, s0 + k * inc, stream)
            torch.cuda.synchronize()
            assert torch.equal(want, e.buf["MKflat"]), k
        e = grp.slot(2)
        want = torch.empty_like(e.buf["gMKflat"])
        L.rng_dropout_mask(want.data_ptr(), want.numel(), 0.2, (e.seed + 4 * eng.GOLD) & M64, 0, s0 + 3 * inc, stream)
        torch.cuda.synchronize()
        assert torch.equal(want, e.buf["gMKflat"])
        assert int(eng.iter_state[0]) == s0 + 4 * inc
        assert set(e.buf["MKflat"].unique().tolist()) == {0.0, 1.25}
