#!/usr/bin/env python3
"""bench.py -- wfd-gan hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the reference loop body (src/mlp_df_wgan_train.py:182-275) over one batch of synthetic input:
a D-step, plus a G-step when i % n_critic == 0.  Workload = BASELINE.json configs[1]: 100 sites x 1000 instances,
L = 1400 (S = 1401), batch 64 per GPU, latent 500, n_critic 3, fp32.  Inputs are resident in HBM before the timed
region.  N > 1: data parallel (weak scaling: 64 rows per GPU), gradient all-reduce over RCCL.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the fp32 MFMA GEMM core,
timed live with HIP events around each launch in an instrumented eager pass of the same step), `cpu_baseline` (the
oracle = CPU restatement of the reference loop, timed on this host; rank 0, N = 1 only) and `sampling`
(generated traces/s of the bulk sampler, the second half of BASELINE.json's metric).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0


def synth_dataset_device(n_cls, n_inst, L, n_lo, max_out, max_in, seed, device):
    """BASELINE.md config-2 recipe generated on the device: row = [n, alternating U{1..max_out}/U{1..max_in}, 0...]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N = n_cls * n_inst
    n = torch.randint(n_lo, L + 1, (N,), generator=g, device=device)
    out_v = torch.randint(1, max_out + 1, (N, L), generator=g, device=device)
    in_v = torch.randint(1, max_in + 1, (N, L), generator=g, device=device)
    odd = (torch.arange(L, device=device) % 2 == 0)          # body column 0 is feature column 1 = outgoing
    body = torch.where(odd.unsqueeze(0), out_v, in_v).to(torch.float32)
    body = body * (torch.arange(L, device=device).unsqueeze(0) < n.unsqueeze(1))
    X = torch.cat([n.to(torch.float32).unsqueeze(1), body], 1)
    y = torch.arange(n_cls, device=device).repeat_interleave(n_inst)
    return X, y


def build_engine(args, rank, world, device):
    from wfd_gan_amd.engine import TrainEngine
    from wfd_gan_amd.model import DF, Discriminator, Generator
    S, C, Z, B = args.seq_len, args.classes, args.latent_dim, args.batch
    strong = args.dp == "strong" and world > 1
    if strong:                                # the global batch of --batch rows split over the ranks, SyncBN
        assert B % world == 0, "--dp strong needs --batch divisible by the number of ranks"
        B //= world
    # weak: every rank trains on its own data; strong: all ranks hold the same data and permutation
    X, y = synth_dataset_device(C, args.inst, S - 1, 50, 75, 179, 1000 + (0 if strong else rank), device)
    dmin, dmax = X.min(0).values, X.max(0).values
    rngv = torch.where(dmax > dmin, dmax - dmin, torch.ones_like(dmax))
    Xs = (X - dmin) / rngv
    torch.manual_seed(0)                      # same replicas on every rank
    G = Generator(S, C, Z, float(dmin[0]), float(dmax[0]))
    D = Discriminator(S, C)
    F = DF(S - 1, C)
    pg = None
    eng = TrainEngine(S, C, Z, B, float(dmin[0]), float(dmax[0]), G.state_dict(), D.state_dict(), F.state_dict(),
                      lr=2e-4, alpha=0.02, n_critic=3, penalty=args.penalty, device=device, seed=1234 + rank,
                      process_group=pg, world_size=world, overlap=bool(args.overlap), rank=rank, sync_bn=strong,
                      force_sync=bool(getattr(args, "force_dp", False)),
                      operands=args.operands, wk=bool(args.wk), fuse_opt={"0": False, "1": True}.get(args.fuse_opt, args.fuse_opt))
    eng.attach_dataset(Xs, y, shuffle_seed=0 if strong else rank)
    return eng, (dmin.double().cpu().numpy(), dmax.double().cpu().numpy())


def run_steps(eng, start, n):
    """n iterations from iteration index `start`; whole groups of n_critic iterations go out as one graph replay
    where the engine offers it (re-permutes by itself at epoch ends)."""
    i, end = start, start + n
    if hasattr(eng, "run_iterations"):          # GroupedTrainer: whole groups of n_critic iterations (D D D G) as one graph;
        eng.run_iterations(n)                   # a K that is no multiple of n_critic stops inside a group and the next
        return                                  # block continues there (the group's per-iteration graphs)
    group = getattr(eng, "iteration_group", None)
    nc = getattr(eng, "n_critic", 3)
    while i < end:
        if group is not None and i % nc == 0 and i + nc <= end:
            i += group(i)
        else:
            eng.iteration(i)
            i += 1


def live_traffic(timeout_s: float = 90.0):
    """HBM bytes per launch of the contraction core, MEASURED for this run: two `rocprofv3 --pmc` passes (FETCH_SIZE, then
    WRITE_SIZE: separate runs, as MI355X_MICROARCH's HBM section prescribes; --kernel-trace only beside them) of this
    script's short un-graphed form as CHILD processes (python3 directly behind `--`), summarised by tools/pmc_traffic.py
    with the guide's unit and gfx950 corrections (KiB; FETCH x 2).  None when rocprofv3 is missing or a pass fails or times
    out -- the caller then falls back to the committed passes and says so."""
    import importlib.util
    import shutil
    import signal
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    # already under a profiler (someone runs `rocprofv3 ... -- python3 bench.py`): no nested passes
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="wfd_pmc_", dir="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--repeats", "1", "--no-graph",
             "--no-cpu-baseline", "--no-sampling", "--no-modes", "--no-extras"]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, counter), "--"] + child
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)          # the process group this function started, nothing else
                p.wait()
                return None
            if rc != 0:
                return None
        spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(ROOT, "tools", "pmc_traffic.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        res = mod.summarise(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), "live")
        return res if res["_gemm_all"]["launches"] > 0 else None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def large_batch_point(args, device, B: int = 4096):
    """BASELINE configs[3] (large-batch sweep), one point: the same step at B rows, L = 1400, f32 operands -- whole-step
    iterations/s and the contraction core's TFLOP/s (tools/sweep_batch.py's protocol: >= 0.3 s timed after >= 0.1 s warm-up;
    parity of this shape: tests/test_gpu_step.py::test_large_batch_step_exact_on_own_parameters)."""
    a2 = argparse.Namespace(**{**vars(args), "batch": B, "inst": max(200, 2 * B // 100 + 1), "penalty": "div"})
    eng, _ = build_engine(a2, 0, 1, device)
    eng.capture_graphs()
    run_steps(eng, 0, 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(eng, 6, 6)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 6
    warm = 3 * max(1, int(0.1 / per / 3))
    steps = 3 * max(10, int(0.3 / per / 3))
    run_steps(eng, 12, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(eng, 12 + warm, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = gemm_roofline(eng, rep=4, rounds=2)
    out = {"batch": B, "iterations_per_s": steps / dt, "samples_per_s": steps * B / dt, "core_tflops": roof["tflops"],
           "frac_of_f32_mfma_peak": roof["tflops"] / PEAK_F32_MFMA_TFLOPS, "avg_us_per_launch": roof["avg_us"],
           "note": "config 4 (large batch): the same step at 4096 rows per batch, L = 1400, f32 operands, single-iteration plans"}
    del eng
    torch.cuda.empty_cache()
    return out


def df_training_point(device, B: int = 128, L: int = 1400, C: int = 100, steps: int = 40):
    """DF observer training (src/train_df.py:111-161: batch 128, Adamax, train-mode BatchNorm + dropout), steps/s
    of the HIP-graph replay of wfd_gan_amd.df_engine.DFTrainEngine on synthetic bursts (tools/df_train_bench.py's protocol)."""
    from wfd_gan_amd.df_engine import DFTrainEngine
    from wfd_gan_amd.model import DF
    torch.manual_seed(0)
    eng = DFTrainEngine(L, C, B, DF(L, C).state_dict())
    g = torch.Generator().manual_seed(1)
    eng.set_batch(torch.rand(B, L, generator=g), torch.randint(0, C, (B,), generator=g))

    def step():
        eng.draw_masks()
        eng.plan.run()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    sps = steps / (time.perf_counter() - t0)
    out = {"steps_per_s": sps, "samples_per_s": sps * B, "batch": B, "launches_per_step": len(eng.plan.calls),
           "note": "DF observer training step (batch 128, L = 1400, 100 classes, fp32, Adamax), HIP-graph replay; parity: "
                   "tests/test_gpu_df_train.py; CPU oracle on 16 threads: ~30 steps/s (tools/df_train_bench.py)"}
    del graph, eng
    torch.cuda.empty_cache()
    return out


def gemm_roofline(eng, plans=None, rep=16, rounds=3):
    """Live timing of the dominant kernel (the fp32 MFMA GEMM core, all its layouts) at the launches of one group of
    three iterations (3 D-steps + 1 G-step).  Every GEMM launch of the fast-mode plans is captured `rep` times
    back-to-back in a HIP graph of its own and the replay is bracketed by HIP events on the launch stream
    (torch.cuda.Event on the current stream); a launch only writes its slabs / output, so repeating it changes no
    state.  Per-launch time = replay time / rep, which amortises the event pair and the graph launch and keeps the
    host out of the measurement.  achieved = sum(flops) / sum(time) over the launches of the group, weighted as they
    occur (D-step launches x3)."""
    from wfd_gan_amd.parallel import graph_capture_kwargs
    from wfd_gan_amd.plan import FORK, HOST, JOIN
    L = eng.L
    gemm_fns = {L.gemm_f32: "gemm", L.conv8_gemm_f32: "conv", L.gemm_f32_fused: "gemm_fused",
                L.conv8_gemm_f32_fused: "conv", L.conv8_gemm_f32_fused_pool: "conv", L.gemm_f32_pair: "pair", L.gemm_f32_wk: "gemm_wk", L.conv8_f32_wk: "conv",
                L.gemm_f32_rmsprop: "gemm_rms", L.gemm_f32_rmsprop_multi: "rms_multi", L.gemm_f32_wk_multi: "wk_multi",
                # (the same launch also carrying the step's column sums / small updates in a few extra workgroups)
                L.gemm_f32_rmsprop_multi_colsum: "rms_multi", L.gemm_f32_rmsprop_multi_end: "rms_multi",
                # (data parallel, late gradients: the same products, stored instead of consumed)
                L.gemm_f32_multi_grad: "rms_multi"}
    calls = []
    for plan, weight in (plans or ((eng.fast_d, eng.n_critic), (eng.fast_g, 1))):
        for fn, a, _lane in plan.calls:
            if fn is FORK or fn is JOIN or fn is HOST:
                continue
            kind = gemm_fns.get(fn)
            if kind is None:
                continue
            if kind == "rms_multi":      # several weight gradients in one launch: their FLOPs add
                w = a[0]
                calls.append((fn, a, sum(2.0 * w[10 * k + 4] * w[10 * k + 5] * w[10 * k + 6] for k in range(a[1])), weight))
                continue
            if kind == "wk_multi":       # independent products of one configuration in one launch: their FLOPs add
                w = a[0]
                calls.append((fn, a, sum(2.0 * w[24 * k + 5] * w[24 * k + 6] * w[24 * k + 7] for k in range(a[1])), weight))
                continue
            if kind == "pair":           # two products in one launch: their FLOPs add
                calls.append((fn, a, 2.0 * a[6] * a[7] * a[8] + 2.0 * a[17] * a[18] * a[19], weight))
                continue
            if kind == "gemm":
                M, N, K = a[8], a[9], a[10]
            elif kind == "gemm_fused":
                M, N, K = a[7], a[8], a[9]
            elif kind in ("gemm_wk", "gemm_rms"):
                M, N, K = a[5], a[6], a[7]
            else:
                M, N, K = a[1] * a[2], a[6], 8 * a[3]
            calls.append((fn, a, 2.0 * M * N * K, weight))
    L.set_gemm_mode(eng.gemm_mode)                     # the plans' operand mode (Plan.run sets it the same way)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # load every kernel variant before capturing
        for fn, a, _, _ in calls:
            fn(*a, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    flops = ms = 0.0
    n = 0
    per_call = []
    for fn, a, fl, weight in calls:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **graph_capture_kwargs()):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(rep):
                fn(*a, st)
        g.replay()
        tot = 0.0
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            tot += e0.elapsed_time(e1) / rep
        flops += fl * weight
        ms += tot / rounds * weight
        n += weight
        per_call.append((gemm_fns[fn], fl, tot / rounds * 1e3, weight, a))
    L.set_gemm_mode(0)
    return dict(launches=n, avg_flops=flops / n, avg_us=ms * 1e3 / n, tflops=flops / (ms * 1e-3) / 1e12,
                per_call=per_call)


def cpu_baseline(args, seconds=20.0):
    """The oracle (CPU restatement of the reference loop, pinned bit-exact against the reference) on this host."""
    from oracle import wfdgan_oracle as O
    S, C, Z, B = args.seq_len, args.classes, args.latent_dim, args.batch
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    X, y = O.synth_dataset(C, 8, S - 1, 50, 75, 179, seed=0)
    dmin, dmax = X.min(0), X.max(0)
    scale, min_ = O.scaler_params(dmin, dmax)
    Xs = torch.from_numpy(X * scale + min_)
    yt = torch.from_numpy(y)
    st = O.make_state(S, C, Z, dmin[0], dmax[0], seed=0, df_seed=1)
    rng = np.random.RandomState(0)
    n, t0 = 0, None
    i = 0
    while True:
        idx = rng.randint(0, Xs.shape[0], B)
        z = torch.from_numpy(rng.normal(0, 1, (B, Z))).float()
        O.iteration(st, Xs[idx], yt[idx], z, do_g=(i % 3 == 0))
        i += 1
        if i == 3:                      # warm-up of 3 iterations, then time whole groups of 3
            t0, n = time.perf_counter(), 0
        elif i > 3:
            n += 1
            if n % 3 == 0 and time.perf_counter() - t0 > seconds:
                break
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="iterations/s", cores=threads, kind="port",
                sample=f"{n} iterations of the same config-2 step (B={B}, S={S}) on {threads} host threads")


def cli_epoch(device, epochs=4, extra=()):
    """Steady-state epochs of the trainer CLI (wfd_gan_amd.mlp_df_wgan_train.main, the reference's flags) on the config-2
    dataset: 1562 full batches + a tail of 32 per epoch, the DF-accuracy pass over the epoch's ~33 k generated traces and all
    per-epoch host work included; log line to log line from epoch 2 on (start-up, graph capture and the epoch-1 checkpoint
    drop out)."""
    import logging
    import shutil
    import tempfile
    from wfd_gan_amd import mlp_df_wgan_train
    from wfd_gan_amd.model import DF
    X, y = synth_dataset_device(100, 1000, 1400, 50, 75, 179, 0, device)
    # burst counts capped at L - 1: with a count of L in the data the length head can reach index S, where the reference
    # raises IndexError (src/model.py:44) and so does this trainer at the epoch's end
    full = X[:, 0] >= 1400
    X[full, 0] = 1399
    X[full, 1400] = 0
    tmp = tempfile.mkdtemp(prefix="wfd_epoch_")
    try:
        ddir = os.path.join(tmp, "set", "feature")
        os.makedirs(ddir)
        npz = os.path.join(ddir, "raw_feature_synth.npz")
        np.savez(npz, features=X.cpu().numpy().astype(np.float64), labels=y.cpu().numpy().astype(np.int64))
        del X, y
        torch.manual_seed(0)
        ck = os.path.join(tmp, "df.ckpt")
        torch.save(DF(1400, 100).state_dict(), ck)
        stamps = []

        class Stamp(logging.Handler):
            def emit(self, record):
                if "[Epoch" in record.getMessage():
                    torch.cuda.synchronize()
                    stamps.append(time.perf_counter())

        orig = mlp_df_wgan_train.utils.init_logger

        def init_logger(*a, **k):
            lg = orig(*a, **k)
            lg.handlers = [h for h in lg.handlers if isinstance(h, logging.FileHandler)]     # keep stdout to the JSON line
            lg.addHandler(Stamp())
            return lg
        mlp_df_wgan_train.utils.init_logger = init_logger
        try:
            mlp_df_wgan_train.main(["-d", npz, "--f_model", ck, "--n_epochs", str(epochs), "--freq", "1000", "--seed", "1"]
                                   + list(extra))
        finally:
            mlp_df_wgan_train.utils.init_logger = orig
        # the epoch-1 checkpoint (savez_compressed of 33 k rows: seconds) is written AFTER epoch 1's log line: time from
        # epoch 2's line on
        ep = (stamps[-1] - stamps[1]) / (len(stamps) - 2)
        return dict(seconds_per_epoch=ep, iterations_per_s=1563 / ep, epochs_timed=len(stamps) - 2,
                    note="python -m wfd_gan_amd.mlp_df_wgan_train on the config-2 npz: 1563 iterations per epoch (1562 full "
                         "batches + the ragged tail), DF-accuracy pass over the epoch's generated traces, loss read-out and "
                         "logging included; epochs 3.. (epoch 1 carries start-up and graph capture, its checkpoint is written "
                         "during epoch 2's interval)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def launch_times(eng, plan, rep=8, rounds=2):
    """Isolated device time of EVERY launch of a plan (not only the contractions): each launch replayed `rep` times in a
    graph of its own between HIP events, as gemm_roofline does.  Launches that update state (optimizer, counters, running
    statistics) are repeated too, so the caller snapshots and restores the training state around this."""
    from wfd_gan_amd.parallel import graph_capture_kwargs
    from wfd_gan_amd.plan import FORK, HOST, JOIN
    eng.L.set_gemm_mode(eng.gemm_mode)
    tot, n = 0.0, 0
    for fn, a, _lane in plan.calls:
        if fn is FORK or fn is JOIN or fn is HOST:
            continue
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **graph_capture_kwargs()):
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(rep):
                fn(*a, st)
        g.replay()
        t = 0.0
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            t += e0.elapsed_time(e1) / rep
        tot += t / rounds
        n += 1
    eng.L.set_gemm_mode(0)
    return n, tot * 1e3          # launches, sum of their isolated times in us


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes of this same command line
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), relay their output and return the first
    non-zero exit code.  Called BEFORE this process makes any GPU call (torch.cuda.device_count() does not initialise the
    device on this stack) -- the children are new processes, never a re-exec of one that touched the GPU.  The reference
    has no launcher to mirror: src/mlp_df_wgan_train.py:98-100 picks one device."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    shared = os.environ.get("WFD_DIST_BACKEND") == "gloo"        # test rig: gloo ranks may share a GPU (never a measurement)
    if have < 1 or (have < n and not shared):
        print(f"bench.py: --gpus {n} but {have} HIP device(s) visible; refusing to run fewer ranks than asked for "
              "(WFD_DIST_BACKEND=gloo lets test ranks share a GPU)", file=sys.stderr)
        return 2
    with socket.socket() as s:                                    # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   WFD_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))
    out0, _ = procs[0].communicate()            # rank 0 prints the ONE JSON line; the others print nothing on stdout
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"bench.py: rank exit codes {bad}", file=sys.stderr)
        return next(c for _, c in bad) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq_len", type=int, default=1401)
    ap.add_argument("--classes", type=int, default=100)
    ap.add_argument("--inst", type=int, default=1000)
    ap.add_argument("--latent_dim", type=int, default=500)
    ap.add_argument("--penalty", default="div", choices=["div", "gp"])
    ap.add_argument("--operands", default="f32", choices=["f32", "bf16", "bf16x3"],
                    help="contraction operands: f32 = f32-input MFMA (parity path, the default); bf16 = bf16 operands / "
                         "fp32 accumulate; bf16x3 = split-bf16 (hi*hi + hi*lo + lo*hi) on the bf16 matrix pipe")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--overlap", type=int, default=0, help="1: independent branches of a step on a side stream")
    ap.add_argument("--dp", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = --batch rows per GPU (DDP semantics, the driver's scaling runs); strong = --batch "
                         "rows split over the GPUs with SyncBN (result-equal to one GPU)")
    ap.add_argument("--grouped", type=int, default=1,
                    help="1: wfd_gan_amd.engine.GroupedTrainer (stacked generator forwards per group of n_critic "
                         "iterations); 0: one iteration at a time")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="1: software-pipelined driver (wfd_gan_amd.engine.PipelinedTrainer; measured 3 %% slower than "
                         "the sequential engine on one MI355X, see DESIGN.md section 6); 0: sequential engine")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps iterations; the median is reported")
    ap.add_argument("--wk", type=int, default=1,
                    help="1: small-batch products with the K split inside the workgroup and the consumer fused into the "
                         "store (gemm_wk.hip); 0: split-K slabs + consumer launches everywhere")
    ap.add_argument("--fuse-opt", dest="fuse_opt", default="1", choices=["0", "1", "d", "g"],
                    help="1 (single GPU): RMSprop applied inside each weight gradient's store (wfd_gemm_f32_rmsprop), the "
                         "gradient never written; 0: gradient buffers + one flat optimizer pass (what data parallel runs)")
    ap.add_argument("--observer-ahead", dest="observer_ahead", type=int, default=0,
                    help="1: the DF observer's forward of a group runs on a side lane beside the group's critic updates")
    ap.add_argument("--force-dp", action="store_true",
                    help="one GPU: run the data-parallel code path (bucketed gradient all-reduce over a 1-rank RCCL group, "
                         "flat optimizer pass) to price its launch structure; WFD_DP_GRAPH=0 keeps the collectives out of the graphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampling", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the bf16x3 / bf16 operand-mode runs of the same step")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short WGAN-GP (--penalty gp) block, the trainer-CLI epoch and the per-launch time sum")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC passes instead of two rocprofv3 --pmc passes of this run")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly (the shape of the driver's 1-GPU command): start the N ranks ourselves, before any GPU call
        raise SystemExit(self_launch(args.gpus))

    from wfd_gan_amd import parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or plainly "
                         f"(python bench.py --gpus {args.gpus}) and bench.py starts the ranks itself")
    if world > 1 and os.environ.get("WFD_DIST_BACKEND") != "gloo" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} HIP device(s) visible")
    if os.environ.get("WFD_DIST_BACKEND") == "gloo" and local >= torch.cuda.device_count():
        local %= torch.cuda.device_count()          # test rig: several gloo ranks sharing one GPU (never a measurement)
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    if args.force_dp and world == 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)

    eng, (dmin, dmax) = build_engine(args, rank, world, device)
    strong = args.dp == "strong" and world > 1
    pipelined = bool(args.pipeline) and not strong
    grouped = bool(args.grouped) and not strong and not pipelined
    if grouped:             # the generator forwards of the n_critic iterations sharing generator parameters run stacked
        from wfd_gan_amd.engine import GroupedTrainer
        drv = GroupedTrainer(eng, graphs=not args.no_graph, observer_ahead=bool(args.observer_ahead))
        if (args.steps % eng.n_critic or args.warmup % eng.n_critic) and not args.observer_ahead:
            drv._build_parts()
    elif pipelined:         # generator forward of iteration i+1 on a second stream beside iteration i's critic update
        from wfd_gan_amd.engine import PipelinedTrainer
        drv = PipelinedTrainer(eng, graphs=not args.no_graph)
    else:
        drv = eng
        if not args.no_graph:
            eng.capture_graphs()
    run_steps(drv, 0, args.warmup)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The timed block is EXACTLY --steps iterations between barrier + synchronize on both sides, max over ranks.  It is
    # repeated --repeats times back to back (training simply continues) and the MEDIAN block time is reported, so that a
    # short block (the driver's --steps 20 is 7 ms) says the same as a long one; every block's time is in `block_ms`.
    # A --steps that is no multiple of n_critic ends a block inside a group of n_critic iterations and the next block
    # continues there (GroupedTrainer.run_iterations: the group's per-iteration graphs): blocks then differ by SCHEDULE (6 or
    # 7 G-steps and stacked generator forwards in 20 iterations), not by noise.  In that case the number of blocks is rounded
    # up to a multiple of n_critic (n_critic consecutive blocks see every phase once) and the MEAN block time is reported.
    nc = eng.n_critic
    ragged = args.steps % nc != 0
    nblocks = max(1, args.repeats)
    if ragged:
        nblocks = (nblocks + nc - 1) // nc * nc
    dts, it0 = [], args.warmup
    for _ in range(nblocks):
        barrier()
        t0 = time.perf_counter()
        run_steps(drv, it0, args.steps)
        barrier()
        dts.append(parallel.max_over_ranks(time.perf_counter() - t0, device))
        it0 += args.steps
    dt = float(np.mean(dts)) if ragged else float(np.median(dts))
    # On this synthetic recipe (burst counts up to L) the generator's length head saturates after ~2000 iterations and
    # the reference raises IndexError there (src/model.py:44; the CPU oracle does after 1.2 epochs, DESIGN.md section
    # 6).  The kernels clamp the index, so a long timing run stays valid; the condition is reported, not raised.
    try:
        drv.check_length_index()
        saturated = False
    except IndexError:
        saturated = True
    ms_per_step = dt * 1e3 / args.steps
    steps_per_s = args.steps / dt
    # Whole-job aggregate.  The unit is one pass of the loop body over one batch of --batch rows.  Weak scaling: every
    # rank performs one such pass per step on its own batch (the gradient all-reduce couples them into one optimizer
    # step on the global batch), so the job completes world passes per step.  Strong scaling: the ranks share ONE
    # batch of --batch rows per step.
    its = steps_per_s * (1 if strong else world)

    # the launches of the plans that were timed above (one group of n_critic iterations either way)
    roof = gemm_roofline(eng, [(drv.group, 1)] if grouped else None)
    # HBM bytes per contraction launch: measured for THIS run by two PMC passes of the un-graphed step in child processes
    # (headline configuration, one rank); otherwise -- and when a pass fails -- the committed passes (tools/pmc_traffic.py)
    traffic, traffic_commit, traffic_live = None, None, None
    headline = (world == 1 and grouped and args.operands == "f32" and args.batch == 64 and args.seq_len == 1401
                and args.classes == 100 and args.penalty == "div" and not args.no_graph)
    if headline and not args.no_extras and not args.no_live_traffic:
        traffic_live = live_traffic()
    if traffic_live is not None:
        traffic = traffic_live["_gemm_all"]["hbm_bytes_per_launch"]
    else:
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = pm["_gemm_all"]["hbm_bytes_per_launch"]
            traffic_commit = pm.get("_meta", {}).get("commit")
        except Exception:
            pass
    samp = None
    if not args.no_sampling:
        from wfd_gan_amd.sampler import BulkSampler
        chunk = 32768           # 0.62 of the f32 MFMA peak at 8192 rows per chunk, 0.65 from 16384 on (tools/sampler_chunk_sweep.py)
        bs = BulkSampler(eng.generator_state(), args.seq_len, args.classes, args.latent_dim, dmin, dmax, chunk=chunk,
                         device=device, seed=99 + rank)
        bs.capture()
        labels = (torch.arange(chunk, device=device) % args.classes).to(torch.int32)
        nchunks = 4

        def timed_chunks(sampler, blocks=5):
            """median time of `blocks` blocks of nchunks chunks (a block is ~15 ms: one block alone sits on whatever the
            clocks are doing after the capture; the first run of this measured 6.0 M where every other gave 9.2 M)"""
            for _ in range(4):
                sampler.run_chunk(labels)
            ts = []
            for _ in range(blocks):
                barrier()
                t = time.perf_counter()
                for _ in range(nchunks):
                    sampler.run_chunk(labels)
                barrier()
                ts.append(parallel.max_over_ranks(time.perf_counter() - t, device))
            return float(np.median(ts))

        sdt = timed_chunks(bs)
        G_MAC = (args.latent_dim + args.classes) * 512 + 512 * 1024 + 1024 * 2048 + 2048 * args.seq_len   # per trace
        tps = world * nchunks * chunk / sdt
        samp = dict(value=tps, unit="traces/s", chunk=chunk, operands="f32",
                    note="int32 burst vectors left on device; eval generator + inverse-scale/rint/trim epilogue",
                    roofline={"bound": "mfma", "achieved": tps / world * 2.0 * G_MAC / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                              "unit": "TFLOP/s", "frac": tps / world * 2.0 * G_MAC / 1e12 / PEAK_F32_MFMA_TFLOPS,
                              "flops_per_trace": 2.0 * G_MAC,
                              "note": "whole sampler chunk (4 contractions + epilogues) per GPU against the f32-input MFMA peak"})
        # the same chunks with the int32 burst vectors and lengths copied to (pinned) host memory, copy of chunk i overlapped with
        # the generation of chunk i + 1 (two device staging buffers, a copy stream): the PCIe-inclusive rate, reported beside
        # `value`, never as it
        host = [torch.empty((chunk, args.seq_len), dtype=torch.int32).pin_memory() for _ in range(2)]
        hlen = [torch.empty(chunk, dtype=torch.int32).pin_memory() for _ in range(2)]
        stage = [torch.empty_like(bs.out) for _ in range(2)]
        stlen = [torch.empty_like(bs.lens) for _ in range(2)]
        cs = torch.cuda.Stream(device=device)
        main = torch.cuda.current_stream()
        done = [None, None]
        barrier()
        t1 = time.perf_counter()
        for i in range(nchunks):
            k = i & 1
            bs.run_chunk(labels)
            if done[k] is not None:
                main.wait_event(done[k])            # the staging buffer's previous copy has left
            stage[k].copy_(bs.out); stlen[k].copy_(bs.lens)
            ev = torch.cuda.Event(); ev.record(main)
            cs.wait_event(ev)
            with torch.cuda.stream(cs):
                host[k].copy_(stage[k], non_blocking=True); hlen[k].copy_(stlen[k], non_blocking=True)
                done[k] = torch.cuda.Event(); done[k].record(cs)
        cs.synchronize()
        barrier()
        samp["d2h_inclusive_traces_per_s"] = world * nchunks * chunk / parallel.max_over_ranks(time.perf_counter() - t1, device)
        samp["d2h_bytes_per_trace"] = 4 * args.seq_len + 4
        del host, hlen, stage, stlen
        # same sampler with bf16-operand MFMA (not the parity path; tolerance in tests/test_gpu_bf16.py)
        bs16 = BulkSampler(eng.generator_state(), args.seq_len, args.classes, args.latent_dim, dmin, dmax, chunk=chunk,
                           device=device, seed=99 + rank, operands="bf16")
        bs16.capture()
        samp["bf16_operands_traces_per_s"] = world * nchunks * chunk / timed_chunks(bs16)
        # and with split-bf16 contractions over the fp32 weights (products ~1e-5 relative)
        bs3 = BulkSampler(eng.generator_state(), args.seq_len, args.classes, args.latent_dim, dmin, dmax, chunk=chunk,
                          device=device, seed=99 + rank, operands="bf16x3")
        bs3.capture()
        samp["bf16x3_operands_traces_per_s"] = world * nchunks * chunk / timed_chunks(bs3)

    # roofline of the contraction core in the mode that ran: f32-input MFMA 157.3 TF; bf16 MFMA 2.5 PF dense, of which
    # the split-bf16 mode spends three instructions per product term (peak / 3 in fp32-equivalent FLOP)
    peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": 2500.0, "bf16x3": 2500.0 / 3}[args.operands]
    dtype = {"f32": "f32", "bf16": "bf16 operands, f32 accumulate (data f32 in HBM)",
             "bf16x3": "f32 via split-bf16 (3 bf16 MFMAs per product), f32 accumulate"}[args.operands]
    core = ("fp32 MFMA core: gemm_wk_kernel (small-batch products, K split inside the workgroup, fused consumers) + "
            "gemm_f32_rms_multi_kernel / gemm_f32_kernel (weight gradients with RMSprop in the store, one launch per network) + "
            "conv8_c32_kernel" if args.operands == "f32"
            else "bf16 MFMA core over fp32 data: gemm_mx_kernel (all layouts, implicit-GEMM conv)")
    # the same step with the contractions on the bf16 matrix pipe (gemm_mx.hip): reported beside the headline, never as
    # `value` (the headline stays the f32-operand parity path unless --operands says otherwise)
    modes = None
    if world == 1 and grouped and not args.no_modes and args.operands == "f32" and not args.no_graph:
        from wfd_gan_amd.engine import GroupedTrainer
        modes = {}
        for op in ("bf16x3", "bf16"):
            a2 = argparse.Namespace(**vars(args))
            a2.operands = op
            e2, _ = build_engine(a2, rank, world, device)
            d2 = GroupedTrainer(e2, graphs=True)
            # the headline's protocol (a 90-iteration block reads ~12 % high: the clocks have not settled yet)
            n2 = max(args.steps - args.steps % eng.n_critic, eng.n_critic)
            run_steps(d2, 0, 30)
            dts2 = []
            for _ in range(3):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                run_steps(d2, 30, n2)
                torch.cuda.synchronize()
                dts2.append(time.perf_counter() - t2)
            dt2 = float(np.median(dts2))
            r2 = gemm_roofline(e2, [(d2.group, 1)])
            modes[op] = {"iterations_per_s": n2 / dt2, "core_tflops": r2["tflops"],
                         "avg_us_per_launch": r2["avg_us"]}
            del d2, e2
            torch.cuda.empty_cache()
        modes["note"] = ("same config-2 step and the SAME launch structure as the headline (weight gradients through the "
                         "f32-operand launch with RMSprop in its store, merged step ends: like for like since round 3); only the "
                         "forward / chain contractions run as bf16 MFMA over the fp32 data: bf16x3 = split-bf16 (products ~1e-5 "
                         "relative, meets the 1e-3 parity bar: tests/test_gpu_step.py), bf16 = bf16 operands (config 4 numerics)")
    # driver-visible extras (world 1, headline configuration only): the launch structure of a group, the same step with the
    # interpolated WGAN-GP penalty north_star names (src/utils.py:64-84; 3 B stacked critic rows), one trainer-CLI epoch
    extras = {}
    if world == 1 and grouped and not args.no_extras and not args.no_graph:
        from wfd_gan_amd.engine import GroupedTrainer
        snap, gfs = eng._snapshot(), eng.gf_state.clone()
        nl, us = launch_times(eng, drv.group)
        torch.cuda.synchronize()
        eng._restore(snap)
        eng.gf_state.copy_(gfs)
        group_us = ms_per_step * 1e3 * eng.n_critic
        extras["launches_per_group"] = nl
        extras["kernel_time_fraction"] = {
            "sum_isolated_launch_us": us, "group_wall_us": group_us, "frac": us / group_us,
            "note": f"sum over the {nl} launches of one group of {eng.n_critic} iterations (each replayed 8x back to back in a "
                    "graph of its own, hot operands) / wall time of a group in the timed blocks; the difference is launch "
                    "boundaries and cold first tiles"}
        if args.penalty == "div":
            a2 = argparse.Namespace(**vars(args))
            a2.penalty = "gp"
            e2, _ = build_engine(a2, rank, world, device)
            d2 = GroupedTrainer(e2, graphs=True)
            n2 = max(args.steps - args.steps % eng.n_critic, eng.n_critic)
            run_steps(d2, 0, 30)
            dts2 = []
            for _ in range(3):
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                run_steps(d2, 30, n2)
                torch.cuda.synchronize()
                dts2.append(time.perf_counter() - t2)
            r2 = gemm_roofline(e2, [(d2.group, 1)])
            extras["penalty_gp"] = {"iterations_per_s": n2 / float(np.median(dts2)), "core_tflops": r2["tflops"],
                                    "frac_of_f32_mfma_peak": r2["tflops"] / PEAK_F32_MFMA_TFLOPS,
                                    "avg_us_per_launch": r2["avg_us"], "launches_timed": r2["launches"],
                                    "note": "same config-2 step with --penalty gp (interpolates as a third row group of the "
                                            "critic input: 192 stacked rows; parity: tests/test_gpu_gp_and_cli.py)"}
            del d2, e2
            torch.cuda.empty_cache()
        if args.operands == "f32" and args.batch == 64 and args.seq_len == 1401 and args.classes == 100 and args.inst == 1000:
            extras["cli_epoch"] = cli_epoch(device)
            extras["large_batch"] = large_batch_point(args, device)
            extras["df_training"] = df_training_point(device)
    if eng._mesh is not None:
        eng._mesh.check()                 # a timed-out wait for a peer would make the line meaningless: fail before printing it
    if rank == 0:
        out = {
            "metric": "WGAN train iterations/s (100-class, L=1400, fp32; one iteration = loop body on one 64-row batch: "
                      "D-step every iteration, G-step every 3rd; N GPUs weak-scaled = N such batches per optimizer step)",
            "value": its * 1.0, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ranks_in_collective": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
            "launcher": os.environ.get("WFD_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "plain"),
            "ms_per_step": ms_per_step, "timed_blocks": len(dts), "block_stat": "mean" if ragged else "median",
            "block_ms": [round(x * 1e3, 3) for x in dts],
            "higher_is_better": True, "scaling": "strong" if (args.dp == "strong" and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 100 sites x 1000 inst synthetic bursts, L=1400 (S=1401), batch "
                                   f"{args.batch}/GPU, latent {args.latent_dim}, n_critic 3, alpha 0.02, RMSprop 2e-4, "
                                   f"penalty={args.penalty} ({'W-div k=2 p=6, what the reference runs' if args.penalty == 'div' else 'WGAN-GP interpolate'})",
                       "global_batch": eng.B * world, "optimizer_steps_per_s": steps_per_s,
                       "samples_per_s": steps_per_s * eng.B * world,
                       "parallelism": f"dp{world}", "graphs": not args.no_graph, "pipelined": pipelined,
                       "grouped_generator_forwards": grouped, "length_index_saturated": saturated,
                       "operands": args.operands, "wk_fused_products": bool(args.wk),
                       "dp_code_path_forced": bool(args.force_dp),
                       # how the ranks' gradients meet: RCCL all-reduce + flat optimizer pass (default), WFD_DP_SHARD=1
                       # reduce-scatter / shard RMSprop / all-gather, WFD_DP_MESH=1 the library's peer-to-peer launch
                       "dp_form": (("mesh" if eng.mesh_dp else "shard" if eng.shard else "allreduce") if eng.dp else None),
                       "collectives_in_graph": parallel.graph_capturable(),
                       "optimizer_in_gradient_store": {"critic": bool(eng.fuse_d), "generator": bool(eng.fuse_g)}},
            "roofline": {"bound": "mfma", "achieved": roof["tflops"], "peak": peak, "unit": "TFLOP/s",
                         "frac": roof["tflops"] / peak, "traffic": traffic if args.operands == "f32" else None,
                         "traffic_note": ("bytes/launch, MEASURED IN THIS RUN: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes "
                                          "(separate child processes) of this command's un-graphed form (--no-graph --steps 12), "
                                          "FETCH x2 per MI355X_MICROARCH HBM section; average over the contraction launches"
                                          if traffic_live is not None else
                                          "bytes/launch, NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                          "this command (--no-graph --steps 12) committed as profiles/pmc_traffic.json, taken at "
                                          f"commit {traffic_commit}; FETCH x2 per MI355X_MICROARCH HBM section"),
                         "traffic_by_kernel": ({k: round(v["hbm_bytes_per_launch"]) for k, v in traffic_live.items()
                                                if not k.startswith("_") and (k.startswith("gemm") or k.startswith("conv8_c32"))}
                                               if traffic_live is not None else None),
                         "kernel": core, "launches_timed": roof["launches"],
                         "avg_flops_per_launch": roof["avg_flops"], "avg_us_per_launch": roof["avg_us"]},
        }
        out.update(extras)
        if samp:
            out["sampling"] = samp
        if modes:
            out["operand_modes"] = modes
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    # the engines (and their HIP graphs with captured RCCL nodes) go before the communicator does
    del drv, eng
    parallel.shutdown(destroy=True)


if __name__ == "__main__":
    main()
