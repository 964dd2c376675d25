"""Peer-to-peer ("mesh") gradient exchange (include/wfdgan.h wfd_mesh_*, csrc/mesh.hip; SURVEY section 8 e).  The test box
has ONE GPU: the ranks are separate processes that share it, so what is checked is the protocol (IPC mapping, flags,
shards, in-place exchange, the fused optimizer step, graph replay) and its bits, not link bandwidth."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world, script, tmp_path, extra_env=None, timeout=240):
    port = free_port()
    out = str(tmp_path / "res.json")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", WFD_DIST_BACKEND="gloo")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, script), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            logs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = []
    for r in range(world):
        f = out + f".{r}"
        got = json.load(open(f)) if os.path.exists(f) else None
        assert procs[r].returncode == 0, f"rank {r} failed: {got}\n{logs[r][-3000:]}"
        res.append(got)
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8])
def test_mesh_exchange_between_processes(world, tmp_path):
    for r in run_ranks(world, "mesh_worker.py", tmp_path):
        assert r["ok"] and r["timeouts"] == 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["mesh", "allreduce"])
@pytest.mark.parametrize("cfg", [(41, 5, 16, 8, "div"), (501, 10, 500, 64, "div"), (501, 10, 500, 64, "gp")])
def test_late_gradient_dp_step_equals_the_fused_single_gpu_step(form, cfg, monkeypatch):
    """The data-parallel step in its "late gradients" shape (whole chain first, all weight gradients + column sums STORED by
    one launch -- wfd_gemm_f32_multi_grad --, then ONE exchange: the mesh launch that sums, applies RMSprop and
    redistributes, or an all-reduce of the whole buffer + the flat pass) at world 1 must reproduce the fused single-GPU
    engine (RMSprop inside the weight-gradient stores) bit for bit: same products, same summation orders, same update."""
    import torch
    from tests.test_gpu_gp_and_cli import _setup, DEV
    from wfd_gan_amd.engine import TrainEngine
    monkeypatch.setenv("WFD_DP_MESH", "1" if form == "mesh" else "0")
    monkeypatch.setenv("WFD_DP_LATE_GRADS", "1")
    S, C, Z, B, penalty = cfg
    st, e0, Xs, y = _setup(S, C, Z, B, penalty)
    mk = lambda **kw: TrainEngine(S, C, Z, B, e0.scaler_min, e0.scaler_max, e0.generator_state(), e0.discriminator_state(),
                                  {k: v for k, v in st.DF.items()}, device=DEV, penalty=penalty, **kw)
    e1, e2 = mk(fuse_opt=True), mk(world_size=1, force_sync=True)
    assert e1.fuse_d and e1.fuse_g and e2.dp and e2.late_grads and e2.mesh_dp == (form == "mesh")
    for e in (e1, e2):
        e.attach_dataset(torch.from_numpy(Xs), torch.from_numpy(y), shuffle_seed=5)
        e.capture_graphs()
    L = e2.L
    fns = [fn for fn, _, _ in e2.fast_d.calls]
    assert L.gemm_f32_multi_grad in fns and (L.mesh_reduce_rmsprop in fns) == (form == "mesh")
    assert len(e2.graph_d) == 1
    for i in range(7):
        e1.iteration(i)
        e2.iteration(i)
    torch.cuda.synchronize()
    assert torch.equal(e1.Dp.flat, e2.Dp.flat) and torch.equal(e1.Gp.flat, e2.Gp.flat)
    assert torch.equal(e1.Dp.sq, e2.Dp.sq) and torch.equal(e1.Gp.sq, e2.Gp.sq)
    assert e1.d_stats() == e2.d_stats()
    if form == "mesh":
        done, timeouts = e2._mesh.status()
        assert timeouts == 0 and done >= 7
