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
@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_mesh_exchange_between_processes(world, tmp_path):
    for r in run_ranks(world, "mesh_worker.py", tmp_path):
        assert r["ok"] and r["timeouts"] == 0, r
