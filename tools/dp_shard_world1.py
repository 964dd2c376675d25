#!/usr/bin/env python3
"""The sharded data-parallel optimizer (WFD_DP_SHARD=1, parallel.ShardedSync) on the real "nccl" (RCCL) backend with a
1-rank group, outside the pytest suite: runs the test bodies of tests/test_gpu_gp_and_cli.py with shard=True -- collectives
captured inside the step graphs, eager between graph stages, and the capture-failure fallback -- each in a process of its
own, and prints one line per case.  usage: python tools/dp_shard_world1.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "captured": "T._data_parallel_world1_nccl(True, MP(), True)",
    "eager": "T._data_parallel_world1_nccl(False, MP(), True)",
    "fallback": "T.test_capture_fallback_drops_stale_collective_handles.__wrapped__(True, MP(), None) if hasattr("
                "T.test_capture_fallback_drops_stale_collective_handles, '__wrapped__') else "
                "T.test_capture_fallback_drops_stale_collective_handles(True, MP(), None)",
}
BODY = """
import os, sys, gc, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WFD_DP_SHARD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1)
import pytest
from tests import test_gpu_gp_and_cli as T
MP = pytest.MonkeyPatch
mp = None
{call}
gc.collect(); torch.cuda.synchronize(); dist.destroy_process_group()
print("CASE OK")
"""

if __name__ == "__main__":
    bad = 0
    for name, call in CASES.items():
        p = subprocess.run([sys.executable, "-c", BODY.format(root=ROOT, call=call)], cwd=ROOT, capture_output=True, text=True,
                           timeout=280)
        ok = p.returncode == 0 and "CASE OK" in p.stdout
        bad += not ok
        print(f"sharded world-1 RCCL {name}: {'ok' if ok else 'FAILED rc=%d' % p.returncode}")
        if not ok:
            print(p.stderr[-3000:])
    sys.exit(1 if bad else 0)
