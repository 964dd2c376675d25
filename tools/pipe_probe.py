import sys, time, torch, types
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
args = types.SimpleNamespace(seq_len=1401, classes=100, latent_dim=500, batch=64, inst=1000, penalty="div", overlap=0, dp="weak")
dev = torch.device("cuda:0")
eng, _ = bench.build_engine(args, 0, 1, dev)
from wfd_gan_amd.engine import PipelinedTrainer
for mode in ("pipe", "seq"):
    if mode == "pipe":
        drv = PipelinedTrainer(eng, graphs=True)
    else:
        eng.capture_graphs(); drv = eng
    for i in range(30): drv.iteration(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30, 330): drv.iteration(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(mode, "enqueue ms/iter", (t1 - t0) / 300 * 1e3, "total ms/iter", (t2 - t0) / 300 * 1e3)
