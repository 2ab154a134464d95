#!/usr/bin/env python3
"""Is the bench step host-bound on the multi-GPU path?  Host enqueue time per step vs wall time per step, with the
exchange forced on one GPU (world 1), and host time of each call inside step()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from gci_amd.device import Engine
eng = Engine(0)
for exch in (False, True):
    w = bench.Workload(eng, 0, 1, bench.CHR19_LEN, 40.0, exchange=exch)
    for _ in range(5):
        w.step()
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        w.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("exchange=%s: host enqueue %.1f us/step, wall %.1f us/step" % (exch, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
    # host-only cost: enqueue while the GPU is kept busy elsewhere is not separable; instead sync before each step
    t = 0.0
    for _ in range(50):
        torch.cuda.synchronize()
        a = time.perf_counter(); w.step(); t += time.perf_counter() - a
    print("   host time of one step() on an idle GPU: %.1f us" % (t / 50 * 1e6), flush=True)
    if exch:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200):
            w.step()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
dist.destroy_process_group()
