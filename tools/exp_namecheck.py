import os, sys, time, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
os.environ.pop("NCCL_DEBUG", None)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gci_amd import shard
from gci_amd.device import Engine, REC_DTYPE, name_hash_np
e = Engine(0)
n = 136766
r = np.zeros(n, dtype=REC_DTYPE); r["name_hash"] = name_hash_np([b"q%d" % i for i in range(n)]); r["flags"] = 1
recs = e.to_device(r.view(np.uint8).reshape(n, 32))
chk = shard.NameCheck(n, e.device, e.hash_bucket, e.hash_conflicts)
def T(label, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); print("%-28s %8.1f us" % (label, (time.perf_counter() - t) / reps * 1e6))
T("conflicts() total", lambda: chk.conflicts(recs))
T("bucket", lambda: e.hash_bucket(recs, 1, chk.cap, chk.send_h, chk.send_c))
T("a2a hashes", lambda: dist.all_to_all_single(chk.recv_h, chk.send_h))
T("a2a counts", lambda: dist.all_to_all_single(chk.recv_c, chk.send_c))
T("conflict kernel", lambda: e.hash_conflicts(chk.recv_h, chk.recv_c, 1, chk.cap, chk.n_conf))
T("all_reduce", lambda: dist.all_reduce(chk.n_conf))
T("item", lambda: chk.n_conf.item())
dist.destroy_process_group()
