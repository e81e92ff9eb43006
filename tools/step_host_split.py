import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from art_amd.engine import MipsEngine
eng = MipsEngine(768, torch.float16, [0])
bench.build_shard(eng, 0, 2626916, 1234, torch.float16)
q = torch.randn((64, 768), device="cuda").half()
eng.set_stats(False)
for _ in range(50): eng.search(q, 32)
torch.cuda.synchronize()
n = 400
t0 = time.perf_counter()
for _ in range(n): eng.search(q, 32)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e6
i = eng.info()
print("step %.1f us; host enqueue %.1f us, host sync %.1f us -> python+ctypes outside the library %.1f us" % (dt, i["last_host_enqueue_us"], i["last_host_sync_us"], dt - i["last_host_enqueue_us"] - i["last_host_sync_us"]))
