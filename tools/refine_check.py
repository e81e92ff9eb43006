import os, sys, torch
R=os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0,R)
os.environ["ARTMIPS_LIBRARY"]=R+"/art_amd/lib/libartmips_diag.so"
import bench
from art_amd.engine import MipsEngine
eng=MipsEngine(768, torch.float16,[0]); bench.build_shard(eng,0,int(sys.argv[1]),1234,torch.float16)
q=torch.randn((2048,768),device="cuda").half()
for r in ("0","1"):
    os.environ["ARTMIPS_REFINE"]=r
    for _ in range(2): eng.search(q,1000)
    i=eng.info(); print("refine",r,"cand/query",i["last_candidates"]/2048,"scan ms",i["last_scan_ms"],"total",i["last_total_ms"])
