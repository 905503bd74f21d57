import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["SSK_BENCH_EXCHANGE_ALWAYS"]="1"
import torch, numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
rank, world = pkg.dist.init_from_env("nccl", always=True)
dev=torch.device("cuda",0)
S=256
shared=torch.zeros(S*12,dtype=torch.float64,device=dev); gathered=torch.zeros(S*12,dtype=torch.float64,device=dev)
ctx=pkg.Context(0)
x=torch.zeros(64*1024*1024,device=dev)
def busy():
    with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream(),device=0)):
        for _ in range(20): x.mul_(1.0001)
for rep in range(3):
    torch.cuda.synchronize()
    busy()
    t0=time.perf_counter()
    pkg.dist.exchange_shared(shared, gathered, stream=ctx.stream())
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print("all_gather host call %.3f ms (with ~%.1f ms of queued GPU work in front), sync after %.3f ms" % (1e3*(t1-t0), 1e3*(t2-t0), 1e3*(t2-t1)))
