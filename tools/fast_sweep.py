"""One-off measurement: fast_nms_kernel time vs FAST threshold (how much of it is the always-on scan)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = pkg.Context(0, stream=torch.cuda.current_stream().cuda_stream)
cur = capi.KfBatch(ctx, B, synth.W, synth.H, 512)
for i in range(8):
    l, r = synth.synth_stereo_pair(i)
    cur.upload(i, l, r)
ctx.synchronize()
ic = cur.image_tensor()
for k in range(8, B):
    ic[k, :, :, :synth.W] = torch.roll(ic[k % 8, :, :, :synth.W], shifts=5 * (k // 8), dims=-1)
torch.cuda.synchronize()
for th in (20, 40, 80, 254):
    d = dict(synth.KITTI_FRONTEND); d["fast_threshold"] = th
    fp = capi.FrontendParams(**d)
    for _ in range(2): cur.run_frontend(fp)
    ctx.profile_enable(True)
    for _ in range(5): cur.run_frontend(fp)
    p = ctx.profile_read(); ctx.profile_enable(False)
    lib = ctx.lib
    import ctypes as C
    print("th", th, {k: round(v[0] / v[1], 4) for k, v in p.items() if v[1]}, "GB/s", round(B * 2 * synth.W * synth.H / (p["fast_nms"][0] / p["fast_nms"][1] * 1e-3) / 1e9, 1))
