#!/usr/bin/env python3
"""Un-batched latency of the host drop-in path (BASELINE configs[0]/[3] shape): N synthetic KITTI-shaped stereo pairs
streamed one keyframe at a time through the C++ facade (CStereoSLAMKF::create -> data association with the previous
keyframe -> change-in-pose filter -> rba.define_new_keyframe), i.e. every call pays its own launches, PCIe copies and
symbolic LM set-up.  Prints ms per keyframe; the batched throughput number is bench.py's."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402


def full_loop(n):
    """BASELINE configs[3] shape at reduced length: n synthetic frames through the WHOLE frame loop of the facade
    (performStereoSLAM: odometry stand-in, keyframe policy, place recognition, data association, relative BA) with the
    reference's KITTI parameters, un-batched, then the global pose graph.  The vocabulary is trained on the sequence."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    import bow_vocab as bv
    pkg = ge.load_package()
    from ssk_amd import capi
    demo = os.path.join(ROOT, "srba-stereo-slam_amd", "host", "facade_demo")
    frames, _ = synth.synth_translation_sequence(9, n, z_max=30.0)
    ctx = pkg.Context(0)
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    voc = bv.train([ctx.frontend_run(l, r, prm)["desc_left"] for l, r in frames[::max(1, n // 24)]], k=8, L=4, seed=5)
    ctx.close()
    with tempfile.TemporaryDirectory() as d:
        vf = os.path.join(d, "voc.yml.gz")
        bv.write_yaml(voc, vf)
        for k, (l, r) in enumerate(frames):
            l.tofile(os.path.join(d, "left_%04d.raw" % k))
            r.tofile(os.path.join(d, "right_%04d.raw" % k))
        env = dict(os.environ) if os.environ.get("SSK_FACADE_TIMING") or os.environ.get("SSK_TIMING") else None
        p = subprocess.run([demo, "stream", vf, d, str(n), str(synth.W), str(synth.H)], capture_output=True, text=True, env=env)
        assert p.returncode == 0, p.stderr
        for l in p.stdout.splitlines():
            if l.startswith(("timing", "keyframes", "posegraph", "LMT")):
                print(l)
        for l in p.stderr.splitlines()[-8:]:
            if "facade timing" in l or "ssk timing" in l:
                print(l)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "loop":
        return full_loop(int(sys.argv[2]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    demo = os.path.join(ROOT, "srba-stereo-slam_amd", "host", "facade_demo")
    frames, _ = synth.synth_translation_sequence(3, n)
    with tempfile.TemporaryDirectory() as d:
        for k, (l, r) in enumerate(frames):
            l.tofile(os.path.join(d, "left_%04d.raw" % k))
            r.tofile(os.path.join(d, "right_%04d.raw" % k))
        subprocess.run([demo, d, "2", str(synth.W), str(synth.H)], capture_output=True)      # warm-up: runtime + code objects
        t0 = time.perf_counter()
        p1 = subprocess.run([demo, d, "2", str(synth.W), str(synth.H)], capture_output=True, text=True)
        t1 = time.perf_counter()
        p2 = subprocess.run([demo, d, str(n), str(synth.W), str(synth.H)], capture_output=True, text=True, env=dict(os.environ, SSK_FACADE_TIMING="1", **({"SSK_TIMING": "1"} if os.environ.get("SSK_TIMING") else {})))
        t2 = time.perf_counter()
        assert p1.returncode == 0 and p2.returncode == 0, p2.stderr
        per_kf = ((t2 - t1) - (t1 - t0)) / (n - 2)
        print("facade stream: %d keyframes, %.2f ms per keyframe (process start-up subtracted), %.1f keyframes/s un-batched"
              % (n, 1e3 * per_kf, 1.0 / per_kf))
        print(p2.stdout.splitlines()[-1])
        for l in p2.stderr.splitlines()[-6:]:
            print(l)


if __name__ == "__main__":
    main()
