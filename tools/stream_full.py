#!/usr/bin/env python3
"""BASELINE configs[3] at full length: N synthetic KITTI-shaped stereo frames (default 4541 = KITTI-00) through the whole
frame loop of the estimator (C-ABI include/ssk_slam.h: odometry stand-in, keyframe policy, place recognition, data
association, relative BA, then the global pose graph), un-batched, frames generated on the fly.  Prints frames/s of the loop
(generation excluded) and the final drift."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import bow_vocab as bv  # noqa: E402
import synth            # noqa: E402


def run(n, n_back, seed=9, verbose=True, voc_every=None, **options):
    """The run of tests/test_config3_gpu.py and of bench.py's `unbatched` object: n frames (the last n_back of them back over the
    places just seen) through ssk_slam_process_frame, then the global pose graph.  Returns a dict of what was observed."""
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ssk_amd import capi
    pos = synth.out_and_back_positions(n, n_back)
    get = synth.translation_frame_source(seed, max(pos))
    ctx = pkg.Context(0)
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    step = voc_every or max(1, n // 32)
    train = [ctx.frontend_run(*get(pos[k], k), prm)["desc_left"] for k in range(0, n, step)][:32]
    out = dict(frames=n, n_back=n_back, invalid_odometry=[], loop_closures=[], multi_edge_kfs=0, kf_frames=[], kf_pos=[], error_increased=[], stage2_edges=[], stage2_obs=[])
    with tempfile.TemporaryDirectory() as d:
        voc = os.path.join(d, "voc.yml.gz")
        bv.write_yaml(bv.train(train, k=8, L=4, seed=5), voc)
        slam = capi.Slam(ctx, synth.W, synth.H, "kitti", **options)
        slam.load_vocabulary(voc)
        t_loop, t_kf, worst = 0.0, 0.0, 0.0
        for k in range(n):
            l, r = get(pos[k], k)
            t0 = time.perf_counter()
            rep = slam.process_frame(l, r)
            dt = time.perf_counter() - t0
            if k > 0:
                t_loop += dt
                worst = max(worst, dt)
                if rep.inserted_kf:
                    t_kf += dt
            if rep.inserted_kf:
                out["kf_frames"].append(k); out["kf_pos"].append(pos[k])
                if k > 0:                                   # the local area define_new_keyframe optimised (stage 2): its size per keyframe
                    out["stage2_edges"].append(int(rep.lm_stage2_edges)); out["stage2_obs"].append(int(rep.lm_stage2_obs))
                if rep.n_created_edges > 1:
                    out["multi_edge_kfs"] += 1
                if k > 0 and not rep.total_sqr_error_final <= rep.total_sqr_error_init:
                    out["error_increased"].append(k)
            if rep.confirmed_loop_closure:
                out["loop_closures"].append((k, int(rep.lc_kf_id), bool(rep.inserted_kf)))
                if verbose:
                    print("frame %d: loop closure with keyframe %d (inserted %d, %d edges, err %.1f -> %.1f)" % (k, rep.lc_kf_id, rep.inserted_kf, rep.n_created_edges, rep.total_sqr_error_init, rep.total_sqr_error_final))
            if not rep.vo_valid:
                out["invalid_odometry"].append(k)
                if verbose:
                    print("frame %d: odometry invalid" % k)
        nk, ne, nl = slam.counts()
        tm = slam.timing()
        t0 = time.perf_counter()
        X, info = slam.optimize_pose_graph()
        out.update(keyframes=nk, edges=ne, landmarks=nl, loop_s=t_loop, frames_per_s=(n - 1) / t_loop, ms_per_frame=1e3 * t_loop / (n - 1),
                   ms_per_keyframe_frame=1e3 * t_kf / max(len(out["kf_frames"]) - 1, 1), worst_ms=1e3 * worst, posegraph_s=time.perf_counter() - t0,
                   pg_stop=info.stop_reason, pg_iters=info.iters, pg_cost_init=info.cost_init, pg_cost_final=info.cost_final, poses=X,
                   timing={f: getattr(tm, f) for f, _ in tm._fields_})
        slam.close()
    ctx.close()
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
    n_back = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    opts = dict(kv.split("=") for kv in sys.argv[3:])
    o = run(n, n_back, **{k: float(v) for k, v in opts.items()})
    X, kp = o["poses"], np.array(o["kf_pos"])
    # robot frame: the camera's right is -y (camera_pose_on_robot, EST.cpp:1105-1112); keyframe k sits 0.54 * kf_pos[k] m to the right
    err = np.abs(X.reshape(-1, 3, 4)[:, 1, 3] + 0.54 * kp)
    Xr = X.reshape(-1, 3, 4)
    print("max |x| %.3f max |z| %.3f max rotation deviation %.4f" % (np.abs(Xr[:, 0, 3]).max(), np.abs(Xr[:, 2, 3]).max(), np.abs(Xr[:, :, :3] - np.eye(3)).max()))
    path = 0.54 * (n - 1)
    print("frames %d keyframes %d edges %d landmarks %d | loop %.2f s = %.1f frames/s (%.2f ms per frame, %.2f ms on keyframe frames, worst %.1f ms) | "
          "pose graph %.3f s (stop %d, %d attempts, cost %.4g -> %.4g) | loop closures %d, keyframes with > 1 new edge %d, invalid odometry %d | "
          "position error max %.3f m = %.3f %% of the %.0f m path, last keyframe %.3f m"
          % (n, o["keyframes"], o["edges"], o["landmarks"], o["loop_s"], o["frames_per_s"], o["ms_per_frame"], o["ms_per_keyframe_frame"], o["worst_ms"],
             o["posegraph_s"], o["pg_stop"], o["pg_iters"], o["pg_cost_init"], o["pg_cost_final"], len(o["loop_closures"]), o["multi_edge_kfs"],
             len(o["invalid_odometry"]), err.max(), 100 * err.max() / path, path, err[-1]))
    print("timing", o["timing"])
    # round 5: the distribution of the stage-2 windows the stream produces: percentiles + histogram -> JSON when SSK_STREAM_HIST names a file
    e, ob = np.array(o["stage2_edges"]), np.array(o["stage2_obs"])
    if len(e):
        q = lambda a: {"p%d" % p: float(np.percentile(a, p)) for p in (10, 50, 90, 99)} | {"max": int(a.max()), "mean": float(a.mean())}
        hist = {"frames": n, "keyframes": int(o["keyframes"]), "options": opts, "stage2_optimised_edges": q(e), "stage2_observations": q(ob),
                "edges_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(e, return_counts=True))},
                "observations_histogram_per_2000": {str(int(k) * 2000): int(v) for k, v in zip(*np.unique(ob // 2000, return_counts=True))}}
        print("stage-2 windows:", hist["stage2_optimised_edges"], hist["stage2_observations"])
        if os.environ.get("SSK_STREAM_HIST"):
            import json
            json.dump(hist, open(os.environ["SSK_STREAM_HIST"], "w"), indent=1)


if __name__ == "__main__":
    main()
