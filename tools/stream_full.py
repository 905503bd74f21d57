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


def frame_source(seed, n, w=synth.W, h=synth.H, d_max=40, z_max=60.0, noise=1.0):
    """synth.synth_translation_sequence, lazily: one baseline to the right per frame."""
    rng = np.random.default_rng(0x5EED1000 + seed)
    wide = synth.synth_texture(rng, w + (n + 1) * d_max + 8, h)
    strips, y = [], 0
    while y < h:
        bh = int(rng.integers(24, 64))
        z = rng.uniform(synth.KITTI["fx"] * synth.KITTI["baseline"] / d_max, z_max)
        d = max(2, min(int(round(synth.KITTI["fx"] * synth.KITTI["baseline"] / z)), d_max))
        strips.append((y, min(h, y + bh), d))
        y += bh

    def get(k):
        left = np.empty((h, w), np.uint8); right = np.empty((h, w), np.uint8)
        for y0, y1, d in strips:
            left[y0:y1] = wide[y0:y1, k * d:k * d + w]
            right[y0:y1] = wide[y0:y1, (k + 1) * d:(k + 1) * d + w]
        r = np.random.default_rng(seed * 100003 + k)
        left = np.clip(np.rint(left + r.normal(0, noise, left.shape)), 0, 255).astype(np.uint8)
        right = np.clip(np.rint(right + r.normal(0, noise, right.shape)), 0, 255).astype(np.uint8)
        return left, right
    return get


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ssk_amd import capi
    get = frame_source(9, n)
    ctx = pkg.Context(0)
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    train = [ctx.frontend_run(*get(k), prm)["desc_left"] for k in range(0, n, max(1, n // 32))][:32]
    with tempfile.TemporaryDirectory() as d:
        voc = os.path.join(d, "voc.yml.gz")
        bv.write_yaml(bv.train(train, k=8, L=4, seed=5), voc)
        slam = capi.Slam(ctx, synth.W, synth.H, "kitti")
        slam.load_vocabulary(voc)
        t_loop, n_kf_frames, t_kf, worst, vo_sum, lc = 0.0, 0, 0.0, 0.0, 0.0, 0
        for k in range(n):
            l, r = get(k)
            t0 = time.perf_counter()
            rep = slam.process_frame(l, r)
            dt = time.perf_counter() - t0
            if k > 0:
                t_loop += dt
                worst = max(worst, dt)
                if rep.inserted_kf:
                    n_kf_frames += 1; t_kf += dt
            vo_sum += np.array(rep.vo_pose[:]).reshape(3, 4)[0, 3] if rep.vo_valid else 0.0
            lc += rep.confirmed_loop_closure
            if rep.inserted_kf and rep.n_created_edges != 1:
                print("frame %d kf %d: %d edges created, similar %s, err %.1f -> %.1f" % (k, rep.kf_id, rep.n_created_edges, list(rep.similar_kfs[:rep.n_similar]), rep.total_sqr_error_init, rep.total_sqr_error_final))
            if not rep.vo_valid:
                print("frame %d: odometry invalid" % k)
        nk, ne, nl = slam.counts()
        t0 = time.perf_counter()
        X, info = slam.optimize_pose_graph()
        t_pg = time.perf_counter() - t0
        print("frames %d keyframes %d edges %d landmarks %d | loop %.2f s = %.1f frames/s (%.2f ms per frame, %.2f ms on keyframe frames, worst %.1f ms) | "
              "pose graph %.3f s (stop %d, %d attempts, cost %.4g -> %.4g) | odometry sum %.2f m, loop closures %d | last keyframe y %.2f m for %.2f m of motion"
              % (n, nk, ne, nl, t_loop, (n - 1) / t_loop, 1e3 * t_loop / (n - 1), 1e3 * t_kf / max(n_kf_frames, 1), 1e3 * worst, t_pg, info.stop_reason, info.iters, info.cost_init, info.cost_final, vo_sum, lc,
                 X[-1].reshape(3, 4)[1, 3], 0.54 * 4 * (nk - 1)))
        slam.close()
    ctx.close()


if __name__ == "__main__":
    main()
