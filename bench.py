#!/usr/bin/env python3
"""bench.py -- keyframes/sec (detect + match + LM solve) of the MI355X-native srba-stereo-slam hot path.

One "step" = one pass of the whole per-keyframe hot path over a batch of B synthetic KITTI-shaped keyframes that
is already resident in HBM when the timed region starts:
    stereo front-end (FAST+NMS scan, selection, orientation+BRIEF, L<->R match)          [seam B1]
  + keyframe<->previous-keyframe data association (Hamming K=1 + direction/ORB filters
    + fundamental-matrix RANSAC + change-in-pose filters)                               [seam B2]
  + one relative-BA LM solve per keyframe on a BASELINE configs[2]-shaped window         [seam B3]
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), keyframe streams are partitioned one
batch per GPU (weak scaling) and each step ends with ONE all-gather of the shared-landmark blocks.

Prints ONE JSON line (rank 0).  `roofline` is for the kernel that dominates the step (the LM solve), `roofline_scan`
for the front-end's HBM-bound image scan (fast_nms_kernel); both measured live with HIP events on the stream the
kernels run on; `cpu_baseline` is the CPU oracle (the port, not the
reference: the reference's path needs OpenCV/MRPT/SRBA/stereo-vo, absent here) timed on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1241, 376
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # AMD MI355X data sheet: FP64 vector = FP64 matrix (MFMA) 78.6 TFLOP/s (the guide lists no FP64 row)


def lm_algorithmic_flops(win):
    """SURVEY 8(d) per LM iteration of one window: ~1.0 kflop per (observation x edge on its path) for residual, Jacobians
    and Hessian blocks, 324 flop per landmark x (edges touching it)^2 for the Schur complement, (6E)^3 / 3 for the
    factorisation of the reduced system."""
    off = np.asarray(win.obs_path_off)
    path_len = np.diff(off)
    unknown = np.asarray(win.edge_unknown_slot) >= 0
    pe = np.asarray(win.path_edge)
    lm = np.asarray(win.obs_lm)
    touch = {}
    for o in range(len(lm)):
        es = pe[off[o]:off[o + 1]] >> 1                                   # path entries are edge * 2 + direction
        touch.setdefault(int(lm[o]), set()).update(int(e) for e in es if unknown[e])
    schur = sum(len(v) ** 2 for v in touch.values()) * 324.0
    n6 = 6.0 * int(unknown.sum())
    return 1000.0 * float(path_len.sum()) + schur + n6 ** 3 / 3.0



def _synth_pair(seed):
    import synth
    return synth.synth_stereo_pair(seed)


def build_workload(pkg, capi, ctx, ctx_lm, B, rank, n_base=64, n_lm_distinct=64, shared=None):
    """B keyframes + B 'previous' keyframes resident in HBM, B LM windows uploaded."""
    from multiprocessing.pool import ThreadPool
    import torch
    import synth_lm as sl
    cur = capi.KfBatch(ctx, B, W, H, 512)
    prev = capi.KfBatch(ctx, B, W, H, 512)
    n_base = min(n_base, B)
    # the distinct base pairs are rendered on the host cores side by side (~2 s each on one core; numpy / scipy release the interpreter
    # lock).  THREADS, not processes: under rocprofv3 the process has initialised the GPU before main() runs, and a fork / exec from
    # such a process is what the GPU boxes of this pool refuse (a spawn pool hung a profiled run for 40 minutes)
    with ThreadPool(max(1, min(16, os.cpu_count() or 1, n_base))) as pool:
        pairs_lr = pool.map(_synth_pair, [1000 * rank + i for i in range(n_base)])
    for i, (l, r) in enumerate(pairs_lr):
        cur.upload(i, l, r)
    ctx.synchronize()
    ic, ip = cur.image_tensor(), prev.image_tensor()
    for k in range(n_base, B):      # distinct content per keyframe: circular shifts of the base pairs
        ic[k, :, :, :W] = torch.roll(ic[k % n_base, :, :, :W], shifts=5 * (k // n_base), dims=-1)
    # previous keyframe: the same scene seen from a camera displaced sideways -- horizontal image motion that differs from
    # band to band (2..6 px, as strips at different depths), so the pair has a proper epipolar geometry (a uniform image
    # shift is a degenerate configuration for the fundamental-matrix stage: every 7-point sample is singular)
    for b0 in range(0, H, 47):
        ip[:, :, b0:b0 + 47, :W] = torch.roll(ic[:, :, b0:b0 + 47, :W], shifts=-(2 + (b0 // 47) % 5), dims=-1)
    torch.cuda.synchronize()
    wins = [sl.make_submap_problem(100 * rank + s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2,
                                   obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(n_lm_distinct)]
    batch = [wins[k % n_lm_distinct] for k in range(B)]
    shared_idx = None
    if shared is not None:
        # BASELINE configs[4]: window 0 of every rank is ITS submap of the shared set (8 config-3-shaped submaps, seeds 50.., + 256 landmarks
        # of one identity each, observed from two of the submaps): the window the exchange packs; the rest of the batch as before
        n_sub, n_shared = shared
        sw, si, _ = sl.make_shared_submaps(n_sub=n_sub, n_shared=n_shared, seed0=50, only={rank % n_sub}, n_kf=200, n_lm=5000, submap_size=10,
                                           max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))
        batch[0] = sw[0]
        shared_idx = si[0]
    lm = capi.LmBatch(ctx_lm, batch)
    return cur, prev, lm, wins, shared_idx, batch


BOW_DB_ENTRIES = 200       # place-recognition database: the 200 keyframes of BASELINE configs[2]'s submap


def build_place_recognition(capi, ctx, prev):
    """Vocabulary of the reference's shape (demo/voc.yml.gz: k = 8, L = 5; here a seeded synthetic complete tree, 37 448
    nodes / 32 768 words) and a database holding the bag-of-words vectors of BOW_DB_ENTRIES earlier keyframes."""
    import bow_vocab as bv
    v = bv.synth(k=8, L=5, seed=2024)
    voc = capi.BowVocab(ctx, v)
    db = capi.BowDb(ctx, voc, BOW_DB_ENTRIES, 512)
    prev.bow_transform(voc)
    for e in range(BOW_DB_ENTRIES):
        db.add_from_batch(prev, e % prev.n_kf)
    ctx.synchronize()
    return v, voc, db


def params(capi, sl):
    import ctypes as C
    import synth
    fp = capi.FrontendParams(**synth.KITTI_FRONTEND)
    da = capi.DaParams(1, 1, H, 60.0)
    lp = capi.LmParams()
    C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
    cp = capi.CipParams(**synth.KITTI_CIP)
    fm = capi.FmParams(**synth.KITTI_FM)
    return fp, da, lp, cp, fm


def cpu_baseline(sample_kf, wins, vocab):
    """The oracle (CPU port, 1 thread, -O3 -march=native -ffp-contract=off: the reference builds -O3 -mtune=native, CMakeLists.txt:98-100) on `sample_kf` keyframes of the same
    workload: front-end + DA vs a shifted previous keyframe + place-recognition query + LM solve."""
    import oracle_lib as ol
    import synth
    import synth_lm as sl
    out_dir = os.path.join("/tmp", "ssk_oracle_native_%d" % os.getpid())
    try:
        oracle = ol.load(march="native", out=out_dir, opt="-O3")
    except Exception:
        oracle = ol.load()
    fp = ol.FrontendParams(**synth.KITTI_FRONTEND)
    da = ol.DaParams(1, 1, H, 60.0)
    lp = sl.make_params()
    cipp = ol.kitti_cip_params()
    ident = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(12)
    pat = ol.default_pattern()
    base = [synth.synth_stereo_pair(i) for i in range(4)]
    def banded(img):
        out = img.copy()
        for b0 in range(0, H, 47):
            out[b0:b0 + 47] = np.roll(img[b0:b0 + 47], -(2 + (b0 // 47) % 5), axis=1)
        return out
    prev_kf = [oracle.frontend(banded(l), banded(r), fp, pat) for l, r in base]
    fmp = ol.kitti_fm_params()
    ovoc = oracle.bow_vocab(vocab)
    odb = oracle.bow_db(ovoc.n_words)
    for e in range(BOW_DB_ENTRIES):
        odb.add(*ovoc.transform(prev_kf[e % len(prev_kf)]["desc_left"]))
    t0 = time.perf_counter()
    t_fe = t_da = t_lm = t_bow = 0.0
    for k in range(sample_kf):
        l, r = base[k % len(base)]
        a = time.perf_counter()
        kf = oracle.frontend(l, r, fp, pat)
        b = time.perf_counter()
        va, _ka = ol.make_kf_view(**kf)
        vb, _kb = ol.make_kf_view(**prev_kf[k % len(base)])
        _m, st, _n = oracle.data_association(va, vb, da)

        def stereo_obs(k):
            m = k["matches"]
            return np.stack([k["kp_left"]["x"][m["queryIdx"]], k["kp_left"]["y"][m["queryIdx"]],
                             k["kp_right"]["x"][m["trainIdx"]], k["kp_right"]["y"][m["trainIdx"]]], 1)
        so_a, so_b = stereo_obs(kf), stereo_obs(prev_kf[k % len(base)])
        st, _r = oracle.fundamental_filter(so_a[:, :2], so_b[:, :2], _m["trainIdx"], st, fmp)
        oracle.change_in_pose(so_a, so_b, _m["trainIdx"], st, ident, cipp)
        c0 = time.perf_counter()
        odb.query(*ovoc.transform(kf["desc_left"]), 4)
        c = time.perf_counter()
        oracle.lm_solve(wins[k % len(wins)], lp, trace=False)
        d = time.perf_counter()
        t_fe += b - a; t_da += c0 - b; t_bow += c - c0; t_lm += d - c
    dt = time.perf_counter() - t0
    return {"value": sample_kf / dt, "unit": "keyframes/s", "cores": 1, "kind": "port",
            "sample": "%d keyframes of the same workload (CPU oracle, 1 thread): front-end %.1f ms + DA %.2f ms + place recognition %.2f ms + LM %.1f ms per keyframe"
                      % (sample_kf, 1e3 * t_fe / sample_kf, 1e3 * t_da / sample_kf, 1e3 * t_bow / sample_kf, 1e3 * t_lm / sample_kf),
            "host_cores": os.cpu_count()}


def cpu_worker(n_kf):
    """--cpu-worker: one process of the all-cores CPU baseline; builds its own copy of the workload (2 LM windows, the same
    vocabulary) and prints {"kf": n, "seconds": t} for its timed section."""
    import bow_vocab as bv
    import synth_lm as sl
    wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2,
                                   obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(2)]
    r = cpu_baseline(n_kf, wins, bv.synth(k=8, L=5, seed=2024))
    print(json.dumps({"kf": n_kf, "seconds": n_kf / r["value"]}))


def cpu_baseline_all_cores(procs, n_kf):
    """The same CPU oracle, one single-threaded process per keyframe stream on `procs` host cores at once (SURVEY 8(d):
    "an N-thread run, one KF/submap per thread, for fairness").  Aggregate = sum of the processes' own rates; their timed
    sections (>= several seconds each) run concurrently."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(n_kf)], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(procs)]
    rate, done = 0.0, 0
    for p in ps:
        out, _ = p.communicate()
        try:
            r = json.loads(out.strip().splitlines()[-1])
            rate += r["kf"] / r["seconds"]
            done += 1
        except Exception:
            pass
    if done == 0:
        return None
    return {"value": rate, "unit": "keyframes/s", "cores": done, "kind": "port",
            "sample": "%d single-threaded oracle processes side by side, %d keyframes each (same per-keyframe work as cpu_baseline)" % (done, n_kf),
            "host_cores": os.cpu_count()}


def cpu_quota():
    """CPUs the container may use (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="keyframes per GPU per step (256: 67.3 k, 512: 70.1 k, 1024: 72.7 k keyframes/s on one MI355X)")
    ap.add_argument("--cpu-sample", type=int, default=128, help="keyframes timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the all-cores CPU baseline (-1: half the host's logical cores, at most 64; 0 = skip)")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--side-stream", type=int, default=1,
                    help="2: behind the front-end the data-association chain, place recognition and the LM batch on three HIP streams side by side; "
                         "1: the place-recognition kernels on a second HIP stream beside the data-association chain (both depend only on the "
                         "front-end; round 6, the default: 130.7 -> 135.6 k keyframes/s -- the small kernels of both chains are latency bound at one or two "
                         "wavefronts per SIMD); 0: one stream.  2 (the LM batch beside both) measured 124 k: the small kernels take wave slots and LDS "
                         "in front of the LM workgroups; consecutive batches overlapped (the chains of batch k beside the front-end of batch k + 1, two "
                         "resident keyframe batches) measured 125 k: kernels of different queues share the CUs badly when both sides want them")
    ap.add_argument("--shared-landmarks", type=int, default=256)
    ap.add_argument("--pcie-steps", type=int, default=5,
                    help="steps of the PCIe-inclusive pass at N=1 (images uploaded from pinned host memory every step; reported as "
                         "`pcie_inclusive`, never as `value`; 0 = skip)")
    ap.add_argument("--lm-distinct", type=int, default=64,
                    help="distinct LM windows in the batch (the rest are replicas): 64 submaps (seeds 0..63, x16 replicas at --batch 1024) give ragged "
                         "iteration counts; rounds 1-2 quoted 8")
    ap.add_argument("--base-images", type=int, default=64, help="distinct rendered stereo pairs (the rest of the batch: circular shifts of them)")
    ap.add_argument("--lm-depth5", type=int, default=1024,
                    help="windows of the `lm_depth5` pass at N=1: the reference's real per-keyframe LM load (KITTI.ini:104-106: optimise depth 5 => a "
                         "local area of ~40 edges) as stage 1 + stage 2 per keyframe, batched and one keyframe at a time; 0 = skip")
    ap.add_argument("--unbatched-frames", type=int, default=401,
                    help="frames of the un-batched pass at N=1 (the estimator's frame loop, one ssk_slam_process_frame per stereo pair: what a "
                         "user of performStereoSLAM sees; reported as `unbatched`, never as `value`; 0 = skip)")
    ap.add_argument("--e2e-streams", type=int, default=1024, help="streams of the `e2e_stage2` pass at N=1 (lock-step keyframe streams with stage 1 + stage 2 per keyframe; 0 = skip)")
    ap.add_argument("--e2e-groups", type=int, default=4, help="groups the e2e_stage2 streams are dealt out over (a host thread and a HIP stream each; 1 = serial)")
    ap.add_argument("--e2e-depth5-streams", type=int, default=128, help="streams of its optimise-depth-5 figure (0 = skip)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: LM solves on a second HIP stream beside the front-end / DA kernels (measured: +1 %, the LM workgroups fill every CU)")
    args = ap.parse_args()
    if args.cpu_worker > 0:
        cpu_worker(args.cpu_worker)
        return

    import torch
    import torch.distributed as dist
    torch.set_num_threads(4)                               # (no CPU tensor work to speak of: idle intra-op threads spinning after an op would eat into the box's CPU quota beside the LM builder's threads)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    # SSK_BENCH_EXCHANGE_ALWAYS=1: run the shared-landmark exchange (pack, RCCL all-gather, stitch) on a single rank too
    exchange = os.environ.get("SSK_BENCH_EXCHANGE_ALWAYS") == "1"
    rank, world = pkg.dist.init_from_env("nccl", always=exchange)       # one process per GPU, RCCL
    exchange = exchange or world > 1
    if args.gpus != world:
        # the driver launches N > 1 through torch.distributed.run with the same --gpus N: a mismatch is a mis-launch
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d (launch N > 1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`)" % (args.gpus, world))
    from ssk_amd import capi
    import synth_lm as sl

    stream = torch.cuda.current_stream().cuda_stream
    ctx = pkg.Context(local, stream=stream)
    # the stream the context really enqueues on (a null handle above makes it create its own non-blocking stream)
    main_stream = torch.cuda.ExternalStream(ctx.stream(), device=local)
    # the LM solves of a batch are independent of the same step's front-end / data association (in a keyframe stream:
    # LM of batch k runs beside the front-end of batch k+1), so they go to a second HIP stream: per-window iteration
    # counts differ, and CUs whose window has converged pick up front-end workgroups instead of idling
    lm_stream = torch.cuda.Stream() if args.overlap else torch.cuda.current_stream()
    ctx_lm = pkg.Context(local, stream=lm_stream.cuda_stream) if args.overlap else ctx
    lm_ext_stream = torch.cuda.ExternalStream(ctx_lm.stream(), device=local)      # the stream the LM context really enqueues on
    B = args.batch
    S = args.shared_landmarks
    cur, prev, lm, wins, shared_idx, lm_batch_wins = build_workload(pkg, capi, ctx, ctx_lm, B, rank, n_base=max(1, args.base_images), n_lm_distinct=max(1, min(args.lm_distinct, B)),
                                                                    shared=(max(world, 2), S) if exchange else None)
    fp, da, lp, cp, fm = params(capi, sl)
    pairs = np.stack([np.arange(B), np.arange(B)], 1).astype(np.int32)
    dev = torch.device("cuda", local)
    N = 512
    d_idx = torch.empty(B * N, dtype=torch.int32, device=dev)
    d_dist = torch.empty(B * N, dtype=torch.int32, device=dev)
    d_st = torch.empty(B * N, dtype=torch.int32, device=dev)
    d_tr = torch.empty(B, dtype=torch.int32, device=dev)
    shared = torch.zeros(S * 12, dtype=torch.float64, device=dev)
    gathered = torch.zeros(world * S * 12, dtype=torch.float64, device=dev) if exchange else None
    if exchange:
        # the consensus positions of the shared landmarks (identical on every rank: the same seeded set) and, per shared landmark, its
        # index in this rank's window 0 (-1: this submap does not observe it)
        _, _, sh = sl.make_shared_submaps(n_sub=max(world, 2), n_shared=S, seed0=50, only=set())
        shared_pos = torch.from_numpy(sh["init"].reshape(-1).copy()).to(dev)
        d_shared_idx = torch.from_numpy(shared_idx.astype(np.int32)).to(dev)

    prev.run_frontend(fp)          # state: the previous keyframes' features (not part of a step)
    ctx.synchronize()
    vocab, voc, db = build_place_recognition(capi, ctx, prev)
    side = torch.cuda.Stream() if args.side_stream else None
    ctx_side = pkg.Context(local, stream=side.cuda_stream) if side else None
    side2 = torch.cuda.Stream() if args.side_stream >= 2 else None
    ctx_side2 = pkg.Context(local, stream=side2.cuda_stream) if side2 else None

    exch_events = []                 # (start, end) HIP events around the exchange of a step, on the LM stream

    def step(serial=False, cur=cur, lm=lm, lm_ctx=None):
        if side2 and not serial and not exchange and lm_ctx is None:
            # (round 6) behind the front-end three things that do not depend on each other run side by side: the data-association chain (matcher,
            # direction / distance filters, RANSAC, change in pose) on one stream, place recognition on another, the LM batch on the main one.  The
            # small kernels are latency bound at one or two wavefronts per SIMD (the RANSAC kernel takes 0.26 ms for 1 024 pairs and for 4 096)
            # and are queued first: they sit in the gaps of the LM workgroups instead of in front of them
            cur.run_frontend(fp)
            side.wait_stream(main_stream); side2.wait_stream(main_stream)
            ctx_side.da_run_batch(cur, prev, pairs, da, d_idx, d_dist, d_st, d_tr)
            ctx_side.fm_run_batch(cur, prev, pairs, fm, d_idx, d_st, d_tr)
            ctx_side.cip_run_batch(cur, prev, pairs, cp, d_idx, d_st, d_tr)
            cur.bow_transform(voc, ctx=ctx_side2)
            cur.bow_query(db, 4, ctx=ctx_side2)
            lm.reset()
            lm.run(lp)
            main_stream.wait_stream(side); main_stream.wait_stream(side2)
            return
        cur.run_frontend(fp)
        if side:
            side.wait_stream(main_stream)
            cur.bow_transform(voc, ctx=ctx_side)
            cur.bow_query(db, 4, ctx=ctx_side)
        ctx.da_run_batch(cur, prev, pairs, da, d_idx, d_dist, d_st, d_tr)
        ctx.fm_run_batch(cur, prev, pairs, fm, d_idx, d_st, d_tr)            # stage 2 of the DA (da_stage2_method = 3): fundamental-matrix
        ctx.cip_run_batch(cur, prev, pairs, cp, d_idx, d_st, d_tr)           # RANSAC filter, then the change-in-pose filter
        if side:
            main_stream.wait_stream(side)
        else:
            cur.bow_transform(voc)                                           # place recognition (queryDB, EST.cpp:426): words of the left
            cur.bow_query(db, 4)                                             # descriptors, L1 query of the keyframe database, 4 results
        if serial:
            torch.cuda.synchronize()      # profile pass: one kernel at a time, clean per-kernel durations
        lm.reset(ctx=lm_ctx)
        lm.run(lp, ctx=lm_ctx)
        if exchange:
            # the one exchange step of the path (BASELINE configs[4]): every submap's g / H blocks of the shared landmarks at its solved
            # state to every GPU, summed in rank order (deterministic), Gauss-Newton step of the shared landmarks, written back: one round of
            # the block-coordinate descent on the joint problem; enqueued behind the solve on its stream (RCCL orders the all-gather against
            # the stream handed over)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(lm_ext_stream)
            lm.pack_shared(0, d_shared_idx, lp, shared)
            pkg.dist.exchange_shared(shared, gathered, stream=ctx_lm.stream())
            ctx_lm.stitch_shared(gathered, world, S, shared_pos)
            lm.set_shared(0, d_shared_idx, shared_pos)
            ev[1].record(lm_ext_stream)
            exch_events.append(ev)
        if serial:
            torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local])
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    exch_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    # the exchange of the timed steps on this rank (pack + RCCL all-gather + stitch + write-back, HIP events on the LM stream): every rank's mean
    # goes into the line, so that the first multi-GPU run is diagnosable
    exchange_ms = float(np.mean([a.elapsed_time(b) for a, b in exch_events])) if exch_events else None
    exchange_ms_all = [exchange_ms]
    if world > 1:
        exchange_ms_all = [None] * world
        dist.all_gather_object(exchange_ms_all, exchange_ms)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline pass: per-stage kernel time with HIP events on the same stream (all ranks run it, rank 0 reports)
    sync_all()
    ctx.profile_enable(True)
    extra_ctx = [c for c in (ctx_lm, ctx_side, ctx_side2) if c is not None and c is not ctx]
    for c in extra_ctx:
        c.profile_enable(True)
    prof_steps = max(3, min(args.steps, 10))
    for _ in range(prof_steps):
        step(serial=True)
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    for c in extra_ctx:
        for k, v in c.profile_read().items():
            prof[k] = (prof[k][0] + v[0], prof[k][1] + v[1]) if k in prof else v
        c.profile_enable(False)
    # ---- configs[4]: how many exchange rounds the joint problem needs from here (after the timed steps; N > 1 or SSK_BENCH_EXCHANGE_ALWAYS):
    # Anderson-mixed rounds (pkg.dist.SharedLandmarkAccelerator: the same host arithmetic on every rank) until the consensus positions
    # move by less than 1e-6 m; plain rounds converge linearly (factor ~0.7) and need about three times as many
    exchange_rounds = None
    if exchange:
        acc = pkg.dist.SharedLandmarkAccelerator(m=5)
        for rnd in range(40):
            lm.run(lp)
            lm.pack_shared(0, d_shared_idx, lp, shared)
            pkg.dist.exchange_shared(shared, gathered, stream=ctx_lm.stream())
            x = shared_pos.cpu().numpy().copy()
            ctx_lm.stitch_shared(gathered, world, S, shared_pos)
            ctx_lm.synchronize()
            nxt = acc.step(x, shared_pos.cpu().numpy())
            shared_pos.copy_(torch.from_numpy(nxt))
            lm.set_shared(0, d_shared_idx, shared_pos)
            # every rank must take the same decision (the host arithmetic could differ by an ulp between hosts: one rank leaving the loop
            # while the others enter the next all-gather would hang the collective)
            stop = torch.tensor([1 if float(np.abs(nxt - x).max()) < 1e-6 else 0], dtype=torch.int32, device=dev)
            if world > 1:
                dist.all_reduce(stop, op=dist.ReduceOp.MIN)
            if int(stop.item()):
                exchange_rounds = rnd + 1
                break
        sync_all()
        lm.reset()
        lm.run(lp)
    res = lm.results()
    # a number from a run whose solves did not end properly is not a number: every window must have stopped for a contract reason
    # (1 rmse, 2 lambda, 3 nothing to optimise, 4 converged, 0 iteration limit) below its initial error
    bad = [(k, r.stop_reason, r.err_init, r.err_final) for k, r in enumerate(res) if r.stop_reason == 5 or not r.err_final <= r.err_init]
    if bad:
        sys.exit("bench.py: %d of %d LM windows ended badly (window, stop_reason, err_init, err_final): %s" % (len(bad), len(res), bad[:4]))
    nkp, nm = cur.counts()
    tracked = d_tr.cpu().numpy()

    # ---- PCIe-inclusive pass (reported beside `value`, never as it): the same step when every batch of images arrives from
    # pinned host memory.  Two resident batches; a copy stream uploads batch k+1 while the step of batch k runs.
    def pcie_pass():
        # (a function of its own: the pinned buffer is released while the context's stream still exists -- torch's host
        # allocator records an event on every stream a pinned block was used on when the block is freed)
        cur2 = capi.KfBatch(ctx, B, W, H, 512)
        bufs = [cur, cur2]
        host = cur.image_tensor().cpu().pin_memory()
        copy_stream = torch.cuda.Stream()
        main = main_stream
        uploaded = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(k):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[k & 1])
                bufs[k & 1].image_tensor().copy_(host, non_blocking=True)
                uploaded[k & 1].record(copy_stream)

        for e in consumed:
            e.record(main)
        torch.cuda.synchronize()
        tp = None
        n_pcie = args.pcie_steps + 1                      # the first step is the warm-up (its upload is not overlapped)
        upload(0)
        for k in range(n_pcie):
            if k == 1:
                torch.cuda.synchronize()
                tp = time.perf_counter()
            upload(k + 1)                                 # (the timed region holds as many uploads as steps: 2 .. n_pcie)
            main.wait_event(uploaded[k & 1])
            step(cur=bufs[k & 1])
            consumed[k & 1].record(main)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp
        # the serial variant: upload, then the step, nothing overlapped
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for k in range(args.pcie_steps):
            with torch.cuda.stream(main):
                cur.image_tensor().copy_(host, non_blocking=True)
            step()
        torch.cuda.synchronize()
        dts = time.perf_counter() - ts
        img_bytes = host.numel()
        pcie = {"value": B * args.pcie_steps / dtp, "unit": "keyframes/s", "ms_per_step": 1e3 * dtp / args.pcie_steps,
                "steps": args.pcie_steps, "host_bytes_per_step": img_bytes,
                "serial_value": B * args.pcie_steps / dts, "serial_ms_per_step": 1e3 * dts / args.pcie_steps,
                "note": "images of every batch uploaded from pinned host memory; value: upload of batch k+1 on a copy stream beside the "
                        "step of batch k (two resident batches), serial_value: upload then step on one stream"}
        cur2.close()
        return pcie

    pcie = pcie_pass() if world == 1 and args.pcie_steps > 0 else None
    torch.cuda.synchronize()

    # ---- un-batched pass (reported beside `value`, never as it): the reference's loop is sequential -- one stereo pair at a time
    # through ssk_slam_process_frame (performStereoSLAM, EST.cpp:44-937) with the KITTI parameters: every call pays its own launches,
    # PCIe copies, host-side window build and symbolic LM set-up; both LM stages per keyframe on the depth-5 local area.
    def unbatched_pass(n):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import stream_full
        o = stream_full.run(n, max(8, n // 10), verbose=False)
        t = o["timing"]
        kf = max(t["kf_frames"], 1)
        return {"frames": n, "keyframes": o["keyframes"], "frames_per_s": o["frames_per_s"], "ms_per_frame": o["ms_per_frame"],
                "ms_per_keyframe_frame": o["ms_per_keyframe_frame"], "worst_frame_ms": o["worst_ms"],
                "per_frame_ms": {"front_end_odometry": t["vo_ms"] / max(t["frames"], 1)},
                "per_keyframe_ms": {"place_recognition_query": t["query_ms"] / kf, "data_association": t["da_ms"] / kf,
                                    "insert_total": t["insert_ms"] / kf, "lm_window_build": t["lm_window_build_ms"] / kf,
                                    "lm_tiler_and_upload": t["lm_batch_create_ms"] / kf, "lm_solve_stage1_plus_stage2": t["lm_solve_ms"] / kf,
                                    "lm_readback": t["lm_readback_ms"] / kf},
                "lm_iters_per_keyframe": t["lm_iters"] / kf, "lm_max_optimised_edges": t["lm_max_opt_edges"], "lm_max_observations": t["lm_max_obs"],
                "note": "synthetic KITTI-shaped sequence (tests/synth.py), 0.54 m per frame, KITTI.ini parameters (keyframe every 4th frame, submap 10, "
                        "tree depth 4, optimise depth 5), host clock around each ssk_slam_process_frame call; frame generation excluded"}

    unbatched = unbatched_pass(args.unbatched_frames) if world == 1 and rank == 0 and args.unbatched_frames > 0 else None

    # ---- the symbolic build inside the loop (reported beside `value`): in the reference the LM window is built from the keyframe's own data and
    # define_new_keyframe does its symbolic update before it optimises (EST.cpp:666-704, 782-784); `value` re-solves windows whose tables were
    # built once.  Here every step solves a batch whose ssk_lm_batch_create (symbolic build on the host's threads + allocation + upload, through
    # a context of its own) ran beside the previous step: two batches alive, a creator thread, the same step otherwise.
    def symbolic_pass(n):
        import threading, queue
        ctx_c = pkg.Context(local)
        mar = capi.LmBatch.marshal(lm_batch_wins)       # (the C view of the windows, once: the creator thread then spends its time inside the library)
        warm = [capi.LmBatch(ctx_c, lm_batch_wins, marshalled=mar) for _ in range(3)]      # (three slabs: one batch solved, one waiting, one built; code objects)
        sym_thr = warm[0].symbolic_ms()[1]
        cur = warm[0]
        for w_ in warm[1:]: w_.close()
        torch.cuda.synchronize()
        # The creator thread owns ctx_c: it builds batch after batch (at most one finished batch waits for the step) and closes the batches the
        # step thread hands back -- a create returns with its uploads queued (round 6), so the next build runs beside them.
        q_ready, q_done = queue.Queue(maxsize=1), queue.Queue()
        failed = []
        def creator():
            try:
                for _ in range(n):
                    while True:
                        try: q_done.get_nowait().close()
                        except queue.Empty: break
                    q_ready.put(capi.LmBatch(ctx_c, lm_batch_wins, marshalled=mar))
            except BaseException as e:                       # (the step thread must not wait for a batch that will not come)
                failed.append(e)
                q_ready.put(None)
        t_create, t_sym = [], []
        th = threading.Thread(target=creator)
        t0s = time.perf_counter()
        th.start()
        for k in range(n):
            step(lm=cur, lm_ctx=ctx_lm)
            torch.cuda.synchronize()
            nxt = q_ready.get()
            if nxt is None:
                raise RuntimeError("symbolic pass: the creator thread failed: %r" % (failed[0],))
            t_create.append(nxt.create_ms); t_sym.append(nxt.symbolic_ms()[0])
            q_done.put(cur)
            cur = nxt
        dts = time.perf_counter() - t0s
        th.join()
        while not q_done.empty(): q_done.get().close()
        cur.close()
        t_alone = []
        for _ in range(3):                                 # (the same create with nothing beside it, right after the loop)
            bq = capi.LmBatch(ctx_c, lm_batch_wins, marshalled=mar)
            t_alone.append((bq.create_ms, bq.symbolic_ms()[0])); bq.close()
        ctx_c.close()
        sym_ms = float(np.mean(t_sym))
        return {"value_incl_symbolic": B * n / dts, "ms_per_step": 1e3 * dts / n, "steps": n,
                "lm_symbolic_ms_per_window": sym_ms / B, "lm_symbolic_ms_per_batch": sym_ms, "host_threads": sym_thr,
                "lm_batch_create_ms": float(np.mean(t_create)),
                "right_after": {"lm_batch_create_ms": float(np.mean([a for a, _ in t_alone])), "lm_symbolic_ms_per_batch": float(np.mean([b_ for _, b_ in t_alone]))},
                "host_cpu_quota": cpu_quota(),
                "note": "every step solves an LM batch created (ssk_lm_batch_create: symbolic build on `host_threads` threads, allocation, upload) beside "
                        "the previous steps by a creator thread with a context of its own (it builds continuously, at most one finished batch waiting; a create "
                        "returns with its uploads queued); lm_symbolic_ms_per_window: host time of the symbolic part / windows of the batch.  The "
                        "builder's threads share the container's CPU quota (host_cpu_quota, cgroup cpu.max: 16 CPUs on the GPU boxes of this pool, whatever "
                        "the core count): back-to-back creates are bound by it, which is why the library caps its builder threads at twice the quota "
                        "(64 threads: 14.9 ms per create, 32: 9.7, 16: 13.8; tools/sym_threads.sh); a single create in an idle process takes ~10 ms "
                        "(symbolic 6.5 ms; tools/lm_create_time.py)"}

    symbolic = symbolic_pass(max(3, min(args.steps, 8))) if world == 1 and rank == 0 and not exchange else None

    # ---- end-to-end pass (reported beside `value`): the LM windows of a step built FROM THE STEP'S OWN front-end output, as the reference does per
    # keyframe (EST.cpp:666-704 -> 782-784): stereo matches -> a10 / a11 (ssk_build_observations_batch: pixels + projectMatchTo3D (+) camera pose)
    # -> the matches tracked against the previous keyframe -> stage 1 of define_new_keyframe for every keyframe of the batch (the new edge
    # previous -> current optimised, the tracked landmarks held at the previous keyframe's triangulation) -> ssk_lm_batch_create (symbolic
    # build) -> solve.  Everything between the images and the solved edges is inside the timed region: downloads, host assembly, create, solve.
    def e2e_pass(n):
        import ctypes as C
        Nf = 512
        d_px_c = torch.zeros(B, Nf, 4, dtype=torch.float32, device=dev)
        d_pos_p = torch.zeros(B, Nf, 3, dtype=torch.float64, device=dev)
        prev.build_observations(lp, None, d_pos_p)             # state, like the previous keyframes' features
        ctx.synchronize()
        pos_p = d_pos_p.cpu().numpy()
        es = np.zeros(1, np.int32); ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
        ar = np.arange(Nf + 1, dtype=np.int32); pe = np.zeros(Nf, np.int32)
        TRACKED = 0                                            # EST.h:102
        ar_dev = torch.arange(N, device=dev)
        t_ph = np.zeros(5)
        stats = []
        def one(timed):
            t0 = time.perf_counter()
            cur.run_frontend(fp)
            ctx.da_run_batch(cur, prev, pairs, da, d_idx, d_dist, d_st, d_tr)
            ctx.fm_run_batch(cur, prev, pairs, fm, d_idx, d_st, d_tr)
            ctx.cip_run_batch(cur, prev, pairs, cp, d_idx, d_st, d_tr)
            cur.bow_transform(voc); cur.bow_query(db, 4)
            cur.build_observations(lp, d_px_c, None)
            ctx.synchronize()
            _, nm_c = cur.counts()
            t1 = time.perf_counter()
            # the tracked matches of every keyframe, compacted ON THE DEVICE (mask -> nonzero -> gathers), then ONE download of the compact arrays
            # (stereo pixels of the current keyframe's match, the previous keyframe's triangulation of the matched landmark)
            st_d, idx_d = d_st.view(B, N), d_idx.view(B, N)
            nm_d = torch.as_tensor(np.asarray(nm_c, np.int64), device=dev)
            mask = (st_d == TRACKED) & (ar_dev[None, :] < nm_d[:, None]) & (idx_d >= 0)
            cnt_d = mask.sum(1)
            bb_d, ii_d = mask.nonzero(as_tuple=True)
            jj_d = idx_d[bb_d, ii_d].long()
            off_d = torch.cumsum(cnt_d, 0) - cnt_d
            obs_px = d_px_c[bb_d, ii_d].contiguous().cpu().numpy()             # [M, 4] float32
            lm_pos = d_pos_p[bb_d, jj_d].contiguous().cpu().numpy()            # [M, 3] float64
            obs_lm = (torch.arange(bb_d.numel(), device=dev) - off_d[bb_d]).to(torch.int32).cpu().numpy()
            cnt = cnt_d.cpu().numpy().astype(np.int64)
            off = np.concatenate([[0], np.cumsum(cnt)])
            lm_slot = -np.ones(len(obs_lm), np.int32)
            rec = np.zeros(B, capi.LM_WINDOW_DTYPE)                            # the batch's ssk_lm_window array, column by column
            o8 = off[:B].astype(np.uint64)
            rec["n_edges_total"] = 1; rec["edge_unknown_slot"] = es.ctypes.data; rec["edge_inv_pose"] = ident.ctypes.data
            rec["n_lms_total"] = cnt; rec["lm_unknown_slot"] = lm_slot.ctypes.data + 4 * o8; rec["lm_pos"] = lm_pos.ctypes.data + 24 * o8
            rec["n_obs"] = cnt; rec["obs_lm"] = obs_lm.ctypes.data + 4 * o8; rec["obs_px"] = obs_px.ctypes.data + 16 * o8
            rec["obs_path_off"] = ar.ctypes.data; rec["path_edge"] = pe.ctypes.data
            arr = capi.lm_window_array(rec)
            t2 = time.perf_counter()
            lb = capi.LmBatch(ctx, [None] * B, marshalled=(arr, [rec, es, ident, ar, pe, lm_slot, lm_pos, obs_lm, obs_px]))
            t3 = time.perf_counter()
            lb.run(lp)
            r = lb.results()
            t4 = time.perf_counter()
            if timed:
                t_ph[:] += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
                stats.append((float(cnt.mean()), float(np.mean([x.iters for x in r])), float(np.mean([x.err_final <= x.err_init for x in r])), lb.symbolic_ms()[0]))
            lb.close()
        one(False)
        for _ in range(n):
            one(True)
        t_ph /= n
        st_ = np.mean(np.array(stats), 0)
        return {"value": B / t_ph[4], "unit": "keyframes/s", "ms_per_step": 1e3 * t_ph[4], "steps": n,
                "ms": {"front_end_da_a10_a11": 1e3 * t_ph[0], "window_assembly": 1e3 * t_ph[1], "lm_batch_create": 1e3 * t_ph[2],
                       "lm_solve_and_results": 1e3 * t_ph[3]},
                "lm_symbolic_ms_per_window": st_[3] / B, "mean_tracked_observations": st_[0], "mean_lm_iters": st_[1], "windows_with_error_not_increased": st_[2],
                "note": "every step: front-end + DA of the batch, ssk_build_observations_batch, download of the tracked matches, stage-1 windows of "
                        "define_new_keyframe (the new edge optimised, the tracked landmarks held at the previous keyframe's triangulation): the tracked matches compacted on the "
                        "device, one download, the ssk_lm_window array filled on the host, ssk_lm_batch_create, solve, results -- nothing prepared outside the timed region but the previous "
                        "keyframes' state; serial (no overlap of host work with the device)"}

    e2e = e2e_pass(3) if world == 1 and rank == 0 and not exchange else None

    # ---- end-to-end pass WITH stage 2 (round 5; reported beside `value` as `e2e_stage2`): B independent keyframe streams advanced in LOCK-STEP.  Every
    # step every stream adds one keyframe: front-end of the batch's images -> data association against the stream's previous keyframe (+ the
    # fundamental-matrix and change-in-pose filters, place recognition) -> a10 / a11 (pixels, projectMatchTo3D (+) camera pose) -> landmark ids (a tracked
    # match inherits the id of the previous keyframe's match, the others are new landmarks, EST.cpp:666-704) -> define_new_keyframe of every stream
    # (EST.cpp:782-784) through ssk_rba_define_kf_batch: graph bookkeeping + both LM windows per stream on host threads, stage 1 of all streams ONE
    # LM batch, the local areas (optimise depth `depth`) another, results written back to the streams' maps.  Everything is inside the timed region.
    # The streams alternate between the batch's two image sets (each step the other set is the new keyframe: the same scene from the other
    # viewpoint), so a stream's landmarks are re-observed keyframe after keyframe and its local areas reach their steady size after
    # `warm` keyframes (submap_size 10, depth 2: the headline's E = 10 window shape; depth 5: the reference's KITTI.ini:104-106).
    # Round 6: the streams are independent, so they are dealt out over `groups` host threads, each with a context (HIP stream) of its own and its share of
    # the streams: while one group's host side builds windows / symbolic tables, another group's kernels run -- the serial form left the device idle for
    # two thirds of a step.  groups = 1 is the serial form (reported as `serial`).
    def e2e_stage2_pass(depth, n_streams, warm, timed, groups=1):
        import threading
        flib = capi.load_facade_library()
        opts = capi.RbaOptions.kitti(submap_size=10, max_tree_depth=4, max_optimize_depth=depth)
        G = max(1, min(groups, n_streams))
        bounds = [n_streams * g // G for g in range(G + 1)]

        class Group:
            def __init__(self, g):
                self.lo, self.hi = bounds[g], bounds[g + 1]
                Bs = self.Bs = self.hi - self.lo
                self.ctx = ctx if G == 1 else pkg.Context(local)
                self.kf = [capi.KfBatch(self.ctx, Bs, W, H, 512), capi.KfBatch(self.ctx, Bs, W, H, 512)]
                self.kf[0].image_tensor().copy_(cur.image_tensor()[self.lo:self.hi]); self.kf[1].image_tensor().copy_(prev.image_tensor()[self.lo:self.hi])
                self.prs = np.stack([np.arange(Bs), np.arange(Bs)], 1).astype(np.int32)
                self.s_idx = torch.empty(Bs * N, dtype=torch.int32, device=dev); self.s_dist = torch.empty(Bs * N, dtype=torch.int32, device=dev)
                self.s_st = torch.empty(Bs * N, dtype=torch.int32, device=dev); self.s_tr = torch.empty(Bs, dtype=torch.int32, device=dev)
                self.d_px = torch.zeros(Bs, N, 4, dtype=torch.float32, device=dev); self.d_pos = torch.zeros(Bs, N, 3, dtype=torch.float64, device=dev)
                self.engines = [capi.Rba(self.ctx, opts) for _ in range(Bs)]
                self.handles = (C.c_void_p * Bs)(*[e.h_ for e in self.engines])
                self.ids_prev = None
                self.ids_cur = np.empty((Bs, N), np.int64)
                self.next_id = np.zeros(Bs, np.int64)
                self.obs = np.zeros((Bs, N), capi.RBA_OBS_DTYPE)
                self.ar = np.arange(N)[None, :]
                self.t_ph = np.zeros(4)
                self.last_tm, self.stats, self.per_step = None, [], []
                self.error = None

            def one(self, k, timed_):
                Bs = self.Bs
                c, p = self.kf[k % 2], self.kf[(k + 1) % 2]
                t0 = time.perf_counter()
                c.run_frontend(fp)
                if k > 0:
                    self.ctx.da_run_batch(c, p, self.prs, da, self.s_idx, self.s_dist, self.s_st, self.s_tr)
                    self.ctx.fm_run_batch(c, p, self.prs, fm, self.s_idx, self.s_st, self.s_tr)
                    self.ctx.cip_run_batch(c, p, self.prs, cp, self.s_idx, self.s_st, self.s_tr)
                    c.bow_transform(voc); c.bow_query(db, 4)
                c.build_observations(lp, self.d_px, self.d_pos)
                self.ctx.synchronize()
                t1 = time.perf_counter()
                _, nm = c.counts()
                nm = np.minimum(np.asarray(nm, np.int64), N)
                px = self.d_px.cpu().numpy(); pos = self.d_pos.cpu().numpy()
                st = self.s_st.view(Bs, N).cpu().numpy() if k > 0 else None
                ti = self.s_idx.view(Bs, N).cpu().numpy() if k > 0 else None
                nm32 = nm.astype(np.int32)
                # EST.cpp:666-704 for the batch: tracked matches inherit the previous keyframe's landmark ids, the others found new landmarks (library entry)
                rc = flib.ssk_rba_obs_from_da_batch(Bs, N, capi._ptr(nm32), capi._ptr(st) if k > 0 else None, capi._ptr(ti) if k > 0 else None, capi._ptr(px), capi._ptr(pos),
                                                    capi._ptr(self.ids_prev) if k > 0 else None, capi._ptr(self.next_id), capi._ptr(self.ids_cur), capi._ptr(self.obs))
                assert rc == 0
                self.ids_prev, self.ids_cur = self.ids_cur, (self.ids_prev if self.ids_prev is not None else np.empty((Bs, N), np.int64))
                t2 = time.perf_counter()
                infos, tm = capi.rba_define_kf_batch(self.engines, (self.obs, nm32), optimize=k > 0, handles=self.handles)
                t3 = time.perf_counter()
                if timed_:
                    self.t_ph[:] += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
                    self.last_tm = tm
                    self.per_step.append(dict(tm, step_ms=1e3 * (t3 - t0)))
                    tracked_n = float(((st == 0) & (ti >= 0) & (self.ar < nm[:, None])).sum(1).mean()) if k > 0 else 0.0      # (statistics: outside the timed interval)
                    self.stats.append((tracked_n, float(np.mean([i.total_sqr_error_final <= i.total_sqr_error_init * (1 + 1e-12) for i in infos]))))

            def close(self):
                for e in self.engines:
                    e.close()
                self.kf[0].close(); self.kf[1].close()
                if self.ctx is not ctx:
                    self.ctx.close()

        grp = [Group(g) for g in range(G)]
        torch.cuda.synchronize()
        # groups side by side share the box's CPU quota: each gets its share of the host threads (1.5 x the one-group count in all -- 4 groups x 12
        # threads measured best under a 16-CPU quota: 30.3 k keyframes/s against 21.6-26.6 k for 2 groups x 32); the library reads both per call
        env_keep = {k_: os.environ.get(k_) for k_ in ("SSK_RBA_BATCH_THREADS", "SSK_LM_BUILD_THREADS")}
        if G > 1 and all(v_ is None for v_ in env_keep.values()):
            one_group = int(max(2, min((os.cpu_count() or 2) // 2, 64, round(2 * (cpu_quota() or 1e9)))))
            per_group = str(max(4, (3 * one_group // 2) // G))
            os.environ["SSK_RBA_BATCH_THREADS"] = per_group; os.environ["SSK_LM_BUILD_THREADS"] = per_group
        gate = threading.Barrier(G + 1)
        def run(gr):
            try:
                for k in range(warm):
                    gr.one(k, False)
                gate.wait()                                        # (every group warm: the clock starts)
                for k in range(warm, warm + timed):
                    gr.one(k, True)
            except BaseException as ex:                          # (a failed group must not leave the others at the barrier)
                gr.error = ex
                gate.abort()
        th = [threading.Thread(target=run, args=(gr,)) for gr in grp]
        for t_ in th:
            t_.start()
        try:
            gate.wait()
        except threading.BrokenBarrierError:
            pass
        tw0 = time.perf_counter()
        for t_ in th:
            t_.join()
        wall = time.perf_counter() - tw0
        for k_, v_ in env_keep.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_
        for gr in grp:
            if gr.error is not None:
                raise gr.error
        g0 = grp[0]
        t_ph = g0.t_ph / timed
        last_tm, per_step = g0.last_tm, g0.per_step
        st_ = np.mean(np.array([x for gr in grp for x in gr.stats]), 0)
        for gr in grp:
            gr.close()
        Bs = n_streams
        # (means over the timed steps: a submap's cycle has local areas of 1 .. 10+ edges, the last step alone says little; maxima stay maxima)
        tmd = {}
        for k2, v in last_tm.items():
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                vals = [d[k2] for d in per_step]
                tmd[k2] = (max(vals) if k2.startswith("stage2_max") else round(float(np.mean(vals)), 4)) if k2 != "host_threads" else v
            else:
                tmd[k2] = v
        if os.environ.get("SSK_BENCH_E2E_STEPS") == "1":
            tmd["per_step"] = [{k2: (round(v, 3) if isinstance(v, float) else v) for k2, v in d.items()} for d in per_step]
        return {"value": Bs * timed / wall, "unit": "keyframes/s", "streams": Bs, "stream_groups": G, "optimize_depth": depth, "keyframes_per_stream_before_timing": warm, "steps": timed,
                "ms_per_step": 1e3 * wall / timed,
                "group0_streams": g0.Bs, "group0_ms_per_step": 1e3 * t_ph[3],
                "group0_ms": {"front_end_da_a10_a11": 1e3 * t_ph[0], "download_and_landmark_ids": 1e3 * t_ph[1], "define_new_keyframe_batch": 1e3 * t_ph[2]},
                "define_new_keyframe_batch": tmd, "mean_tracked_vs_prev": st_[0], "windows_with_error_not_increased": st_[1], "host_cpu_quota": cpu_quota(),
                "note": "B independent streams: per step and stream one new keyframe -- front-end, data association against the stream's previous "
                        "keyframe, the two outlier filters, place recognition, a10 / a11, landmark ids, then define_new_keyframe (EST.cpp:782-784) of all "
                        "streams of a group by ssk_rba_define_kf_batch: stage 1 (the new edge, landmarks held: lm_pose_only_kernel) as ONE LM batch, the local area "
                        "(optimise depth as stated; submap_size 10, max_tree_depth 4) as another, results written back to every stream's map.  Host parts "
                        "(graph bookkeeping, window building, the LM batches' symbolic builds) run on `host_threads` threads under the box's CPU quota.  "
                        "`stream_groups` groups of streams, each in lock-step on a host thread and a HIP stream of its own, run side by side (one group's host "
                        "work beside another's kernels); value = streams x timed steps / wall time of all groups; group0_*: the first group's own clock "
                        "(its steps overlap the other groups'); define_new_keyframe_batch: means over group 0's timed steps"}

    e2e_stage2 = None
    if world == 1 and rank == 0 and not exchange and args.e2e_streams > 0:
        # (21 keyframes before the clock: two cycles of a 10-keyframe submap -- the pinned table arena and the slabs have their steady sizes; ten timed
        # steps: one full cycle, local areas of 1 .. 10+ edges)
        e2e_stage2 = e2e_stage2_pass(2, min(args.e2e_streams, B), 21, 10, groups=args.e2e_groups)
        if args.e2e_groups > 1:
            ser = e2e_stage2_pass(2, min(args.e2e_streams, B), 21, 10, groups=1)
            e2e_stage2["serial"] = {k2: ser[k2] for k2 in ("value", "ms_per_step", "group0_ms", "define_new_keyframe_batch")}
        if args.e2e_depth5_streams > 0:
            e2e_stage2["depth5"] = e2e_stage2_pass(5, min(args.e2e_depth5_streams, B), 44, 2, groups=min(args.e2e_groups, 2))

    # ---- the reference's real per-keyframe LM load (reported beside `value`): define_new_keyframe does TWO solves per keyframe
    # (EST.cpp:782-784): stage 1 = optimize_edges on the new edge with the landmarks held, stage 2 = optimize_local_area with
    # srba_max_optimize_depth = 5 (KITTI.ini:105): ~40 edges.  The headline's window is BASELINE configs[2]'s "window = 10".
    def lm_depth5_pass(n):
        wins5 = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20),
                                        z_vis=(2.5, 22))[0] for s in range(4)]
        def stage1(w):
            newest = int(np.flatnonzero(w.edge_unknown_slot == w.E - 1)[0])
            keep = [o for o in range(len(w.obs_lm)) if w.obs_path_off[o + 1] > w.obs_path_off[o] and (w.path_edge[w.obs_path_off[o]] >> 1) == newest]
            off, pe = [0], []
            for o in keep:
                pe.extend(w.path_edge[w.obs_path_off[o]:w.obs_path_off[o + 1]]); off.append(len(pe))
            return sl.Window(np.where(w.edge_unknown_slot == w.E - 1, 0, -1), w.edge_inv_pose, -np.ones(len(w.lm_pos), np.int32), w.lm_pos,
                             w.obs_lm[keep], w.obs_px[keep], off, pe)
        st1 = [stage1(w) for w in wins5]
        out = {"window": {"stage2_edges": int(wins5[0].E), "stage2_observations": int(len(wins5[0].obs_lm)), "stage2_landmarks": int(wins5[0].L),
                          "stage1_observations": int(len(st1[0].obs_lm))}}
        def timed(batch_wins, reps):
            b = capi.LmBatch(ctx, batch_wins)
            b.run(lp); b.results()                       # warm-up (code objects, slab)
            t = 0.0
            for _ in range(reps):
                b.reset(); ctx.synchronize()
                t0 = time.perf_counter()
                b.run(lp); r = b.results()
                t += time.perf_counter() - t0
            assert all(x.stop_reason != 5 and x.err_final <= x.err_init for x in r)
            it = float(np.mean([x.iters for x in r]))
            b.close()
            return 1e3 * t / reps, it
        def timed_named(batch_wins, reps):
            b = capi.LmBatch(ctx, batch_wins)
            name = b.kernel_name()
            b.close()
            ms, it = timed(batch_wins, reps)
            return ms, it, name
        ms1, it1, k1 = timed_named([st1[k % 4] for k in range(n)], 3)
        ms2, it2, k2 = timed_named([wins5[k % 4] for k in range(n)], 3)
        out["batched"] = {"windows": n, "stage1_ms": ms1, "stage2_ms": ms2, "keyframes_per_s": n / ((ms1 + ms2) * 1e-3), "stage1_iters": it1, "stage2_iters": it2,
                          "stage1_kernel": k1, "stage2_kernel": k2}
        # the stage-2 launch priced in FP64 flops (SURVEY 8(d) per-unit figures, as roofline_fp64): the kernel that carries the reference's real load
        fl = float(np.mean([lm_algorithmic_flops(w) for w in wins5])) * it2 * n
        out["roofline_depth5"] = {"kernel": k2, "bound": "fp64", "achieved": fl / (ms2 * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": fl / (ms2 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": fl, "avg_launch_ms": ms2,
                                  "note": "host clock around ssk_lm_batch_run + results of %d windows (phase-split kernels: linearise / reduce / solve / score / decide per LM pass, ssk_lm_split.h); the dense solve of K2 runs on the matrix cores (v_mfma_f64_16x16x4_f64; FP64 matrix rate = vector rate on MI355X), the rest on the vector ALUs; kernel-trace and PMC summaries: profiles/r06_lm_depth5*" % n}
        # round 5: a window at the 90th percentile of what the full-length un-batched stream produces (profiles/r05_stream_stage2_windows.json: stage-2
        # local areas of 32 / 38 / 68 / 94 optimised edges at the 10th / 50th / 90th percentile / maximum): optimise depth 8 -> 70 edges, 15 k observations
        try:
            w90 = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=8, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(2)]
            n90 = min(n, 256)
            ms90, it90, k90 = timed_named([w90[k % 2] for k in range(n90)], 2)
            l90, _, _ = timed_named([w90[0]], 5)
            b90 = capi.LmBatch(ctx, [w90[0]]); sv90 = b90.solver_name(); b90.close()
            out["p90_window"] = {"stage2_edges": int(w90[0].E), "stage2_observations": int(len(w90[0].obs_lm)), "windows": n90, "stage2_ms": ms90, "stage2_iters": it90,
                                 "keyframes_per_s_stage2_only": n90 / (ms90 * 1e-3), "one_keyframe_stage2_ms": l90, "stage2_kernel": k90, "dense_solver": sv90}
        except Exception as e:
            out["p90_window"] = {"error": str(e)}
        l1, _, kk1 = timed_named([st1[0]], 10)
        l2, _, kk2 = timed_named([wins5[0]], 10)
        out["one_keyframe"] = {"stage1_ms": l1, "stage2_ms": l2, "stage1_kernel": kk1, "stage2_kernel": kk2,
                               "note": "one window per launch (stage 2: its tiles dealt out over up to 32 workgroups), host clock around ssk_lm_batch_run + results, structures already built"}
        # the CPU oracle on the same stage-2 window (1 thread)
        try:
            import oracle_lib
            orc = oracle_lib.load()
            t0 = time.perf_counter()
            orc.lm_solve(wins5[0], sl.make_params(), trace=False)
            out["cpu_ms"] = {"stage2_ms": 1e3 * (time.perf_counter() - t0), "cores": 1, "kind": "port", "sample": "one stage-2 window (the CPU oracle, 1 thread)"}
        except Exception as e:       # the oracle is test infrastructure: its absence does not stop the bench
            out["cpu_ms"] = {"error": str(e)}
        return out

    lm_depth5 = lm_depth5_pass(args.lm_depth5) if world == 1 and rank == 0 and args.lm_depth5 > 0 else None

    if rank == 0:
        # PMC traffic per launch (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, summary
        # committed under profiles/): bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- KB units, FETCH_SIZE doubled on gfx950
        # (MI355X_MICROARCH.md "HBM"; re-calibrated in the same run on lm_reset_kernel, which reads exactly what it writes)
        LM_KERNEL = lm.kernel_name()
        traffic, traffic_src, pmc_ms = {}, None, {}
        for tf in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):          # the newest summary collected at this batch size
            tfp = os.path.join(ROOT, "profiles", tf)
            if traffic or not os.path.exists(tfp):
                continue
            try:
                t = json.load(open(tfp))
                if int(t.get("batch", -1)) == B:
                    traffic = {k: (2.0 * v["fetch_size_kb"] + v["write_size_kb"]) * 1024.0 for k, v in t["kernels"].items()}
                    pmc_ms = {k: v.get("avg_ms") for k, v in t["kernels"].items()}
                    traffic_src = {"file": "profiles/" + tf, "git_head": t.get("git_head")}
            except Exception:
                traffic = {}

        def traffic_of(kernel, live_ms):
            """PMC traffic of a kernel, or None when the summary does not belong to this build: the kernel's average duration in the
            profiled run (stamped into the summary by tools/pmc_traffic.py) must agree with the live one within 8 % (the boxes of the pool differ by ~6 % on this kernel)."""
            if kernel not in traffic:
                return None
            ref = pmc_ms.get(kernel)
            if ref is None or not live_ms or abs(ref / live_ms - 1.0) > 0.08:
                return None
            return traffic[kernel]
        # executed FP64 flops from the PMC pass of the same command (tools/pmc_lm.sh: SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 x 64 lanes)
        pmc_flop = None
        pl = os.path.join(ROOT, "profiles", "r05_pmc_lm_wave_time.json")
        if not os.path.exists(pl):
            pl = os.path.join(ROOT, "profiles", "r04_pmc_lm_wave_time.json")
        if os.path.exists(pl):
            try:
                pmc_flop = json.load(open(pl)).get(LM_KERNEL, {}).get("derived", {}).get("f64_flop")
            except Exception:
                pmc_flop = None
        # dominant kernel by time: the LM solve.  Algorithmic bytes: per LM iteration 56 B per observation
        # (pixels 16 + ids 16 + landmark 24) + 96 B per edge pose, nothing written when J is fused into the H accumulation
        lm_ms, lm_calls = prof["lm_solve"]
        lm_avg_ms = lm_ms / max(lm_calls, 1)
        lm_algo = float(sum(r.iters * (56 * len(lm_batch_wins[k].obs_lm) + 96 * len(lm_batch_wins[k].edge_unknown_slot)) for k, r in enumerate(res)))
        lm_ach = lm_algo / (lm_avg_ms * 1e-3) / 1e9
        per_win_flops = [lm_algorithmic_flops(w) for w in wins]
        lm_flops = float(sum(r.iters * (per_win_flops[k % len(wins)] if lm_batch_wins[k] is wins[k % len(wins)] else lm_algorithmic_flops(lm_batch_wins[k]))
                             for k, r in enumerate(res)))
        lm_tflops = lm_flops / (lm_avg_ms * 1e-3) / 1e12
        # the HBM-bound scan of the front-end (every pixel of both images read once): 933,232 B per keyframe
        fast_ms, fast_calls = prof["fast_nms"]
        fast_avg_ms = fast_ms / max(fast_calls, 1)
        algo_bytes = float(B) * 2 * W * H
        achieved = algo_bytes / (fast_avg_ms * 1e-3) / 1e9
        # the orientation + BRIEF kernel: per keypoint a 45 x 48 B patch read (2160 B; the aligned-dword window of the 45 x 45 patch), 32 B of
        # descriptor + 4 B of angle written
        desc_ms, desc_calls = prof["describe"]
        desc_avg_ms = desc_ms / max(desc_calls, 1)
        desc_algo = float(nkp.sum()) * (45 * 48 + 36)
        desc_ach = desc_algo / (desc_avg_ms * 1e-3) / 1e9
        stages = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items() if v[1] > 0}
        line = {
            "metric": "keyframes/sec (detect+match+LM solve), KITTI-00 stereo",
            "value": world * B * args.steps / dt,
            "unit": "keyframes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32 front-end + DA, f64 LM",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]+[2] per keyframe: 1241x376 stereo pair -> ORB(N=500, FAST 20, 1 level) + L<->R "
                            "(max_y_diff 2, orb_max_distance 60) + KF<->KF Hamming DA with direction/ORB/fundamental-matrix/change-in-pose filters + bag-of-words "
                            "place-recognition query (k=8, L=5 vocabulary, %d-keyframe database, 4 results) + SRBA LM window " % BOW_DB_ENTRIES +
                            "(200-KF/5k-landmark submap, submap_size 10, max_tree_depth 4, window = 10 most recent edges)",
                "keyframes_per_gpu_per_step": B,
                "streams": ("front-end + DA on one HIP stream, LM solves on a second one" if args.overlap else "one HIP stream") + ("; place recognition on a side stream beside the data-association chain" if args.side_stream else ""),
                "parallelism": "keyframe streams partitioned one batch per GPU" + ("; configs[4]: every rank's window 0 is its submap of the shared set (%d landmarks of one identity "
                               "observed from 2 of the %d submaps), 1 RCCL all-gather of their g/H blocks per step + Gauss-Newton update" % (S, max(world, 2)) if exchange else ""),
                "mean_keypoints_per_image": float(nkp.mean()),
                "mean_stereo_matches": float(nm.mean()),
                "mean_tracked_vs_prev": float(tracked.mean()),
                "exchange_rounds_to_1e-6": exchange_rounds,
                "rccl_world": world if exchange else None,
                "lm_distinct_windows": len(wins),
                "lm_window": {"edges": int(wins[0].E), "landmarks": int(wins[0].L), "observations": int(len(wins[0].obs_lm)),
                              "mean_iters": float(np.mean([r.iters for r in res]))},
            },
            "roofline": {"kernel": LM_KERNEL, "bound": "hbm", "achieved": lm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": lm_ach / HBM_PEAK_GBS, "traffic": traffic_of(LM_KERNEL, lm_avg_ms), "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": lm_algo, "avg_launch_ms": lm_avg_ms,
                         "note": "dominant kernel by time; not bandwidth bound: two persistent workgroups per CU, waves wait on LDS / FP64 dependencies "
                                 "(profiles/r06_pmc_lm_wave_time.json, DESIGN.md 5); traffic is null when the committed PMC summary's kernel time differs from the live one by more than 8 %"},
            "roofline_fp64": {"kernel": LM_KERNEL, "bound": "fp64", "achieved": lm_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": lm_tflops / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": lm_flops,
                              "executed_flops_per_launch_pmc": pmc_flop,
                              "executed_frac_pmc": (pmc_flop / (lm_avg_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if pmc_flop else None,
                              "note": "the same launch priced in FP64 flops: `achieved` from SURVEY 8(d)'s per-unit figures (an upper estimate of the work: the node form "
                                      "executes less), `executed_*_pmc` from the SQ_INSTS_VALU_*_F64 counters of the committed PMC pass; FP64 MFMA rate = FP64 vector rate on MI355X"},
            "roofline_scan": {"kernel": "fast_nms_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_of("fast_nms_kernel", fast_avg_ms),
                              "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": fast_avg_ms},
            "roofline_describe": {"kernel": "describe_kernel", "bound": "hbm", "achieved": desc_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": desc_ach / HBM_PEAK_GBS, "traffic": traffic_of("describe_kernel", desc_avg_ms),
                                  "algorithmic_bytes_per_launch": desc_algo, "avg_launch_ms": desc_avg_ms,
                                  "note": "VALU-issue bound (one wavefront per keypoint, ~545 vector instructions each); an image's keypoints run on one XCD, so its patches meet in one L2"},
            "stage_ms_per_launch": stages,
            "value_incl_symbolic": symbolic["value_incl_symbolic"] if symbolic else None,
            "lm_symbolic_ms_per_window": symbolic["lm_symbolic_ms_per_window"] if symbolic else None,
            "lm_symbolic": symbolic,
            "e2e": e2e,
            "e2e_stage2": e2e_stage2,
            "exchange_ms_per_rank": exchange_ms_all if exchange else None,
            "pcie_inclusive": pcie,
            "unbatched": unbatched,
            "lm_depth5": lm_depth5,
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, wins, vocab)
            procs = args.cpu_procs if args.cpu_procs >= 0 else max(1, min(64, (os.cpu_count() or 2) // 2))
            if procs > 0:
                line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(procs, max(32, args.cpu_sample // 2))
        else:
            line["cpu_baseline"] = None
        # RCCL writes its version banner through C stdio, which is flushed at exit when stdout is a pipe: push it out now
        # so that the JSON line is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)
    db.close(); voc.close(); lm.close(); cur.close(); prev.close()
    if ctx_side:
        ctx_side.close()
    if ctx_lm is not ctx:
        ctx_lm.close()
    ctx.close()
    if dist.is_initialized():
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
