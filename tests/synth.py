"""Seeded synthetic inputs shaped like the reference's keyframes."""
import numpy as np

from oracle_lib import DMATCH_DTYPE, KEYPOINT_DTYPE

W, H = 1241, 376          # BASELINE.json image shape
KITTI = dict(fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104, baseline=0.54, rows=370, cols=1226)


def random_descriptors(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(rng, desc, nbits):
    """Return a copy of desc with nbits[i] random bit flips in row i."""
    out = desc.copy()
    for i, nb in enumerate(np.broadcast_to(nbits, (len(desc),))):
        pos = rng.choice(256, size=int(nb), replace=False)
        for p in pos:
            out[i, p >> 3] ^= np.uint8(1 << (p & 7))
    return out


def synth_kf(rng, n_left=500, n_right=500, n_matches=300, integer_px=True):
    """One keyframe: keypoints L/R, descriptors L/R, L<->R matches (queryIdx=left, trainIdx=right)."""
    kpl = np.zeros(n_left, KEYPOINT_DTYPE)
    kpr = np.zeros(n_right, KEYPOINT_DTYPE)
    for kp, n in ((kpl, n_left), (kpr, n_right)):
        xs = rng.uniform(31, W - 32, n)
        ys = rng.uniform(31, H - 32, n)
        if integer_px:
            xs, ys = np.floor(xs), np.floor(ys)
        kp["x"], kp["y"] = xs, ys
        kp["size"], kp["angle"], kp["response"] = 31, rng.uniform(0, 360, n), rng.uniform(0, 1, n)
    dl = random_descriptors(rng, n_left)
    dr = random_descriptors(rng, n_right)
    n_matches = min(n_matches, n_left, n_right)
    m = np.zeros(n_matches, DMATCH_DTYPE)
    m["queryIdx"] = np.sort(rng.choice(n_left, n_matches, replace=False))
    m["trainIdx"] = rng.choice(n_right, n_matches, replace=False)
    m["distance"] = rng.integers(0, 60, n_matches)
    return dict(kp_left=kpl, kp_right=kpr, desc_left=dl, desc_right=dr, matches=m)


def synth_kf_pair(rng, n_feats=500, n_matches=300, overlap=0.7, motion=(6.0, -2.0), noise_bits=12, outlier_frac=0.1):
    """Two keyframes where a fraction `overlap` of kf_b's matched left features reappear in kf_a moved by
    roughly `motion` pixels with `noise_bits`-ish descriptor bit flips; a few are gross outliers."""
    b = synth_kf(rng, n_feats, n_feats, n_matches)
    a = synth_kf(rng, n_feats, n_feats, n_matches)
    nm = len(b["matches"])
    n_shared = int(overlap * min(nm, len(a["matches"])))
    src = rng.choice(nm, n_shared, replace=False)
    dst = rng.choice(len(a["matches"]), n_shared, replace=False)
    for s, d in zip(src, dst):
        ib = b["matches"]["queryIdx"][s]
        ia = a["matches"]["queryIdx"][d]
        bits = int(rng.integers(0, 2 * noise_bits + 1))
        a["desc_left"][ia] = flip_bits(rng, b["desc_left"][ib:ib + 1], bits)[0]
        if rng.uniform() < outlier_frac:
            continue   # descriptor matches, geometry does not
        a["kp_left"]["x"][ia] = np.clip(np.floor(b["kp_left"]["x"][ib] + motion[0] + rng.normal(0, 1.5)), 31, W - 32)
        a["kp_left"]["y"][ia] = np.clip(np.floor(b["kp_left"]["y"][ib] + motion[1] + rng.normal(0, 1.5)), 31, H - 32)
    return a, b


def synth_texture(rng, w, h):
    """Street-scene-like synthetic image (uint8): a smooth, low-contrast background (linear noise at 16 px + weak cubic
    noise at 4 px: gradients mostly below the FAST threshold, like road / sky / facades) plus random rectangles whose
    corners and edges are the features.  On a KITTI-sized image FAST at threshold 20 finds ~3k corners after 3x3 NMS
    (~4 % corner pixels, ~7 % of the pixels pass the 4-point pre-test), the density of a KITTI frame."""
    from scipy.ndimage import zoom
    coarse = zoom(rng.uniform(0, 1, (h // 16 + 2, w // 16 + 2)), 16, order=1)[:h, :w]
    mid = zoom(rng.uniform(0, 1, (h // 4 + 2, w // 4 + 2)), 4, order=3)[:h, :w]
    img = 95 + 70 * (coarse - 0.5) + 25 * (mid - 0.5)
    for _ in range(int(w * h / 350)):
        x0, y0 = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
        ww, hh = int(rng.integers(3, 24)), int(rng.integers(3, 24))
        img[y0:y0 + hh, x0:x0 + ww] += rng.uniform(-70, 70)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_stereo_pair(seed, w=W, h=H, noise=2.0):
    """Rectified stereo pair of a scene made of fronto-parallel strips: row block r has integer disparity
    d_r = fx*b/Z_r with Z in [5,80] m (KITTI.ini intrinsics), right(y,x) = left(y,x+d_r) + sensor noise."""
    rng = np.random.default_rng(0x5EED0000 + seed)
    wide = synth_texture(rng, w + 96, h)
    left = wide[:, :w].copy()
    right = np.empty_like(left)
    y = 0
    while y < h:
        bh = int(rng.integers(24, 64))
        z = rng.uniform(5, 80)
        d = int(round(KITTI["fx"] * KITTI["baseline"] / z))
        d = max(1, min(d, 90))
        right[y:y + bh] = wide[y:y + bh, d:d + w]      # x_R = x_L - d
        y += bh
    if noise > 0:
        right = np.clip(np.rint(right.astype(np.float64) + rng.normal(0, noise, right.shape)), 0, 255).astype(np.uint8)
    return left, right


def synth_translation_sequence(seed, n_frames, w=W, h=H, noise=1.0, d_max=40, z_max=80.0, step=1):
    """n_frames rectified stereo pairs of a scene of fronto-parallel strips seen from a camera that moves one baseline
    (0.54 m) to the RIGHT per frame: strip r has integer disparity d_r, so frame k sees it shifted by k*d_r pixels.
    z_max bounds the depth and `step` is the number of baselines the camera moves per frame: with z_max = 12 m and step = 4
    every strip moves >= 128 px per frame, so nothing stays in view for more than 10 frames.
    Returns ([(left, right)], strip list [(y0, y1, d_r)])."""
    rng = np.random.default_rng(0x5EED1000 + seed)
    wide = synth_texture(rng, w + (n_frames * step + 1) * d_max + 8, h)
    strips, y = [], 0
    while y < h:
        bh = int(rng.integers(24, 64))
        z = rng.uniform(KITTI["fx"] * KITTI["baseline"] / d_max, z_max)
        d = max(2, min(int(round(KITTI["fx"] * KITTI["baseline"] / z)), d_max))
        strips.append((y, min(h, y + bh), d))
        y += bh
    frames = []
    for k in range(n_frames):
        left = np.empty((h, w), np.uint8)
        right = np.empty((h, w), np.uint8)
        for y0, y1, d in strips:
            left[y0:y1] = wide[y0:y1, k * step * d:k * step * d + w]
            right[y0:y1] = wide[y0:y1, (k * step + 1) * d:(k * step + 1) * d + w]
        if noise > 0:
            left = np.clip(np.rint(left + rng.normal(0, noise, left.shape)), 0, 255).astype(np.uint8)
            right = np.clip(np.rint(right + rng.normal(0, noise, right.shape)), 0, 255).astype(np.uint8)
        frames.append((left, right))
    return frames, strips


def translation_frame_source(seed, max_pos, w=W, h=H, d_max=40, z_max=60.0, noise=1.0, noise_pool=16):
    """synth_translation_sequence, lazily and for ANY order of camera positions: get(pos, k) renders the stereo pair seen from
    `pos` baselines (0.54 m each) to the right of the start, 0 <= pos <= max_pos, with the sensor noise of frame k (drawn from a
    pool of `noise_pool` pre-generated noise fields: a 4 541-frame run spends its time in the loop under test, not in numpy's
    normal()).  A camera that comes back to a position sees the same scene under other noise -- what a revisit is."""
    rng = np.random.default_rng(0x5EED1000 + seed)
    wide = synth_texture(rng, w + (max_pos + 2) * d_max + 8, h)
    strips, y = [], 0
    while y < h:
        bh = int(rng.integers(24, 64))
        z = rng.uniform(KITTI["fx"] * KITTI["baseline"] / d_max, z_max)
        d = max(2, min(int(round(KITTI["fx"] * KITTI["baseline"] / z)), d_max))
        strips.append((y, min(h, y + bh), d))
        y += bh
    pool = [np.rint(np.random.default_rng(seed * 100003 + i).normal(0, noise, (2, h, w))).astype(np.int16) for i in range(noise_pool)] if noise > 0 else None

    def get(pos, k=0):
        assert 0 <= pos <= max_pos
        left = np.empty((h, w), np.uint8); right = np.empty((h, w), np.uint8)
        for y0, y1, d in strips:
            left[y0:y1] = wide[y0:y1, pos * d:pos * d + w]
            right[y0:y1] = wide[y0:y1, (pos + 1) * d:(pos + 1) * d + w]
        if pool is not None:
            nz = pool[k % noise_pool]
            left = np.clip(left.astype(np.int16) + nz[0], 0, 255).astype(np.uint8)
            right = np.clip(right.astype(np.int16) + nz[1], 0, 255).astype(np.uint8)
        return left, right
    get.strips = strips
    return get


def out_and_back_positions(n_frames, n_back):
    """Camera positions (baselines to the right of the start) of an n_frames run: out at one baseline per frame, then the last
    n_back frames back over the same places -- the revisit that closes a loop."""
    n_out = n_frames - n_back
    return [k if k < n_out else 2 * (n_out - 1) - k for k in range(n_frames)]


def synth_forward_sequence(seed, n_frames, step=0.5, w=W, h=H, noise=1.0, half_width=6.0, cam_height=1.65, z_far=120.0, px_per_m=48.0, yaw_per_frame=0.0):
    """n_frames rectified stereo pairs from a camera driving FORWARD (+z, `step` metres per frame, optionally turning by
    yaw_per_frame radians about the vertical) through a textured corridor: ground plane, two side walls, a far wall --
    KITTI-like geometry ("forward motion").  Every pixel is the texture of the first surface its ray
    hits (bilinear, three octaves so that structure exists at every distance).  Returns ([(left, right)], [camera poses]):
    pose = (R, t) of the left camera in the world (x right, y down, z forward)."""
    rng = np.random.default_rng(0x5EED2000 + seed)
    T = 2048
    tex = [synth_texture(rng, T, T).astype(np.float32) for _ in range(3)]     # ground, left wall, right wall (tiled)
    far = synth_texture(rng, T, T).astype(np.float32)
    fx, fy, cx, cy, b = KITTI["fx"], KITTI["fy"], KITTI["cx"], KITTI["cy"], KITTI["baseline"]
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d_cam = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], -1)       # ray directions in the camera frame

    def sample(t, a, c, foot):
        """octaves of the tiled texture t at plane coordinates (a, c) in metres; `foot` = metres per image pixel at the hit
        point: an octave whose texels are smaller than the pixel footprint fades out (it would alias into frame-to-frame noise)"""
        out = np.full(a.shape, 95.0, np.float32)
        for o, (sc, wt) in enumerate(((px_per_m, 0.8), (px_per_m / 3, 0.9), (px_per_m / 9, 1.0), (px_per_m / 27, 1.0))):
            fade = np.clip(1.0 / (sc * foot) - 1.0, 0.0, 1.0).astype(np.float32)
            if not fade.any():
                continue
            x = a * sc + 37.0 * o
            y = c * sc + 91.0 * o
            x0 = np.floor(x); y0 = np.floor(y)
            ax = (x - x0).astype(np.float32); ay = (y - y0).astype(np.float32)
            i0 = x0.astype(np.int64) % T; j0 = y0.astype(np.int64) % T
            i1 = (i0 + 1) % T; j1 = (j0 + 1) % T
            out += wt * fade * ((1 - ax) * (1 - ay) * t[j0, i0] + ax * (1 - ay) * t[j0, i1] + (1 - ax) * ay * t[j1, i0] + ax * ay * t[j1, i1] - 95.0)
        return out

    def render(R, c):
        d = d_cam @ R.T                                                          # rays in the world
        big = 1e9
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[..., 1] > 1e-9, (cam_height - c[1]) / d[..., 1], big)
            tl = np.where(d[..., 0] < -1e-9, (-half_width - c[0]) / d[..., 0], big)
            tr = np.where(d[..., 0] > 1e-9, (half_width - c[0]) / d[..., 0], big)
            tf = np.where(d[..., 2] > 1e-9, (z_far - c[2]) / d[..., 2], big)
        tg[tg <= 0] = big; tl[tl <= 0] = big; tr[tr <= 0] = big; tf[tf <= 0] = big
        t = np.minimum(np.minimum(tg, tl), np.minimum(tr, tf))
        P = c + t[..., None] * d
        foot = t / fx                                                            # metres per pixel at the surface (the obliquity of the ground is ignored)
        img = np.where(t == tg, sample(tex[0], P[..., 0], P[..., 2], foot),
                       np.where(t == tl, sample(tex[1], P[..., 2], P[..., 1], foot),
                                np.where(t == tr, sample(tex[2], P[..., 2], P[..., 1], foot), sample(far, P[..., 0], P[..., 1], foot))))
        return img

    frames, poses = [], []
    c = np.array([0.0, 0.0, 0.0])
    yaw = 0.0
    for k in range(n_frames):
        R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])   # camera-to-world, yaw about y (down)
        left = render(R, c)
        right = render(R, c + R @ np.array([b, 0.0, 0.0]))
        out = []
        for im in (left, right):
            im = im + rng.normal(0, noise, im.shape) if noise > 0 else im
            out.append(np.clip(np.rint(im), 0, 255).astype(np.uint8))
        frames.append(tuple(out)); poses.append((R.copy(), c.copy()))
        c = c + R @ np.array([0.0, 0.0, step])
        yaw += yaw_per_frame
    return frames, poses


KITTI_FRONTEND = dict(n_feats=500, fast_threshold=20, n_levels=1, nms_radius=2, orb_max_distance=60,
                      max_y_diff=2.0, min_disparity=1.0, max_disparity=128.0, harris_k=0.04)   # disparity 128 = 3 m at fx*b = 382


# change-in-pose filter parameters: stereo camera + [LEAST_SQUARES] (KITTI.ini:40-50) + residual_th (KITTI.ini:119)
KITTI_CIP = dict(fx_l=707.0912, fy_l=707.0912, cx_l=601.8873, cy_l=183.1104, fx_r=707.0912, fy_r=707.0912, cx_r=601.8873,
                 cy_r=183.1104, baseline=0.54, initial_max_iters=30, max_iters=30, max_incr_cost=3, use_robust_kernel=1,
                 residual_threshold=15.0, min_mod_out_vector=0.001, kernel_param=2.0, bad_tracking_th=5, reserved=0, residual_th=30.0)

# fundamental-matrix filter parameters: max_y_diff_epipolar / ransac_fit_prob (KITTI.ini:121-122), OpenCV's 2000 iterations and
# 15-point minimum; the 7-point solver normalises pixels about the image centre
KITTI_FM = dict(threshold=2.0, confidence=0.95, norm_cx=620.5, norm_cy=188.0, norm_scale=2.0 / 1241.0, seed=0x5EED0001,
                max_iters=2000, min_points=15)
