"""Deterministic synthetic workloads; numpy only, no GPU, no oracle.

* frames: textured-gradient 8-bit images that yield ~2k ORB keypoints at 640x480 with the default
  parameters, as a translating sequence (frame t+1 = frame t shifted by (3,1) px + fresh noise) so
  that consecutive frames produce realistic match sets.
* local-BA scenes: cameras on an arc looking at a box of points, EuRoC intrinsics
  (reference example/euroc/EuRoC_mono.yaml:8-11), octave-dependent pixel noise, a few outliers.

EuRoC / KITTI imagery is not available offline; BASELINE.json configs 2 and 4 run on these
sequences at 752x480 / 1241x376.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


class XorShift64Star:
    """xorshift64* (Vigna 2014) -- tiny, portable, seedable."""

    def __init__(self, seed: int):
        self.s = (seed & _M64) or 0x9E3779B97F4A7C15

    def next(self) -> int:
        x = self.s
        x ^= x >> 12
        x ^= (x << 25) & _M64
        x ^= x >> 27
        self.s = x
        return (x * 0x2545F4914F6CDD1D) & _M64

    def randint(self, lo: int, hi: int) -> int:
        """uniform integer in [lo, hi]"""
        return lo + self.next() % (hi - lo + 1)

    def uniform(self) -> float:
        return (self.next() >> 11) * (1.0 / (1 << 53))


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _canvas(width: int, height: int, seed: int, rects_per_mpx: float = 4883.0) -> np.ndarray:
    """gradient base ((x+2y)>>2)&255 plus axis-aligned filled rectangles (4..40 px, random grey)."""
    ys, xs = np.mgrid[0:height, 0:width]
    img = (((xs + 2 * ys) >> 2) & 255).astype(np.uint8)
    rng = XorShift64Star(seed)
    n_rect = int(round(rects_per_mpx * width * height / 1e6))  # 1500 at 640x480
    for _ in range(n_rect):
        rw, rh = rng.randint(4, 40), rng.randint(4, 40)
        x0, y0 = rng.randint(0, width - 1), rng.randint(0, height - 1)
        g = rng.randint(0, 255)
        img[y0:y0 + rh, x0:x0 + rw] = g
    return img


def frame_sequence(n_frames: int, width: int = 640, height: int = 480, seed: int = 0x5EED,
                   shift: tuple[int, int] = (3, 1), noise: int = 3) -> np.ndarray:
    """(n_frames, height, width) uint8.  Frame t is the canvas window at offset t*shift plus
    uniform integer noise in [-noise, +noise] (fresh per frame)."""
    sx, sy = shift
    cw, ch = width + sx * (n_frames - 1), height + sy * (n_frames - 1)
    canvas = _canvas(cw, ch, seed).astype(np.int16)
    out = np.empty((n_frames, height, width), np.uint8)
    idx = np.arange(width * height, dtype=np.uint64).reshape(height, width)
    for t in range(n_frames):
        win = canvas[t * sy:t * sy + height, t * sx:t * sx + width]
        h = _splitmix64(idx + np.uint64(((seed + 1) * 0x10001 + t) * (width * height)))
        nz = (h % np.uint64(2 * noise + 1)).astype(np.int16) - noise
        out[t] = np.clip(win + nz, 0, 255).astype(np.uint8)
    return out


def frame(width: int = 640, height: int = 480, seed: int = 0x5EED) -> np.ndarray:
    return frame_sequence(1, width, height, seed)[0]


# --------------------------------------------------------------------------------------------- BA

def _rodrigues(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def ba_scene(num_kf: int = 20, num_lm: int = 10000, obs_per_lm: int = 6, num_fixed: int = 4, seed: int = 1234,
             outlier_frac: float = 0.03, pose_noise=(0.02, 0.5), point_noise: float = 0.03, stereo: bool = False,
             scale_factor: float = 1.2, num_levels: int = 8, loop: bool = False, equirect: bool = False) -> dict:
    """Synthetic local-BA problem on the flat arrays the C-ABI carries.

    Cameras on an arc (radius 5 m, 18 deg span; `loop=True`: full circle for the global-BA config) looking
    at points uniform in a 4x3x2 m box 4-8 m away.  Each point is observed by `obs_per_lm` cameras that see
    it inside a 752x480 image.  Returns ground truth next to the perturbed initial estimate.
    `equirect=True`: the cameras are 1920x960 equirectangular (camera/equirectangular.h: u = cols (0.5 + atan2(x, z) / 2 pi),
    v = rows (0.5 + asin(y / |p|) / pi)); a point is visible from every camera; intrinsics rows are {0, 0, cols, rows, 0}.
    """
    rng = np.random.default_rng(seed)
    fx = fy = 458.654
    cx, cy = 367.215, 248.375
    W, H = 752, 480
    span = 2 * np.pi if loop else np.deg2rad(18.0)
    radius = 5.0 if not loop else 12.0
    Rs, ts = [], []
    for i in range(num_kf):
        a = -span / 2 + span * (i + 0.5) / num_kf
        c = np.array([radius * np.sin(a), 0.05 * np.sin(3 * a), -radius * np.cos(a)])  # camera centre
        z = -c / np.linalg.norm(c) if not loop else np.array([np.sin(a), 0, -np.cos(a)])
        if loop:
            z = np.array([np.sin(a), 0.0, -np.cos(a)])  # look outward along the radius
        x = np.cross(np.array([0.0, 1.0, 0.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R_wc = np.stack([x, y, z], axis=1)
        R_cw = R_wc.T
        Rs.append(R_cw)
        ts.append(-R_cw @ c)
    Rs, ts = np.array(Rs), np.array(ts)

    pts = np.empty((num_lm, 3))
    obs_pose, obs_point, obs_uvr, obs_oct = [], [], [], []
    fxb = fx * 0.11 if stereo else 0.0
    made = 0
    tries = 0
    while made < num_lm:
        tries += 1
        if loop:
            a = rng.uniform(0, 2 * np.pi)
            r = radius + rng.uniform(4.0, 8.0)
            p = np.array([r * np.sin(a) + rng.uniform(-1, 1), rng.uniform(-1.5, 1.5), -r * np.cos(a) + rng.uniform(-1, 1)])
        else:
            p = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(-1, 1)])
        pc = Rs @ p + ts
        ok = pc[:, 2] > 0.5
        u = fx * pc[:, 0] / np.where(ok, pc[:, 2], 1) + cx
        v = fy * pc[:, 1] / np.where(ok, pc[:, 2], 1) + cy
        ok &= (u > 20) & (u < W - 20) & (v > 20) & (v < H - 20)
        if equirect:
            ok = np.linalg.norm(pc, axis=1) > 0.5
            u = 1920.0 * (0.5 + np.arctan2(pc[:, 0], pc[:, 2]) / (2 * np.pi))
            v = 960.0 * (0.5 + np.arcsin(pc[:, 1] / np.linalg.norm(pc, axis=1)) / np.pi)
            ok &= (u > 60) & (u < 1920 - 60)  # keep clear of the +-pi seam (the reference's error is a plain difference)
        vis = np.flatnonzero(ok)
        if len(vis) < 2:
            if tries > 50 * num_lm:
                raise RuntimeError("scene generator cannot place points")
            continue
        k = min(obs_per_lm, len(vis))
        # prefer a contiguous run of cameras (covisibility is local in a real map)
        start = rng.integers(0, len(vis) - k + 1)
        cams = vis[start:start + k]
        pts[made] = p
        for c_ in cams:
            octv = int(rng.integers(0, num_levels))
            sig = scale_factor ** octv
            uu = u[c_] + rng.normal(0, 1.0) * sig
            vv = v[c_] + rng.normal(0, 1.0) * sig
            if rng.uniform() < outlier_frac:
                d = rng.uniform(20, 50)
                ang = rng.uniform(0, 2 * np.pi)
                uu += d * np.cos(ang)
                vv += d * np.sin(ang)
            ur = (uu - fxb / pc[c_, 2] + rng.normal(0, 1.0) * sig) if stereo else -1.0
            obs_pose.append(c_)
            obs_point.append(made)
            obs_uvr.append((uu, vv, ur))
            obs_oct.append(octv)
        made += 1

    # perturbed initial estimate
    pose_gt = np.zeros((num_kf, 12))
    pose_init = np.zeros((num_kf, 12))
    fixed = np.zeros(num_kf, np.uint8)
    fixed[:num_fixed] = 1
    for i in range(num_kf):
        pose_gt[i].reshape(3, 4)[:, :3] = Rs[i]
        pose_gt[i].reshape(3, 4)[:, 3] = ts[i]
        if fixed[i]:
            pose_init[i] = pose_gt[i]
            continue
        dw = rng.normal(0, 1, 3)
        dw *= np.deg2rad(pose_noise[1]) / np.linalg.norm(dw)
        dR = _rodrigues(dw)
        c = -Rs[i].T @ ts[i] + rng.normal(0, pose_noise[0] / np.sqrt(3), 3)
        Rn = dR @ Rs[i]
        pose_init[i].reshape(3, 4)[:, :3] = Rn
        pose_init[i].reshape(3, 4)[:, 3] = -Rn @ c
    pts_init = pts + rng.normal(0, point_noise / np.sqrt(3), pts.shape)

    obs_oct = np.array(obs_oct)
    inv_sigma_sq = np.ones(len(obs_oct), np.float32)
    s = np.float32(1.0)
    table = [np.float32(1.0)]
    for _ in range(1, num_levels):  # orb_params.cc:63-71 recurrence in fp32
        s = np.float32(scale_factor) * s
        table.append(np.float32(1.0) / (s * s))
    inv_sigma_sq = np.array(table, np.float32)[obs_oct]
    huber = np.full(len(obs_oct), np.sqrt(np.float32(7.81473 if stereo else 5.99146)), np.float32)
    intr = np.tile(np.array([0.0, 0.0, 1920.0, 960.0, 0.0] if equirect else [fx, fy, cx, cy, fxb]), (num_kf, 1))
    return dict(
        pose_cw=pose_init, pose_gt=pose_gt, pose_fixed=fixed, points=pts_init, points_gt=pts,
        obs_pose=np.array(obs_pose, np.int32), obs_point=np.array(obs_point, np.int32),
        obs_uvr=np.array(obs_uvr, np.float32), obs_inv_sigma_sq=inv_sigma_sq, obs_huber=huber, intr=intr,
    )


def ba_scene_large(num_kf: int = 500, num_lm: int = 200000, obs_per_lm: int = 6, num_fixed: int = 1, seed: int = 5005,
                   outlier_frac: float = 0.03, pose_noise=(0.02, 0.5), point_noise: float = 0.03, scale_factor: float = 1.2,
                   num_levels: int = 8) -> dict:
    """BASELINE config 5 (global BA): `num_kf` keyframes on a closed loop (radius 12 m, looking outward), `num_lm` landmarks in a
    ring 4-8 m outside it, every landmark observed by a contiguous run of `obs_per_lm` of the cameras that see it inside a
    752x480 image (covisibility is local, and the run may wrap around camera 0: the loop closure).  Same model, noise and
    outputs as `ba_scene(loop=True)`, generated with array operations (the per-landmark Python loop of `ba_scene` needs minutes
    at 200 k landmarks); its random stream differs, so the two generators do not produce the same scene for the same seed."""
    rng = np.random.default_rng(seed)
    fx = fy = 458.654
    cx, cy = 367.215, 248.375
    W, H = 752, 480
    radius = 12.0
    ang = -np.pi + 2 * np.pi * (np.arange(num_kf) + 0.5) / num_kf
    c = np.stack([radius * np.sin(ang), 0.05 * np.sin(3 * ang), -radius * np.cos(ang)], 1)
    z = np.stack([np.sin(ang), np.zeros(num_kf), -np.cos(ang)], 1)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    Rs = np.stack([x, y, z], axis=1)                      # R_cw rows = camera axes in world coordinates
    ts = -np.einsum("kij,kj->ki", Rs, c)

    pts = np.empty((num_lm, 3))
    cams = np.empty((num_lm, obs_per_lm), np.int64)
    made = 0
    half = np.arctan((W / 2 - 20) / fx)                   # cameras whose optical axis is within the half-FOV see the point
    span = max(int(half / (2 * np.pi / num_kf)), obs_per_lm)
    while made < num_lm:
        m = int((num_lm - made) * 1.25) + 64
        a = rng.uniform(-np.pi, np.pi, m)
        r = radius + rng.uniform(4.0, 8.0, m)
        p = np.stack([r * np.sin(a) + rng.uniform(-1, 1, m), rng.uniform(-1.5, 1.5, m), -r * np.cos(a) + rng.uniform(-1, 1, m)], 1)
        centre = np.floor((np.arctan2(p[:, 0], -p[:, 2]) + np.pi) / (2 * np.pi) * num_kf).astype(np.int64)
        # a run of obs_per_lm consecutive cameras starting anywhere in the window of cameras around the point's bearing; runs with a
        # camera that does not see the point inside the image are drawn again
        start = centre - span + (rng.uniform(0, 1, m) * (2 * span - obs_per_lm + 2)).astype(np.int64)
        cand = (start[:, None] + np.arange(obs_per_lm)[None, :]) % num_kf
        pc = np.einsum("mkij,mj->mki", Rs[cand], p) + ts[cand]
        ok = pc[..., 2] > 0.5
        zz = np.where(ok, pc[..., 2], 1.0)
        u = fx * pc[..., 0] / zz + cx
        v = fy * pc[..., 1] / zz + cy
        ok &= (u > 20) & (u < W - 20) & (v > 20) & (v < H - 20)
        idx = np.flatnonzero(ok.all(1))[: num_lm - made]
        pts[made:made + len(idx)] = p[idx]
        cams[made:made + len(idx)] = cand[idx]
        made += len(idx)

    obs_point = np.repeat(np.arange(num_lm, dtype=np.int32), obs_per_lm)
    obs_pose = cams.reshape(-1).astype(np.int32)
    E = len(obs_pose)
    pc = np.einsum("eij,ej->ei", Rs[obs_pose], pts[obs_point]) + ts[obs_pose]
    octv = rng.integers(0, num_levels, E)
    sig = scale_factor ** octv
    uu = fx * pc[:, 0] / pc[:, 2] + cx + rng.normal(0, 1.0, E) * sig
    vv = fy * pc[:, 1] / pc[:, 2] + cy + rng.normal(0, 1.0, E) * sig
    out = rng.uniform(0, 1, E) < outlier_frac
    d = rng.uniform(20, 50, E)
    th = rng.uniform(0, 2 * np.pi, E)
    uu = uu + np.where(out, d * np.cos(th), 0.0)
    vv = vv + np.where(out, d * np.sin(th), 0.0)
    obs_uvr = np.stack([uu, vv, np.full(E, -1.0)], 1).astype(np.float32)

    pose_gt = np.concatenate([Rs, ts[:, :, None]], axis=2).reshape(num_kf, 12)
    fixed = np.zeros(num_kf, np.uint8)
    fixed[:num_fixed] = 1
    dw = rng.normal(0, 1, (num_kf, 3))
    dw *= np.deg2rad(pose_noise[1]) / np.linalg.norm(dw, axis=1, keepdims=True)
    cn = c + rng.normal(0, pose_noise[0] / np.sqrt(3), (num_kf, 3))
    pose_init = pose_gt.copy()
    for i in range(num_kf):
        if fixed[i]:
            continue
        Rn = _rodrigues(dw[i]) @ Rs[i]
        pose_init[i].reshape(3, 4)[:, :3] = Rn
        pose_init[i].reshape(3, 4)[:, 3] = -Rn @ cn[i]
    pts_init = pts + rng.normal(0, point_noise / np.sqrt(3), pts.shape)

    sfac = np.float32(1.0)
    table = [np.float32(1.0)]
    for _ in range(1, num_levels):  # orb_params.cc:63-71 recurrence in fp32
        sfac = np.float32(scale_factor) * sfac
        table.append(np.float32(1.0) / (sfac * sfac))
    inv_sigma_sq = np.array(table, np.float32)[octv]
    huber = np.full(E, np.sqrt(np.float32(5.99146)), np.float32)
    intr = np.tile(np.array([fx, fy, cx, cy, 0.0]), (num_kf, 1))
    return dict(pose_cw=pose_init, pose_gt=pose_gt, pose_fixed=fixed, points=pts_init, points_gt=pts, obs_pose=obs_pose,
                obs_point=obs_point, obs_uvr=obs_uvr, obs_inv_sigma_sq=inv_sigma_sq, obs_huber=huber, intr=intr)


# --------------------------------------------------------------------------------------------- map scenes for the matchers

def _look_at(center, target):
    z = target - center
    z /= np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R_cw = np.stack([x, y, z], 0)
    return R_cw, -R_cw @ center


def orb_tables(scale_factor: float = 1.2, num_levels: int = 8):
    """orb_params tables by the reference's fp32 recurrences (feature/orb_params.cc:41-71)."""
    sf = np.ones(num_levels, np.float32)
    for l in range(1, num_levels):
        sf[l] = np.float32(scale_factor) * sf[l - 1]
    inv_sf = np.float32(1.0) / sf
    sigma_sq = sf * sf
    return dict(scale_factors=sf, inv_scale_factors=inv_sf, level_sigma_sq=sigma_sq, inv_level_sigma_sq=np.float32(1.0) / sigma_sq,
                log_scale_factor=float(np.log(np.float32(scale_factor))), num_levels=num_levels)


def map_scene(seed: int = 7, n_lm: int = 1500, n_extra: int = 600, stereo: bool = False, width: int = 752, height: int = 480,
              baseline_shift: float = 0.35, pixel_noise: float = 1.0, flip_bits: int = 36, obs_prob: float = 0.8, num_levels: int = 8) -> dict:
    """Two views of one set of landmarks, flattened the way the matcher entry points take them (SURVEY 8(d) has no matcher
    workload of its own; this is the smallest map that exercises every gate of match::*):
      * cameras 1 and 2 (EuRoC pinhole intrinsics, no distortion) `baseline_shift` metres apart, looking at a slab of points
      * every landmark has a random 256-bit descriptor; a view that observes it (probability `obs_prob`) gets a keypoint at the
        projection + N(0, pixel_noise) px, an octave around the level its depth predicts, an angle that differs by a few degrees
        between the views, and the landmark's descriptor with up to `flip_bits` random bit flips
      * `n_extra` clutter keypoints per view (random position / descriptor / octave)
    Returns dict(views=[v1, v2], landmarks=..., tables=orb tables, K=(fx, fy, cx, cy, fxb)); a view holds rot_cw, trans_cw, xy (n x 2
    f32 undistorted), octave, angle, desc, x_right (stereo) and lm (index of the landmark its keypoint observes or -1)."""
    rng = np.random.default_rng(seed)
    T = orb_tables(1.2, num_levels)
    fx = fy = 458.654
    cx, cy = 367.215, 248.375
    fxb = fx * 0.11 if stereo else 0.0
    pts = np.stack([rng.uniform(-3.2, 3.2, n_lm), rng.uniform(-2.0, 2.0, n_lm), rng.uniform(4.0, 9.0, n_lm)], 1)
    lm_desc = rng.integers(0, 256, (n_lm, 32), dtype=np.uint8)
    centers = [np.array([0.0, 0.0, 0.0]), np.array([baseline_shift, 0.03, 0.06])]
    target = np.array([0.0, 0.0, 6.5])
    views = []
    base_angle = rng.uniform(0, 360, n_lm).astype(np.float32)
    ref_dist = np.linalg.norm(pts - centers[0], axis=1)
    ref_oct = rng.integers(0, num_levels - 2, n_lm)
    max_valid = (ref_dist * T["scale_factors"][ref_oct]).astype(np.float32)
    min_valid = (max_valid * T["inv_scale_factors"][num_levels - 1]).astype(np.float32)
    for v in range(2):
        R, t = _look_at(centers[v], target)
        pc = pts @ R.T + t
        u = fx * pc[:, 0] / pc[:, 2] + cx
        w = fy * pc[:, 1] / pc[:, 2] + cy
        inside = (pc[:, 2] > 0.1) & (u > 5) & (u < width - 5) & (w > 5) & (w < height - 5)
        seen = inside & (rng.uniform(0, 1, n_lm) < obs_prob)
        ids = np.flatnonzero(seen)
        dist = np.linalg.norm(pts[ids] - centers[v], axis=1)
        pred = np.ceil(np.log(max_valid[ids] / dist) / T["log_scale_factor"]).astype(int)
        octave = np.clip(pred + rng.integers(-1, 2, len(ids)), 0, num_levels - 1)
        xy = np.stack([u[ids], w[ids]], 1) + rng.normal(0, pixel_noise, (len(ids), 2)) * T["scale_factors"][octave][:, None]
        angle = np.mod(base_angle[ids] + rng.normal(0, 6.0, len(ids)), 360.0)
        desc = lm_desc[ids].copy()
        nflip = rng.integers(0, flip_bits + 1, len(ids))
        for k in range(len(ids)):
            bits = rng.choice(256, nflip[k], replace=False)
            np.bitwise_xor.at(desc[k], bits >> 3, (1 << (bits & 7)).astype(np.uint8))
        xr = (xy[:, 0] - fxb / pc[ids, 2] + rng.normal(0, 0.5, len(ids))) if stereo else np.full(len(ids), -1.0)
        if stereo:
            xr[rng.uniform(0, 1, len(ids)) < 0.25] = -1.0   # keypoints without a stereo match
        # clutter
        exy = np.stack([rng.uniform(0, width, n_extra), rng.uniform(0, height, n_extra)], 1)
        all_xy = np.concatenate([xy, exy]).astype(np.float32)
        all_oct = np.concatenate([octave, rng.integers(0, num_levels, n_extra)]).astype(np.int32)
        all_ang = np.concatenate([angle, rng.uniform(0, 360, n_extra)]).astype(np.float32)
        all_desc = np.concatenate([desc, rng.integers(0, 256, (n_extra, 32), dtype=np.uint8)])
        all_xr = np.concatenate([xr, np.full(n_extra, -1.0)]).astype(np.float32)
        all_lm = np.concatenate([ids, np.full(n_extra, -1)]).astype(np.int64)
        perm = rng.permutation(len(all_xy))   # keypoint order is unrelated to landmark order
        views.append(dict(rot_cw=R, trans_cw=t, center=centers[v], xy=all_xy[perm], octave=all_oct[perm], angle=all_ang[perm],
                          desc=np.ascontiguousarray(all_desc[perm]), x_right=all_xr[perm], lm=all_lm[perm]))
    normal = pts - centers[0]
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    return dict(views=views, tables=T, K=(fx, fy, cx, cy, fxb), width=width, height=height,
                landmarks=dict(pos_w=pts, desc=lm_desc, min_valid_dist=min_valid, max_valid_dist=max_valid, mean_normal=normal))
