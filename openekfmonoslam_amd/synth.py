"""Deterministic synthetic monocular sequences for the EKF hot path.

Host-side workload generator (numpy only): a static 3-D point cloud, a constant-velocity camera, per-frame
keypoints + 32-byte binary descriptors (true projections with pixel noise and gross outliers, plus
distractors), and a map/covariance seeded with the reference's own new-feature initialisation
(EKF/AddMapFeature.cpp:109-344 semantics: rho0, sigma_rho, pixel sigma from the config) so P has the reference's
correlation structure.  Nothing here touches the GPU or the oracle.
"""
import math

import numpy as np

from .ekftypes import DESC_BYTES, FEATURE_INVERSE_DEPTH, KEYPOINT_DTYPE, s3_camera, s3_params


# ------------------------------------------------------------------------------------------------ camera model
def quat_to_rot(q):
    """Core/EKFMath.cpp:133-155 (R maps camera axes to world axes)."""
    r, x, y, z = q
    return np.array(
        [
            [r * r + x * x - y * y - z * z, 2 * (x * y - r * z), 2 * (z * x + r * y)],
            [2 * (x * y + r * z), r * r - x * x + y * y - z * z, 2 * (y * z - r * x)],
            [2 * (z * x - r * y), 2 * (y * z + r * x), r * r - x * x - y * y + z * z],
        ]
    )


def quat_mul(q1, q2):
    """Core/EKFMath.cpp:82-98."""
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    return np.array(
        [
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        ]
    )


def angles_to_quat(w):
    n = float(np.linalg.norm(w))
    if n < 2.22e-16:
        return np.array([1.0, 0.0, 0.0, 0.0])
    s = math.sin(n / 2)
    return np.array([math.cos(n / 2), s * w[0] / n, s * w[1] / n, s * w[2] / n])


def distort(cam, uv):
    """Undistorted pixel -> distorted pixel, vectorised (EKF/MeasurementPrediction.cpp:47-83)."""
    uv = np.asarray(uv, dtype=np.float64)
    pdx = uv[..., 0] - cam.cx
    pdy = uv[..., 1] - cam.cy
    mx, my = cam.dx * pdx, cam.dy * pdy
    d2 = mx * mx + my * my
    ru = np.sqrt(d2)
    rd = ru / (1.0 + cam.k1 * d2 + cam.k2 * d2 * d2)
    for _ in range(10):
        r2 = rd * rd
        f = rd + cam.k1 * r2 * rd + cam.k2 * r2 * r2 * rd - ru
        fp = 1 + 3 * cam.k1 * r2 + 5 * cam.k2 * r2 * r2
        rd = rd - f / fp
    d = 1.0 + cam.k1 * rd * rd + cam.k2 * rd**4
    return np.stack([cam.cx + pdx / d, cam.cy + pdy / d], axis=-1)


def undistort(cam, uv):
    """Distorted pixel -> undistorted pixel (EKF/AddMapFeature.cpp:42-58)."""
    uv = np.asarray(uv, dtype=np.float64)
    mx = uv[..., 0] - cam.cx
    my = uv[..., 1] - cam.cy
    dx, dy = cam.dx * mx, cam.dy * my
    rd = dx * dx + dy * dy
    dist = 1 + cam.k1 * rd + cam.k2 * rd * rd
    return np.stack([cam.cx + mx * dist, cam.cy + my * dist], axis=-1)


def project(cam, r, R, pts_world):
    """World XYZ -> (distorted pixel, camera-frame XYZ)."""
    h = (pts_world - r) @ R  # rows: R' (y - r)
    u = cam.cx + cam.fx * h[:, 0] / h[:, 2]
    v = cam.cy + cam.fy * h[:, 1] / h[:, 2]
    return distort(cam, np.stack([u, v], axis=-1)), h


# --------------------------------------------------------------------------------------- map seeding (host)
def _jac_quat_to_rot(q, a):
    """EKF/CommonFunctions.cpp:87-145 -> 3x4."""
    q0, qx, qy, qz = q
    mats = [
        np.array([[2 * q0, -2 * qz, 2 * qy], [2 * qz, 2 * q0, -2 * qx], [-2 * qy, 2 * qx, 2 * q0]]),
        np.array([[2 * qx, 2 * qy, 2 * qz], [2 * qy, -2 * qx, -2 * q0], [2 * qz, 2 * q0, -2 * qx]]),
        np.array([[-2 * qy, 2 * qx, 2 * q0], [2 * qx, 2 * qy, 2 * qz], [-2 * q0, 2 * qz, -2 * qy]]),
        np.array([[-2 * qz, -2 * q0, 2 * qx], [2 * q0, -2 * qz, 2 * qy], [2 * qx, 2 * qy, 2 * qz]]),
    ]
    return np.stack([m @ a for m in mats], axis=1)


def new_feature(cam, par, x13, uv):
    """Initial 6-vector, 6x7 and 6x3 Jacobians of a new inverse-depth feature seen at distorted pixel uv
    (EKF/AddMapFeature.cpp:109-216, 293-344)."""
    q = x13[3:7]
    R = quat_to_rot(q)
    up = undistort(cam, uv)
    xyz_c = np.array([-(cam.cx - up[0]) / cam.fx, -(cam.cy - up[1]) / cam.fy, 1.0])
    g = R @ xyz_c
    pos = np.array(
        [x13[0], x13[1], x13[2], math.atan2(g[0], g[2]), math.atan2(-g[1], math.hypot(g[0], g[2])), par.initInvDepthRho]
    )
    xw, yw, zw = g
    xxzz = xw * xw + zw * zw
    dth = np.array([zw / xxzz, 0.0, -xw / xxzz])
    sq = math.sqrt(xxzz)
    nsq = xxzz + yw * yw
    dph = np.array([xw * yw / (nsq * sq), -sq / nsq, zw * yw / (nsq * sq)])
    dgw_dq = _jac_quat_to_rot(q, xyz_c)
    Jpo = np.zeros((6, 7))
    Jpo[0, 0] = Jpo[1, 1] = Jpo[2, 2] = 1.0
    Jpo[3, 3:7] = dth @ dgw_dq
    Jpo[4, 3:7] = dph @ dgw_dq
    sub = np.stack([dth @ R, dph @ R])  # 2x3
    dgc_dhu = np.array([[1.0 / cam.fx, 0.0], [0.0, 1.0 / cam.fy], [0.0, 0.0]])
    ud, vd = uv
    xd, yd = (ud - cam.cx) * cam.dx, (vd - cam.cy) * cam.dy
    rd2 = xd * xd + yd * yd
    a = cam.k1 + 2.0 * cam.k2 * rd2
    b = 1.0 + cam.k1 * rd2 + cam.k2 * rd2 * rd2
    dx2, dy2 = 2.0 * cam.dx * cam.dx, 2.0 * cam.dy * cam.dy
    dhu_dhd = np.array(
        [
            [b + (ud - cam.cx) * a * ((ud - cam.cx) * dx2), (ud - cam.cx) * a * ((vd - cam.cy) * dy2)],
            [(vd - cam.cy) * a * ((ud - cam.cx) * dx2), (vd - cam.cy) * a * ((vd - cam.cy) * dy2) + b],
        ]
    )
    Jhr = np.zeros((6, 3))
    Jhr[3:5, 0:2] = sub @ dgc_dhu @ dhu_dhd
    Jhr[5, 2] = 1.0
    return pos, Jpo, Jhr


def seed_map(cam, par, x13, P13, uvs):
    """Append one inverse-depth feature per row of uvs to (x13, P13) exactly like repeated
    addFeatureToStateAndCovariance calls (EKF/AddMapFeature.cpp:221-289), without re-allocating P each time.
    Returns (feature_pos[N,6], P[n,n])."""
    N = len(uvs)
    n = 13 + 6 * N
    P = np.zeros((n, n))
    P[:13, :13] = P13
    noise = np.diag([cam.pixelErrorX**2, cam.pixelErrorY**2, par.inverseDepthRhoSD**2])
    pos = np.zeros((N, 6))
    for i, uv in enumerate(uvs):
        n0 = 13 + 6 * i
        pos[i], Jpo, Jhr = new_feature(cam, par, x13, uv)
        rows = Jpo @ P[0:7, 0:n0]
        P[n0 : n0 + 6, 0:n0] = rows
        P[0:n0, n0 : n0 + 6] = P[0:n0, 0:7] @ Jpo.T
        P[n0 : n0 + 6, n0 : n0 + 6] = rows[:, 0:7] @ Jpo.T + Jhr @ noise @ Jhr.T
    return pos, P


def initial_state_and_covariance(par):
    """initState / initCovariance (EKF/CommonFunctions.cpp:39-80)."""
    eps = 2.22e-16
    x = np.zeros(13)
    x[3] = 1.0
    x[10:13] = eps
    P = np.zeros((13, 13))
    for i in range(7):
        P[i, i] = eps
    for i in range(3):
        P[7 + i, 7 + i] = par.initLinearAccelSD**2
        P[10 + i, 10 + i] = par.initAngularAccelSD**2
    return x, P


# ------------------------------------------------------------------------------------------------ the sequence
class SyntheticSequence:
    """N static points watched by a camera moving with constant linear and angular velocity.

    Attributes
    ----------
    cam, par : EkfCamera, EkfParams
    x13      : initial camera state (true velocities, so the filter tracks from the first frame)
    feature_pos [N,6], feature_type [N], feature_desc [N,32], P0 [n,n] : the seeded map
    frames   : list of (keypoints[K] KEYPOINT_DTYPE, descriptors[K,32] uint8), one per step
    truth_r, truth_q : camera trajectory (frame 0 = map seeding frame)
    """

    def __init__(self, n_features, n_frames, width=640, height=480, seed=None, pixel_sigma=0.5,
                 outlier_fraction=0.05, distractors_per_feature=1.0, max_bit_flips=20,
                 v=(0.01, 0.0, 0.002), w=(0.0, 0.002, 0.0), depth_range=(2.0, 10.0), margin=12.0, horizon=100):
        self.cam = s3_camera(width, height)
        self.par = s3_params()
        self.n_features = int(n_features)
        self.n_frames = int(n_frames)
        seed = (0x5EED0000 + self.n_features) if seed is None else seed
        rng = np.random.Generator(np.random.PCG64(seed))
        cam = self.cam
        v = np.asarray(v, dtype=np.float64)
        w = np.asarray(w, dtype=np.float64)

        # trajectory, integrated exactly like predictState (EKF/StateAndCovariancePrediction.cpp:43-65).
        # The scene (points, map, and the first k frames) must not depend on how many frames are asked for: the
        # visibility check below runs over a fixed horizon (a multiple of `horizon` frames that covers the run), so
        # SyntheticSequence(N, 25).frames == SyntheticSequence(N, 70).frames[:25].
        T = self.n_frames
        TH = horizon * max(1, -(-T // horizon))
        r = np.zeros((TH + 1, 3))
        q = np.zeros((TH + 1, 4))
        q[0] = [1, 0, 0, 0]
        dq = angles_to_quat(w)
        for t in range(TH):
            r[t + 1] = r[t] + v
            q[t + 1] = quat_mul(q[t], dq)
        Rs = [quat_to_rot(q[t]) for t in range(TH + 1)]
        self.truth_r, self.truth_q = r[: T + 1], q[: T + 1]

        # points: rejection-sample so every point stays inside the frame (with a margin) over the whole horizon
        pts = np.zeros((0, 3))
        check = sorted(set([0, TH // 4, TH // 2, (3 * TH) // 4, TH]))
        while len(pts) < self.n_features:
            nb = max(256, 2 * (self.n_features - len(pts)))
            uv = np.stack([rng.uniform(margin, width - margin, nb), rng.uniform(margin, height - margin, nb)], -1)
            depth = rng.uniform(depth_range[0], depth_range[1], nb)
            cand = np.stack([(uv[:, 0] - cam.cx) / cam.fx * depth, (uv[:, 1] - cam.cy) / cam.fy * depth, depth], -1)
            ok = np.ones(nb, dtype=bool)
            for t in check:
                p, h = project(cam, r[t], Rs[t], cand)
                ok &= (h[:, 2] > 0.5) & (p[:, 0] > margin) & (p[:, 0] < width - margin)
                ok &= (p[:, 1] > margin) & (p[:, 1] < height - margin)
            pts = np.concatenate([pts, cand[ok]])
        self.points = pts[: self.n_features]
        N = self.n_features

        # descriptors
        self.feature_desc = rng.integers(0, 256, (N, DESC_BYTES), dtype=np.uint8)
        self.feature_type = np.full(N, FEATURE_INVERSE_DEPTH, dtype=np.int32)

        # map seeding from frame 0 (measured = true projection + pixel noise)
        x0, P13 = initial_state_and_covariance(self.par)
        x0[7:10] = v
        x0[10:13] = np.where(np.abs(w) > 0, w, 2.22e-16)
        uv0, _ = project(cam, r[0], Rs[0], self.points)
        uv0 = uv0 + rng.normal(0.0, pixel_sigma, uv0.shape)
        self.x13 = x0
        self.feature_pos, self.P0 = seed_map(cam, self.par, x0, P13, uv0)

        # per-frame keypoints
        n_dis = int(round(distractors_per_feature * N))
        self.frames = []
        for t in range(1, T + 1):
            uv, _ = project(cam, r[t], Rs[t], self.points)
            uv = uv + rng.normal(0.0, pixel_sigma, uv.shape)
            gross = rng.random(N) < outlier_fraction
            ang = rng.uniform(0, 2 * np.pi, N)
            mag = rng.uniform(5.0, 15.0, N)
            uv[gross] += np.stack([mag * np.cos(ang), mag * np.sin(ang)], -1)[gross]
            desc = self.feature_desc.copy()
            nflip = rng.integers(0, max_bit_flips + 1, N)
            bits = rng.integers(0, DESC_BYTES * 8, (N, max_bit_flips))
            for k in range(max_bit_flips):
                sel = nflip > k
                np.bitwise_xor.at(desc, (np.nonzero(sel)[0], bits[sel, k] // 8),
                                  (1 << (bits[sel, k] % 8)).astype(np.uint8))
            duv = np.stack([rng.uniform(1, width - 1, n_dis), rng.uniform(1, height - 1, n_dis)], -1)
            ddesc = rng.integers(0, 256, (n_dis, DESC_BYTES), dtype=np.uint8)
            all_uv = np.concatenate([uv, duv])
            all_desc = np.concatenate([desc, ddesc])
            perm = rng.permutation(len(all_uv))
            kps = np.zeros(len(all_uv), dtype=KEYPOINT_DTYPE)
            kps["x"] = all_uv[perm, 0].astype(np.float32)
            kps["y"] = all_uv[perm, 1].astype(np.float32)
            self.frames.append((kps, np.ascontiguousarray(all_desc[perm])))

    @property
    def state_dim(self):
        return 13 + 6 * self.n_features

    # ---- images for matcher mode B (NCC): every point carries a fixed smooth texture patch, pasted at its rounded
    # projection over a noisy background; frame 0 is the map-seeding frame the templates are cut from
    def _textures(self, patch):
        if getattr(self, "_tex", None) is not None and self._tex.shape[1] == patch:
            return self._tex
        rng = np.random.Generator(np.random.PCG64(0x7E87 + self.n_features))
        N = self.n_features
        g = rng.uniform(30.0, 225.0, (N, 5, 5))
        xs = np.linspace(0.0, 4.0, patch)
        i0 = np.minimum(xs.astype(int), 3)
        fr = xs - i0
        rows = g[:, i0, :] * (1 - fr)[None, :, None] + g[:, i0 + 1, :] * fr[None, :, None]
        tex = rows[:, :, i0] * (1 - fr)[None, None, :] + rows[:, :, i0 + 1] * fr[None, None, :]
        tex = tex + rng.normal(0.0, 8.0, tex.shape)
        self._tex = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
        return self._tex

    def pixel_positions(self, t):
        """Integer pixel (x, y) each point's patch is centred on in frame t (t = 0: seeding frame)."""
        R = quat_to_rot(self.truth_q[t])
        uv, _ = project(self.cam, self.truth_r[t], R, self.points)
        return np.rint(uv).astype(np.int64)

    def render_image(self, t, patch=17, noise_sigma=2.0, outlier_fraction=0.05, channels=1):
        """uint8 frame t: [H, W] (channels=1), [H, W, 3] BGR or [H, W, 4] RGBA."""
        W, H = self.cam.pixelsX, self.cam.pixelsY
        rng = np.random.Generator(np.random.PCG64(0x1A6E0000 + 977 * self.n_features + t))
        img = rng.normal(118.0, 6.0, (H, W))
        tex = self._textures(patch)
        px = self.pixel_positions(t)
        if t > 0 and outlier_fraction > 0:
            gross = rng.random(self.n_features) < outlier_fraction
            ang = rng.uniform(0, 2 * np.pi, self.n_features)
            mag = rng.uniform(5.0, 12.0, self.n_features)
            off = np.rint(np.stack([mag * np.cos(ang), mag * np.sin(ang)], -1)).astype(np.int64)
            px = px + off * gross[:, None]
        hp = patch // 2
        for i in range(self.n_features):
            x, y = int(px[i, 0]), int(px[i, 1])
            x0, x1, y0, y1 = max(x - hp, 0), min(x + hp + 1, W), max(y - hp, 0), min(y + hp + 1, H)
            if x0 >= x1 or y0 >= y1:
                continue
            img[y0:y1, x0:x1] = tex[i, y0 - (y - hp):y1 - (y - hp), x0 - (x - hp):x1 - (x - hp)]
        img = img + rng.normal(0.0, noise_sigma, img.shape)
        gray = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if channels == 1:
            return gray
        out = np.zeros((H, W, channels), dtype=np.uint8)
        # a colour cast per channel so the gray conversion is exercised (weights 77/150/29)
        r = np.clip(gray.astype(np.int32) + 9, 0, 255)
        g = gray.astype(np.int32)
        b = np.clip(gray.astype(np.int32) - 14, 0, 255)
        if channels == 3:
            out[..., 0], out[..., 1], out[..., 2] = b, g, r
        else:
            out[..., 0], out[..., 1], out[..., 2], out[..., 3] = r, g, b, 255
        return out
