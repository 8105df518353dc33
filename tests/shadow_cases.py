"""Seeded inputs for the cascade-fit tests and the script that makes their
golden vectors: random cameras, lights and configurations (as ctypes structs of
whatever binding module is passed in - the oracle's and the library's structs
have the same layout)."""
import numpy as np


def random_basis(rng):
    """Orthonormal right-handed camera basis (forward, right, up), float32."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    right, up, back = r[:, 0], r[:, 1], r[:, 2]
    return (-back).astype(np.float32), right.astype(np.float32), up.astype(np.float32)


def random_camera(rng, make_camera):
    forward, right, up = random_basis(rng)
    position = rng.uniform(-300.0, 300.0, 3).astype(np.float32)
    near = float(np.float32(rng.choice([0.05, 0.1, 0.5, 1.0])))
    far = float(np.float32(rng.choice([60.0, 200.0, 500.0, 1500.0])))
    window = [(1280, 720), (1920, 1080), (800, 800), (3440, 1440)][int(rng.integers(0, 4))]
    if rng.random() < 0.15:
        half = float(np.float32(rng.uniform(5.0, 80.0)))
        return make_camera(position, forward, right, up, near, far, 45.0, window,
                           ortho=(-half, half, -half * 0.6, half * 0.6))
    zoom = float(np.float32(rng.uniform(30.0, 100.0)))
    return make_camera(position, forward, right, up, near, far, zoom, window)


def random_config(rng, base):
    """A variation of the default preset (`base`: a fresh config struct holding it)."""
    cfg = base
    cfg.cascade_count = int(rng.integers(1, 9))
    cfg.shadow_map_size = int(rng.choice([512, 1024, 2048, 4096]))
    cfg.cascade_split_lambda = float(np.float32(rng.uniform(0.0, 1.0)))
    cfg.max_shadow_distance = float(np.float32(rng.choice([0.0, 50.0, 120.0, 200.0])))
    cfg.cascade_guard_band_texels = float(np.float32(rng.choice([0.0, 32.0, 128.0])))
    cfg.z_extension_factor = float(np.float32(rng.choice([0.0, 4.0, 5.0])))
    if rng.random() < 0.5:  # per-cascade values off: the profile ramps apply
        for i in range(8):
            cfg.cascade_guard_band_texels_per[i] = 0.0
            cfg.cascade_z_extension_factor_per[i] = 0.0
    cfg.use_constant_cascade_size = int(rng.integers(0, 2))
    cfg.anchor_snap_texels = float(np.float32(rng.choice([0.0, 8.0, 16.0])))
    cfg.stabilize_cascades = int(rng.integers(0, 2))
    cfg.use_scene_bounds = int(rng.random() < 0.4)
    centre = rng.uniform(-200.0, 200.0, 3)
    half = rng.uniform(5.0, 400.0, 3)
    for k in range(3):
        cfg.scene_min[k] = float(np.float32(centre[k] - half[k]))
        cfg.scene_max[k] = float(np.float32(centre[k] + half[k]))
    return cfg


def random_light(rng):
    d = rng.normal(size=3)
    if rng.random() < 0.1:
        d = np.array([0.001, -1.0, 0.002])  # nearly straight down: the other up reference
    return (d / np.linalg.norm(d)).astype(np.float32)


def random_fit(rng, fit_type, like=None):
    f = fit_type()
    if like is None:
        f.center_x, f.center_y = (float(np.float32(v)) for v in rng.uniform(-500, 500, 2))
        f.extent = float(np.float32(rng.uniform(1.0, 600.0)))
        f.min_z = float(np.float32(rng.uniform(-800.0, -10.0)))
        f.max_z = float(np.float32(rng.uniform(-9.0, 50.0)))
        f.world_units_per_texel = float(np.float32(f.extent / 2048.0))
        return f
    t = like.world_units_per_texel
    f.center_x = float(np.float32(like.center_x + rng.uniform(-2.5, 2.5) * t))
    f.center_y = float(np.float32(like.center_y + rng.uniform(-2.5, 2.5) * t))
    f.extent = float(np.float32(like.extent + rng.uniform(-4.0, 4.0) * t))
    f.min_z = float(np.float32(like.min_z + rng.uniform(-4.0, 4.0) * t))
    f.max_z = float(np.float32(like.max_z + rng.uniform(-4.0, 4.0) * t))
    f.world_units_per_texel = float(np.float32(f.extent / 2048.0))
    return f
