#!/usr/bin/env python3
"""Writes tests/golden/shadow_fit.npz: inputs and outputs of the REFERENCE's own
cascade fit (oracle/_ref/libvkr_ref_shadow.so, i.e. this needs /root/reference)
for the steps that are exactly reproducible anywhere - one cascade from a light
view and slice corners (vkr_shadow_compute_cascade_matrix), the caster depth
fit, extent quantisation, fit hysteresis. Run from the repository root:

    python tests/golden/make_shadow_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as P  # noqa: E402
import shadow_cases as cases  # noqa: E402

CONFIG_FIELDS = [n for n, _ in P.ShadowConfig._fields_]


def config_row(cfg):
    return np.frombuffer(bytes(cfg), np.uint8).copy()


def fit_row(f):
    return np.array([getattr(f, n) for n, _ in P.ShadowFit._fields_], np.float32)


def main():
    P.build()
    ref = P.ReferenceShadow()
    rng = np.random.default_rng(20261111)
    out = {}
    # one cascade from (light view, corners)
    n = 96
    keys = ("light_view", "corners", "params", "config", "previous", "has_previous", "vp", "fit", "origin")
    acc = {k: [] for k in keys}
    for _ in range(n):
        cfg = cases.random_config(rng, ref.default_config())
        cam = cases.random_camera(rng, P.shadow_camera)
        light = cases.random_light(rng)
        lv = ref.light_view(cam, cfg, light, cfg.shadow_map_size)
        splits = ref.cascade_splits(cfg, cam.near_clip, cam.far_clip, cfg.cascade_split_lambda)
        i = int(rng.integers(0, cfg.cascade_count))
        corners = ref.frustum_corners(cam, float(splits[i]), float(splits[i + 1]))
        guard = float(np.float32(rng.choice([0.0, 16.0, 40.0])))
        zext = float(np.float32(rng.choice([0.0, 2.0, 5.0])))
        vp0, fit0, _ = ref.cascade_matrix(lv, corners, cfg.shadow_map_size, cfg.stabilize_cascades, guard,
                                          cfg.use_constant_cascade_size, cfg, zext)
        has_prev = bool(rng.random() < 0.5)
        prev = cases.random_fit(rng, P.ShadowFit, like=fit0) if has_prev else P.ShadowFit()
        vp, fit, origin = ref.cascade_matrix(lv, corners, cfg.shadow_map_size, cfg.stabilize_cascades, guard,
                                             cfg.use_constant_cascade_size, cfg, zext,
                                             prev if has_prev else None)
        acc["light_view"].append(lv)
        acc["corners"].append(corners.reshape(24))
        acc["params"].append(np.array([guard, zext], np.float32))
        acc["config"].append(config_row(cfg))
        acc["previous"].append(fit_row(prev))
        acc["has_previous"].append(has_prev)
        acc["vp"].append(vp)
        acc["fit"].append(fit_row(fit))
        acc["origin"].append(origin)
    for k in keys:
        out["cascade_" + k] = np.array(acc[k])
    # caster depth fit
    m = 128
    lvs, boxes, rects, res = [], [], [], []
    for _ in range(m):
        cfg = cases.random_config(rng, ref.default_config())
        cam = cases.random_camera(rng, P.shadow_camera)
        lv = ref.light_view(cam, cfg, cases.random_light(rng), cfg.shadow_map_size)
        mn = np.array(cfg.scene_min[:], np.float32)
        mx = np.array(cfg.scene_max[:], np.float32)
        c = rng.uniform(-300, 300, 2)
        h = rng.uniform(1.0, 500.0)
        rect = np.array([c[0] - h, c[0] + h, c[1] - h, c[1] + h], np.float32)
        ok, lo, hi = ref.fit_relevant_caster_z(lv, mn, mx, *[float(x) for x in rect])
        lvs.append(lv)
        boxes.append(np.concatenate([mn, mx]))
        rects.append(rect)
        res.append(np.array([1.0 if ok else 0.0, lo if ok else 0.0, hi if ok else 0.0], np.float32))
    out["caster_light_view"], out["caster_box"] = np.array(lvs), np.array(boxes)
    out["caster_rect"], out["caster_result"] = np.array(rects), np.array(res)
    # quantisation
    ext = np.concatenate([rng.uniform(0.01, 2000.0, 200), [137.31, 137.3101, 1.0, 256.0, 255.99]]).astype(np.float32)
    sizes = rng.choice([512, 1024, 2048, 4096], ext.shape[0]).astype(np.uint32)
    out["quantize_extent"], out["quantize_size"] = ext, sizes
    out["quantize_result"] = np.array([ref.quantize_extent_up(float(e), int(s)) for e, s in zip(ext, sizes)],
                                      np.float32)
    # hysteresis
    prevs, raws, fits = [], [], []
    for _ in range(200):
        p = cases.random_fit(rng, P.ShadowFit)
        r = cases.random_fit(rng, P.ShadowFit, like=p)
        prevs.append(fit_row(p))
        raws.append(fit_row(r))
        fits.append(fit_row(ref.apply_fit_hysteresis(p, r)))
    out["hysteresis_previous"], out["hysteresis_raw"], out["hysteresis_result"] = (
        np.array(prevs), np.array(raws), np.array(fits))
    path = os.path.join(ROOT, "tests", "golden", "shadow_fit.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
