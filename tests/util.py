"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
import os

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_image(seed, h, w, opaque=True):
    a = np.random.default_rng(seed).integers(0, 256, (h, w, 4), dtype=np.uint8)
    if opaque:
        a[..., 3] = 255
    return a


def smooth_image(seed, h, w, opaque=True):
    """Low-frequency content (gradients + a few blobs): closer to photos than noise, exercises the lerp weights."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w, 4), np.float64)
    for c in range(4):
        fx, fy, ph = rng.uniform(0.5, 3.0), rng.uniform(0.5, 3.0), rng.uniform(0, 6.28)
        out[..., c] = 127.5 + 127.5 * np.sin(fx * xx / max(w, 1) * 6.28 + fy * yy / max(h, 1) * 6.28 + ph)
    out = np.clip(np.round(out), 0, 255).astype(np.uint8)
    if opaque:
        out[..., 3] = 255
    return out


def oracle_limits(opts):
    """Same meaning as imagestitching_amd.stitch._limits, built for the oracle."""
    opts = opts or {}
    plat = opts.get("platform")
    if plat is None:
        lim = O.lifted_limits(1.0)
    else:
        lim = O.default_limits(plat)
    if opts.get("maxSide") is not None:
        lim.max_side = float(opts["maxSide"])
    if opts.get("maxPixels") is not None:
        lim.max_pixels = float(opts["maxPixels"])
    if opts.get("superSample") is not None:
        lim.max_super_sample = float(opts["superSample"])
    return lim


def edge_aa_of(opts):
    """the hosts' default (imagestitching_amd.stitch.edge_aa_of, node/index.js edgeAA): on iff a reference platform's plan
    is requested, unless the caller says otherwise"""
    v = (opts or {}).get("edgeAA")
    return ((opts or {}).get("platform") is not None) if v is None else bool(v)


def oracle_stitch(pixels, direction, opts=None, orientations=None, threads=4):
    opts = opts or {}
    descs = [{"width": a.shape[1], "height": a.shape[0], "orientation": (orientations[i] if orientations else 1)}
             for i, a in enumerate(pixels)]
    rc, pd, rl = O.plan(descs, direction, opts.get("mode", "min"), opts.get("gap", 0), oracle_limits(opts))
    assert rc == 0, rc
    img = O.render(pd, rl, descs, pixels, opts.get("filter", "bilinear"), threads, edge_aa=edge_aa_of(opts))
    return img, pd, rl


def hip_images(pixels, orientations=None):
    return [{"width": a.shape[1], "height": a.shape[0], "data": a, "orientation": (orientations[i] if orientations else 1)}
            for i, a in enumerate(pixels)]


def max_abs_diff(a, b):
    return int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max()) if a.size else 0


def mismatch_fraction(a, b):
    return float((a != b).any(axis=-1).mean()) if a.size else 0.0
