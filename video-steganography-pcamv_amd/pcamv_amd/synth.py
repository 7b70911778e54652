"""Deterministic synthetic I420 clips (SURVEY.md section 8(d) generator).

No sample video ships with the reference, so every test and the bench use this generator:
a blurred-noise texture panned 3 px/frame in x and 2 px/frame in y, per-pixel noise in
[-6, 6], and one 64x64 patch moving (+5, +3) px/frame.  Chroma is a second, smoother texture
panned at half the luma rate (the survey's constant-128 chroma would leave the chroma-ME
paths untested).
"""
import numpy as np

SEEDS = {"qcif": 5, "cif": 7, "720p": 11, "1080p": 13, "2160p": 17}


def _blur5(a, passes):
    a = a.astype(np.float32)
    for _ in range(passes):
        p = np.pad(a, 1, mode="edge")
        a = (p[1:-1, 1:-1] + p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:]) / 5.0
    return a


def make_clip(width, height, frames, seed=7, noise=6, static_cols=0):
    """Return a list of (Y, U, V) uint8 arrays of shape (h, w), (h/2, w/2), (h/2, w/2).

    static_cols > 0 keeps that many left-hand columns (a multiple of 16) motionless and
    noise-free, which is what produces P_SKIP macroblocks."""
    rng = np.random.default_rng(seed)
    F = frames
    th, tw = height + 4 * F + 64, width + 4 * F + 64
    tex = _blur5(rng.integers(0, 256, (th, tw)).astype(np.float32), 3)
    tex = (tex - tex.mean()) * 2.2 + 128.0
    ctex_u = _blur5(rng.integers(0, 256, (th // 2 + 2, tw // 2 + 2)).astype(np.float32), 5)
    ctex_v = _blur5(rng.integers(0, 256, (th // 2 + 2, tw // 2 + 2)).astype(np.float32), 5)
    ctex_u = (ctex_u - ctex_u.mean()) * 3.0 + 128.0
    ctex_v = (ctex_v - ctex_v.mean()) * 3.0 + 128.0
    patch = tex[7:7 + 64, 11:11 + 64].copy()
    out = []
    for t in range(F):
        y0, x0 = 2 * t, 3 * t
        Y = tex[y0:y0 + height, x0:x0 + width].copy()
        Y += rng.integers(-noise, noise + 1, Y.shape)
        py, px = (20 + 3 * t) % max(1, height - 64), (30 + 5 * t) % max(1, width - 64)
        if height >= 64 and width >= 64:
            Y[py:py + 64, px:px + 64] = patch
        U = ctex_u[y0 // 2:y0 // 2 + height // 2, x0 // 2:x0 // 2 + width // 2].copy()
        V = ctex_v[y0 // 2:y0 // 2 + height // 2, x0 // 2:x0 // 2 + width // 2].copy()
        if static_cols:
            sc = static_cols
            Y[:, :sc] = tex[:height, :sc]
            U[:, :sc // 2] = ctex_u[:height // 2, :sc // 2]
            V[:, :sc // 2] = ctex_v[:height // 2, :sc // 2]
        out.append((np.clip(Y + 0.5, 0, 255).astype(np.uint8),
                    np.clip(U + 0.5, 0, 255).astype(np.uint8),
                    np.clip(V + 0.5, 0, 255).astype(np.uint8)))
    return out
