"""The STC sub-matrix schedule (embed.h:340-393) as the GPU computes it.

The reference walks the message bits serially: bit i takes the longer sub-matrix while the columns used so far
stay <= (i + 1) * invalpha + 0.5.  k_embed_prepare gives every message bit to its own thread using the closed form
"columns before bit i = floor(i * invalpha + 0.5)"; this test checks, in the same IEEE double arithmetic, that the
closed form reproduces the serial walk (start column, width and sub-matrix choice of every bit)."""
import math
import random

import numpy as np


def serial(n, m):
    invalpha = n / m
    shorter, longer = math.floor(invalpha), math.ceil(invalpha)
    worm, out = 0, []
    for i in range(m):
        if worm + longer <= (i + 1) * invalpha + 0.5:
            out.append((worm, 1, longer)); worm += longer
        else:
            out.append((worm, 0, shorter)); worm += shorter
    return out, worm


def closed(n, m):
    invalpha = n / m
    shorter, longer = math.floor(invalpha), math.ceil(invalpha)
    i = np.arange(m + 1, dtype=np.float64)
    before = np.floor(i * invalpha + 0.5).astype(np.int64)
    before[0] = 0
    start = before[:-1]
    which = (start + longer).astype(np.float64) <= (i[1:] * invalpha + 0.5)
    width = np.where(which, longer, shorter)
    return start, which.astype(np.int64), width, int(before[m])


def check(n, m):
    ref, worm = serial(n, m)
    start, which, width, nproc = closed(n, m)
    assert nproc == worm, (n, m)
    r = np.array(ref, dtype=np.int64).reshape(-1, 3)
    assert np.array_equal(r[:, 0], start) and np.array_equal(r[:, 1], which) and np.array_equal(r[:, 2], width), (n, m)


def test_closed_form_small_exhaustive():
    for n in range(1, 80):
        for m in range(1, n + 1):
            check(n, m)


def test_closed_form_random_large():
    rng = random.Random(5)
    for _ in range(250):
        n = rng.randint(100, 140000)
        m = rng.choice([max(1, n // 2), max(1, int(0.5 * n)), rng.randint(1, n), max(1, int(n * rng.random() * 0.3)), n])
        check(n, m)
    for n in (8160, 8208, 130560, 99 * 16):
        for rate in (0.1, 0.25, 0.3, 0.5, 0.75, 1.0):
            check(n, max(1, int(np.float32(rate) * np.float32(n))))
