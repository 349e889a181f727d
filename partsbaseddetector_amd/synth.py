"""Portable synthetic inputs (frames and models) for tests and bench.

There is no model or image in the reference tree (``models/`` is an empty git submodule, SURVEY.md
section 0), so every input is generated here.  The generators use a counter-based splitmix64 stream
and integer / exactly-representable float64 arithmetic only (no libm calls), so the same seed gives
bit-identical arrays on every machine -- required because golden fixtures are produced in the build
container and re-checked on the GPU box.
"""
from __future__ import annotations

import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n 64-bit outputs of splitmix64 started at ``seed`` (counter based, vectorised)."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03 + 0x1234567) & _M64)
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = base + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_u32(seed: int, n: int, stream: int = 0) -> np.ndarray:
    return (splitmix64(seed, n, stream) >> np.uint64(32)).astype(np.int64)


def randint(seed: int, n: int, lo: int, hi: int, stream: int = 0) -> np.ndarray:
    """integers in [lo, hi] (inclusive); tiny modulo bias is irrelevant here."""
    return lo + (uniform_u32(seed, n, stream) % (hi - lo + 1))


def normalish(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """Approximately N(0,1) float64 values: Irwin-Hall sum of 12 uniform 16-bit integers.

    Exact integer sums, one exact scaling -> bit-identical everywhere."""
    raw = splitmix64(seed, 3 * n, stream).reshape(n, 3)
    acc = np.zeros(n, dtype=np.int64)
    for k in range(3):
        w = raw[:, k]
        for s in (0, 16, 32, 48):
            acc += ((w >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)
    # 12 uniforms on [0, 65535]: mean 12*32767.5, variance 12*(65536^2-1)/12
    return (acc.astype(np.float64) - 393210.0) / 65536.0


def synthetic_frame(seed: int, rows: int = 480, cols: int = 640, channels: int = 3, kind: str = "scene") -> np.ndarray:
    """uint8 frame (rows, cols, channels), BGR interleaved when channels == 3.

    kind = "scene": smooth background (integer bilinear up-sampling of a 1/16-resolution random
    grid) + 8 random filled rectangles + +-8 uniform noise;
    "noise": uniform noise; "constant": all 128."""
    if kind == "constant":
        return np.full((rows, cols, channels), 128, dtype=np.uint8)
    if kind == "noise":
        v = uniform_u32(seed, rows * cols * channels, stream=7) & 0xFF
        return v.reshape(rows, cols, channels).astype(np.uint8)
    gr, gc = rows // 16 + 2, cols // 16 + 2
    grid = (uniform_u32(seed, gr * gc * channels, stream=1) & 0xFF).reshape(gr, gc, channels)
    y = np.arange(rows, dtype=np.int64)
    x = np.arange(cols, dtype=np.int64)
    y0, fy = y // 16, (y % 16)[:, None, None]
    x0, fx = x // 16, (x % 16)[None, :, None]
    g00 = grid[y0][:, x0]
    g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]
    g11 = grid[y0 + 1][:, x0 + 1]
    img = ((16 - fy) * ((16 - fx) * g00 + fx * g01) + fy * ((16 - fx) * g10 + fx * g11) + 128) >> 8
    rect = uniform_u32(seed, 8 * (4 + channels), stream=2).reshape(8, 4 + channels)
    for r in rect:
        ry, rx = int(r[0] % rows), int(r[1] % cols)
        rh, rw = 8 + int(r[2] % max(rows // 3, 1)), 8 + int(r[3] % max(cols // 3, 1))
        img[ry:ry + rh, rx:rx + rw, :] = (r[4:4 + channels] & 0xFF)[None, None, :]
    noise = (uniform_u32(seed, rows * cols * channels, stream=3) % 17) - 8
    img = img + noise.reshape(rows, cols, channels)
    return np.clip(img, 0, 255).astype(np.uint8)


# 1-based parent table of the 26-part person tree (parent < child as asserted
# in matlab/detection/detect_fast.m:76)
PERSON_PA = [0, 1, 2, 3, 4, 5, 6, 3, 8, 9, 10, 11, 12, 13, 2, 15, 16, 17, 18, 15, 20, 21, 22, 23, 24, 25]
