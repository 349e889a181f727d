"""The scheme of the wavefront-cooperative distance-transform pass (k_dt_coop, pbd_kernels_dp.hip) replayed on the CPU
against a literal transcription of computeRow (include/DistanceTransform.hpp:152-182, Quadratic :89-105): a row's envelope
kept as [top aligned block of W entries | everything written through to a flat array], the pop predicate of ALL window
entries evaluated for a new element and the new top taken as the highest entry whose predicate is false, a lower block
fetched back when the whole window pops, and the read-out as a binary search of z.  Same float / double mix as the kernels
(double-precision intersection rounded once to float).  The GPU kernel itself is checked against the oracle in
tests/test_gpu_parity.py; this test pins the SCHEME: it must make the sequential algorithm's decisions, not similar ones."""
import numpy as np
import pytest

f32 = np.float32


def _isect(a, b, x0, x1, y0, y1):
    with np.errstate(all="ignore"):
        return f32(((np.float64(y1) - np.float64(y0)) - b * (x1 - x0) + a * (x1 * x1 - x0 * x0)) / np.float64((2 * a) * (x1 - x0)))


def _val(a, b, x, y):
    return f32((a * x * x + b * x) + np.float64(y))


def _compute_row(src, a, b, os0):
    n = len(src)
    v, z = [0], [f32(-np.inf)]
    for q in range(1, n):
        k = len(v) - 1
        s = _isect(a, b, v[k], q, src[v[k]], src[q])
        while s <= z[k] and k > 0:
            k -= 1
            s = _isect(a, b, v[k], q, src[v[k]], src[q])
        v, z = v[:k + 1] + [q], z[:k + 1] + [s]
    z = z + [f32(np.inf)]
    out, ptr, k, os = [], [], 0, os0
    for q in range(n):
        while z[k + 1] < f32(os):
            k += 1
        out.append(_val(a, b, os - v[k], src[v[k]]))
        ptr.append(v[k])
        os += 1
    return np.array(out, f32), np.array(ptr)


def _cooperative(src, a, b, os0, W):
    n = len(src)
    zs, ys, vs = np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, int)        # the write-through copy ("LDS")
    ez, ey, ev = np.full(W, f32(-np.inf)), np.zeros(W, f32), np.zeros(W, int)  # the window ("lanes")
    k = wbase = 0
    ey[0] = src[0]
    zs[0], ys[0], vs[0] = -np.inf, src[0], 0
    reloads = 0
    for q in range(1, n):
        while True:
            m, s = 0, [f32(np.nan)] * W
            for sub in range(W):
                e = wbase + sub
                if q != ev[sub]:
                    s[sub] = _isect(a, b, int(ev[sub]), q, ey[sub], src[q])
                if e <= k and not ((s[sub] <= ez[sub]) and e > 0):
                    m |= 1 << sub
            if m:
                break
            wbase -= W
            k = wbase + W - 1
            reloads += 1
            for sub in range(W):
                ez[sub], ey[sub], ev[sub] = zs[wbase + sub], ys[wbase + sub], vs[wbase + sub]
        top = wbase + m.bit_length() - 1
        k = top + 1
        if k % W == 0:
            wbase = k
        sub = k % W
        ev[sub], ey[sub], ez[sub] = q, src[q], s[top % W]
        zs[k], ys[k], vs[k] = s[top % W], src[q], q
    out, ptr = np.zeros(n, f32), np.zeros(n, int)
    for pq in range(n):
        osf, lo, hi = f32(os0 + pq), 0, k
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if zs[mid] < osf:
                lo = mid
            else:
                hi = mid - 1
        out[pq], ptr[pq] = _val(a, b, os0 + pq - vs[lo], ys[lo]), vs[lo]
    return out, ptr, reloads


@pytest.mark.parametrize("W", [16, 8])
def test_cooperative_scheme_makes_the_sequential_decisions(W):
    rng = np.random.default_rng(W)
    reloads = 0
    for t in range(160):
        n = int(rng.integers(1, 180))
        kind = t % 4
        x = np.arange(n) - n / 2
        if kind == 0:
            src = rng.standard_normal(n)
        elif kind == 1:
            src = np.cumsum(rng.standard_normal(n)) * 0.1
        elif kind == 2:
            src = x * x * 0.02 * (1 if t % 8 < 4 else -1) + rng.standard_normal(n) * 0.01    # deep stacks / long pop runs
        else:
            src = np.where(rng.random(n) < 0.15, 3.0, 0.0) + rng.standard_normal(n) * 1e-3   # plateaus with spikes
        src = src.astype(f32)
        a = -float(rng.choice([0.01, 0.05, 0.002]))
        b = -float(rng.choice([0.0, 0.0, 0.03, -0.02]))
        os0 = int(rng.integers(-6, 7))
        want_v, want_p = _compute_row(src, a, b, os0)
        got_v, got_p, r = _cooperative(src, a, b, os0, W)
        reloads += r
        assert np.array_equal(want_v.view(np.uint32), got_v.view(np.uint32)) and np.array_equal(want_p, got_p), (t, n, kind)
    assert reloads > 0          # the lower-block path was exercised
