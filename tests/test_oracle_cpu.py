"""CPU tests of the oracle (the checker) against INDEPENDENT formulations of each stage.

The reference ships no golden vectors for this path and cannot be built here (SURVEY.md section 4 /
8c: "parity unpinned"), so the restatement is cross-checked by code with a different structure:
brute-force max-plus distance transform, scipy correlation, a scatter-form float64 HOG, an integer
numpy pyrDown, and a brute-force tree max-sum for the dynamic program."""
import numpy as np
import pytest
from scipy import signal

from partsbaseddetector_amd import model as M
from partsbaseddetector_amd import synth


# ---------------------------------------------------------------------------------- geometry
def test_pyramid_plan_matches_survey_appendix_b(oracle):
    # SURVEY.md Appendix B (computed from src/HOGFeatures.cpp:99,116-124,174-175)
    for (rows, cols, interval), (levels, pixels, cells) in {
        (480, 640, 10): (46, 2371512, 140725),
        (1080, 1920, 10): (58, 16019919, 980592),
        (240, 320, 5): (18, 315695, 17945),
        (240, 320, 10): (36, 590699, 33459),
    }.items():
        lr, lc, sc = oracle.pyramid_plan(rows, cols, 4, interval)
        assert len(lr) == levels
        assert int(np.sum(lr.astype(np.int64) * lc)) == pixels
        assert sum(int(np.prod(oracle.hog_dims(int(a), int(b), 4))) for a, b in zip(lr, lc)) == cells
    lr, lc, sc = oracle.pyramid_plan(480, 640, 4, 10)
    assert (lr[0], lc[0], lr[1], lc[1], lr[2], lc[2]) == (480, 640, 448, 597, 418, 557)
    assert (lr[-1], lc[-1]) == (22, 29) and oracle.hog_dims(22, 29, 4) == (4, 5)
    assert sc[0] == 4.0 and sc[10] == 8.0 and np.all(sc[10:20] == 2 * sc[:10])


def test_plan_float_vs_double_sfactor_agree():
    # include/HOGFeatures.hpp:78: pow(2.0f, 1.0f/interval) -- float or double overload gives the same float
    for interval in (1, 2, 3, 5, 8, 10):
        a = np.float32(2.0) ** np.float32(np.float32(1.0) / np.float32(interval))
        b = np.float32(2.0 ** float(np.float32(1.0) / np.float32(interval)))
        assert np.float32(a) == b


# ---------------------------------------------------------------------------------- resampling
def test_resize_identity_and_bilinear(oracle):
    im = synth.synthetic_frame(4, 61, 83, 3)
    assert np.array_equal(oracle.resize_linear_u8(im, 61, 83), im)
    out = oracle.resize_linear_u8(im, 44, 59)
    # float64 bilinear with the same half-pixel mapping; the fixed-point path is within 1 grey level
    ys = (np.arange(44) + 0.5) * (61 / 44) - 0.5
    xs = (np.arange(59) + 0.5) * (83 / 59) - 0.5
    y0 = np.floor(ys).astype(int); fy = ys - y0
    x0 = np.floor(xs).astype(int); fx = xs - x0
    y0c, y1c = np.clip(y0, 0, 60), np.clip(y0 + 1, 0, 60)
    x0c, x1c = np.clip(x0, 0, 82), np.clip(x0 + 1, 0, 82)
    f = im.astype(np.float64)
    ref = ((1 - fy)[:, None, None] * ((1 - fx)[None, :, None] * f[y0c][:, x0c] + fx[None, :, None] * f[y0c][:, x1c])
           + fy[:, None, None] * ((1 - fx)[None, :, None] * f[y1c][:, x0c] + fx[None, :, None] * f[y1c][:, x1c]))
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.01


def _resize_u8_numpy(im, dh, dw):
    """SURVEY Appendix E's 8-bit INTER_LINEAR algorithm written again, vectorised and from the text alone (fixed point, 11
    coefficient bits, horizontal pass in int, vertical pass ((b * (r >> 4)) >> 16 twice) + 2 >> 2) -- an independent statement
    of the same published algorithm, not a second opinion on OpenCV itself (which nothing here can provide)."""
    sh, sw = im.shape[:2]

    def coef(dn, sn):
        scale = 1.0 / (float(dn) / float(sn))
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        lo, hi = s0 < 0, s0 >= sn - 1
        f = np.where(lo | hi, np.float32(0), f)
        s0 = np.where(lo, 0, np.where(hi, sn - 1, s0))
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)      # cvRound: half to even, as np.rint
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s0, np.minimum(s0 + 1, sn - 1), a0, a1

    if (dh, dw) == (sh, sw):
        return im.copy()
    sx0, sx1, a0, a1 = coef(dw, sw)
    sy0, sy1, b0, b1 = coef(dh, sh)
    S = im.astype(np.int64)
    R = S[:, sx0] * a0[None, :, None] + S[:, sx1] * a1[None, :, None]              # every source row, horizontally
    r0, r1 = R[sy0], R[sy1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("shape,dst", [((61, 83), (44, 59)), ((61, 83), (58, 80)), ((97, 131), (49, 66)), ((240, 320), (224, 299)),
                                       ((50, 61), (37, 45))])
@pytest.mark.parametrize("cn", [3, 1])
def test_resize_u8_against_an_independent_integer_formulation(oracle, shape, dst, cn):
    im = synth.synthetic_frame(9, shape[0], shape[1], cn)
    if im.ndim == 2:
        im = im[:, :, None]
    got = oracle.resize_linear_u8(im if cn == 3 else im[:, :, 0], dst[0], dst[1])
    want = _resize_u8_numpy(im, dst[0], dst[1])
    assert np.array_equal(np.asarray(got).reshape(want.shape), want)


def test_pyrdown_integer_formulation(oracle):
    for shape, cn in [((37, 52), 3), ((40, 41), 1), ((5, 4), 3)]:
        im = synth.synthetic_frame(9, shape[0], shape[1], cn, kind="noise")
        got = oracle.pyrdown_u8(im)
        pad = np.pad(im.astype(np.int64), ((2, 3), (2, 3), (0, 0)), mode="reflect")   # numpy reflect == REFLECT_101
        k = np.array([1, 4, 6, 4, 1], np.int64)
        dr, dc = (shape[0] + 1) // 2, (shape[1] + 1) // 2
        acc = np.zeros((dr, dc, cn), np.int64)
        for i in range(5):
            for j in range(5):
                acc += k[i] * k[j] * pad[i:i + 2 * dr:2, j:j + 2 * dc:2]
        assert np.array_equal(got, ((acc + 128) >> 8).astype(np.uint8))


# ---------------------------------------------------------------------------------- HOG
def _hog_scatter_f64(im, sbin=4):
    """Scatter-form restatement in float64 following the published algorithm (matlab/mex/features.cc
    as adapted by src/HOGFeatures.cpp); independent code, used only for a tolerance comparison."""
    rows, cols, cn = im.shape
    bh, bw = int(np.floor(rows / sbin + 0.5)), int(np.floor(cols / sbin + 0.5))
    hist = np.zeros((bh, bw, 18))
    uu = np.array([1.000, 0.9397, 0.7660, 0.5000, 0.1736, -0.1736, -0.5000, -0.7660, -0.9397])
    vv = np.array([0.000, 0.3420, 0.6428, 0.8660, 0.9848, 0.9848, 0.8660, 0.6428, 0.3420])
    f = im.astype(np.float64)
    for y in range(1, bh * sbin - 1):
        ys = min(y, rows - 2)
        for x in range(1, bw * sbin - 1):
            xs = min(x, cols - 2)
            dy = f[ys + 1, xs] - f[ys - 1, xs]
            dx = f[ys, xs + 1] - f[ys, xs - 1]
            v = dx * dx + dy * dy
            if cn == 3:
                c = 2                     # start from channel 2, prefer 1, then 0 on strictly larger magnitude
                if v[1] > v[c]: c = 1
                if v[0] > v[c]: c = 0
            else:
                c = 0
            dxc, dyc, vc = dx[c], dy[c], v[c]
            dots = uu * dxc + vv * dyc
            best, bo = 0.0, 0
            for o in range(9):
                if dots[o] > best: best, bo = dots[o], o
                elif -dots[o] > best: best, bo = -dots[o], o + 9
            yp, xp = (y + 0.5) / sbin - 0.5, (x + 0.5) / sbin - 0.5
            iy, ix = int(np.floor(yp)), int(np.floor(xp))
            vy0, vx0 = yp - iy, xp - ix
            mag = np.sqrt(vc)
            for (yy, wy) in ((iy, 1 - vy0), (iy + 1, vy0)):
                for (xx, wx) in ((ix, 1 - vx0), (ix + 1, vx0)):
                    if 0 <= yy < bh and 0 <= xx < bw:
                        hist[yy, xx, bo] += wy * wx * mag
    norm = ((hist[:, :, :9] + hist[:, :, 9:]) ** 2).sum(axis=2)
    oh, ow = max(bh - 2, 0), max(bw - 2, 0)
    feat = np.zeros((oh, ow, 32))
    for y in range(oh):
        for x in range(ow):
            ns = []
            for (yy, xx) in ((y + 1, x + 1), (y, x + 1), (y + 1, x), (y, x)):
                ns.append(1.0 / np.sqrt(norm[yy, xx] + norm[yy, xx + 1] + norm[yy + 1, xx] + norm[yy + 1, xx + 1] + 1e-4))
            h = hist[y + 1, x + 1]
            hs = np.stack([np.minimum(h * n, 0.2) for n in ns])
            feat[y, x, :18] = 0.5 * hs.sum(axis=0)
            s = h[:9] + h[9:]
            feat[y, x, 18:27] = 0.5 * np.stack([np.minimum(s * n, 0.2) for n in ns]).sum(axis=0)
            feat[y, x, 27:31] = 0.2357 * hs.sum(axis=1)
    return feat.reshape(oh, ow * 32)


@pytest.mark.parametrize("shape,cn", [((50, 61), 3), ((47, 38), 1)])
def test_hog_against_scatter_form(oracle, shape, cn):
    im = synth.synthetic_frame(2, shape[0], shape[1], cn)
    got = oracle.hog_features(im, 4)
    ref = _hog_scatter_f64(im, 4)
    assert got.shape == ref.shape
    assert np.abs(got.astype(np.float64) - ref).max() < 2e-5
    assert not got.reshape(got.shape[0], -1, 32)[:, :, 31].any()        # truncation channel is 0 (:338)
    got64 = oracle.hog_features(im, 4, dtype=np.float64)
    assert np.abs(got64 - ref).max() < 1e-9


def test_hog_constant_image_is_zero(oracle):
    im = synth.synthetic_frame(0, 40, 44, 3, kind="constant")
    assert not oracle.hog_features(im, 4).any()


# ---------------------------------------------------------------------------------- convolution
def test_conv_against_scipy(oracle):
    rng = np.random.default_rng(1)
    H, W, k = 13, 17, 5
    feat = rng.random((H, W, 32)).astype(np.float32)
    feat[:, :, 31] = 0
    filt = (rng.standard_normal((k, k, 32)) * 0.1).astype(np.float32)
    filt[0, 0, 3] = 0.0   # a skipped tap
    got = oracle.conv(feat.reshape(H, W * 32), filt.reshape(k, k * 32))
    ref = np.zeros((H, W))
    for c in range(32):
        border = 1.0 if c == 31 else 0.0
        plane = np.pad(feat[:, :, c].astype(np.float64), 2, constant_values=border)
        ref += signal.correlate2d(plane, filt[:, :, c].astype(np.float64), mode="valid")
    assert np.abs(got - ref).max() < 1e-5
    # the border-of-ones: with zero features only channel 31's out-of-image taps contribute
    z = np.zeros((H, W * 32), np.float32)
    got0 = oracle.conv(z, filt.reshape(k, k * 32))
    ref0 = signal.correlate2d(np.pad(np.zeros((H, W)), 2, constant_values=1.0), filt[:, :, 31].astype(np.float64), mode="valid")
    assert np.abs(got0 - ref0).max() < 1e-6 and got0[H // 2, W // 2] == 0.0


def test_conv_even_kernel_anchor(oracle):
    rng = np.random.default_rng(2)
    feat = rng.random((6, 7, 32)).astype(np.float32)
    filt = rng.standard_normal((4, 4, 32)).astype(np.float32)
    got = oracle.conv(feat.reshape(6, 7 * 32), filt.reshape(4, 4 * 32))
    ref = np.zeros((6, 7))
    for c in range(32):
        plane = np.pad(feat[:, :, c].astype(np.float64), ((2, 1), (2, 1)), constant_values=1.0 if c == 31 else 0.0)   # anchor = k/2 = 2
        ref += signal.correlate2d(plane, filt[:, :, c].astype(np.float64), mode="valid")
    assert np.abs(got - ref).max() < 1e-4


# ---------------------------------------------------------------------------------- distance transform
def _dt_brute(score, ax, bx, ay, by, osx, osy):
    M_, N_ = score.shape
    s = score.astype(np.float64)
    n = np.arange(N_)
    dx = osx + n[:, None] - n[None, :]                    # [out n, src n']
    rows = (ax * dx * dx + bx * dx)[None, :, :] + s[:, None, :]   # [m, n, n']
    tmp = rows.max(axis=2)
    ixr = rows.argmax(axis=2)
    tmp32 = tmp.astype(np.float32).astype(np.float64)   # the reference stores the row pass in T
    m = np.arange(M_)
    dy = osy + m[:, None] - m[None, :]
    cols = (ay * dy * dy + by * dy)[:, :, None] + tmp32[None, :, :]   # [m, m', n]
    out = cols.max(axis=1)
    iyr = cols.argmax(axis=1)
    return out, ixr, iyr


@pytest.mark.parametrize("seed,shape,w,os", [(0, (9, 11), (0.01, 0.0, 0.01, 0.0), (0, 0)),
                                              (1, (23, 31), (0.01, 0.0, 0.01, 0.0), (3, -2)),
                                              (2, (17, 40), (0.012, 0.004, 0.02, -0.007), (-4, 4)),
                                              (3, (1, 25), (0.05, 0.0, 0.05, 0.0), (2, 0)),
                                              (4, (30, 1), (0.05, 0.01, 0.03, 0.0), (0, -3))])
def test_dt_against_brute_force(oracle, seed, shape, w, os):
    rng = np.random.default_rng(seed)
    score = rng.standard_normal(shape).astype(np.float32)
    ax, bx, ay, by = (-np.float32(w[0]), -np.float32(w[1]), -np.float32(w[2]), -np.float32(w[3]))
    out, Ix, Iy = oracle.dt(score, float(ax), float(bx), float(ay), float(by), os[0], os[1])
    ref, ixr, iyr = _dt_brute(score, float(ax), float(bx), float(ay), float(by), os[0], os[1])
    # scores: exact max (0 mismatches vs O(N^2) brute force)
    assert np.abs(out.astype(np.float64) - ref).max() < 2e-6
    assert np.mean(out == ref.astype(np.float32)) > 0.995
    # pointers follow the reference's composition Iy[m][n] = IyRaw[m][Ix[m][n]] (DistanceTransform.hpp:233-244)
    agree_x = np.mean(Ix == ixr)
    assert agree_x > 0.98          # envelope vs double brute force may differ on near ties
    quirk = np.take_along_axis(iyr, Ix, axis=1)
    assert np.mean(Iy == quirk) > 0.98
    assert Ix.min() >= 0 and Ix.max() < shape[1] and Iy.min() >= 0 and Iy.max() < shape[0]


def test_dt_quirk_differs_from_true_argmax(oracle):
    rng = np.random.default_rng(7)
    score = rng.standard_normal((23, 31)).astype(np.float32)
    out, Ix, Iy = oracle.dt(score, -0.01, 0.0, -0.01, 0.0, 1, 1)
    ref, ixr, iyr = _dt_brute(score, -0.01, 0.0, -0.01, 0.0, 1, 1)
    true_y = iyr
    assert np.mean(Iy != true_y) > 0.1    # the composition is NOT the true arg-max (Appendix A.3); reproduced on purpose


# ---------------------------------------------------------------------------------- dynamic program
def test_dp_min_against_brute_force_tree(oracle):
    model = M.synthetic_model(seed=5, pa=[0, 1, 1, 2], nmix=2, linear_def=True, anchor_range=2)
    flat = model.flatten()
    rng = np.random.default_rng(3)
    H, W = 7, 9
    resp = rng.standard_normal((flat.nfilters, H, W)).astype(np.float32)
    Ix, Iy, Ik, rootv, rooti = oracle.dp_min(flat, 0, resp)

    def fid(p, m): return model.filterid[0][p][m]
    children = {p: [c for c in range(model.nparts(0)) if model.parentid[0][c] == p] for p in range(model.nparts(0))}

    def score(p, m):   # full (H, W) accumulated score map of part p, mixture m in float64
        s = resp[fid(p, m)].astype(np.float64).copy()
        for c in children[p]:
            s += message(c, m)
        return s

    def message(c, pm):
        best = np.full((H, W), -np.inf)
        for mm in range(len(model.filterid[0][c])):
            d = model.defid[0][c][mm]
            w = np.float32(model.defw[d]).astype(np.float64)
            ax_, ay_ = model.anchors[d]
            sc = score(c, mm)
            out = np.full((H, W), -np.inf)
            for y in range(H):
                for x in range(W):
                    dx = ax_ + x - np.arange(W)[None, :]
                    dy = ay_ + y - np.arange(H)[:, None]
                    out[y, x] = np.max(sc - w[0] * dx * dx - w[1] * dx - w[2] * dy * dy - w[3] * dy)
            b = np.float32(model.biasw[model.biasid[0][c][mm] + pm])
            best = np.maximum(best, out + float(b))
        return best

    rb = float(np.float32(model.biasw[model.biasid[0][0][0]]))
    ref = np.max(np.stack([score(0, m) + rb for m in range(2)]), axis=0)
    assert np.abs(rootv.astype(np.float64) - ref).max() < 1e-4
    assert np.mean(rooti == np.argmax(np.stack([score(0, m) for m in range(2)]), axis=0)) > 0.98


def test_argmin_boxes_and_order(oracle):
    model = M.synthetic_tiny_model(thresh=0.7)
    flat = model.flatten()
    im = synth.synthetic_frame(5, 96, 128)
    cands = oracle.detect(flat, im)
    assert len(cands) > 0
    keys = [(c["level"], c["component"], c["root_y"], c["root_x"]) for c in cands]
    assert keys == sorted(keys)
    _, scales = oracle.features_pyramid(flat, im)
    for c in cands[:50]:
        assert c["score"] > flat.thresh
        s = np.float32(scales[c["level"]])
        x1 = int(np.rint(np.float32(c["root_x"] - 1) * s)); y1 = int(np.rint(np.float32(c["root_y"] - 1) * s))
        w = int(np.rint(np.float32(5) * s)) - 1
        assert tuple(c["parts"][0]) == (min(x1, x1 + w), min(y1, y1 + w), abs(w), abs(w))   # src/DynamicProgram.cpp:238-244


def test_f64_path_close_to_f32(oracle):
    model = M.synthetic_tiny_model(thresh=0.7)
    flat = model.flatten()
    im = synth.synthetic_frame(5, 96, 128)
    a = oracle.detect(flat, im, dtype=np.float32)
    b = oracle.detect(flat, im, dtype=np.float64)
    ka = {(c["level"], c["root_y"], c["root_x"]): c["score"] for c in a}
    kb = {(c["level"], c["root_y"], c["root_x"]): c["score"] for c in b}
    common = set(ka) & set(kb)
    assert len(common) > 0.95 * max(len(ka), len(kb))
    assert max(abs(ka[k] - kb[k]) for k in common) < 1e-4


# ---------------------------------------------------------------------------------------------------
# image depths other than 8-bit (src/HOGFeatures.cpp:136-146: features<uint16_t|float|double>)
# ---------------------------------------------------------------------------------------------------
def test_hog_of_wider_depths_equals_8bit_on_the_same_values(oracle):
    """The gradient is `*(s+a) - *(s-b)` in the pixel type: for 8-bit-valued pixels stored as uint16 / float / double
    the differences are the same numbers, so features<IT> must give the features<uint8_t> result bit for bit."""
    im8 = synth.synthetic_frame(5, 70, 93, 3)
    for T in (np.float32, np.float64):
        want = oracle.hog_features(im8, 4, 18, 32, T)
        for IT in (np.uint16, np.float32, np.float64):
            got = oracle.hog_features(im8.astype(IT), 4, 18, 32, T)
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (T, IT)
    # and a genuinely 16-bit image: scaling every pixel by 256 scales the unnormalised gradients by 256, which the
    # four block normalisations remove up to the 1e-4 epsilon -> same features within 1e-4
    a = oracle.hog_features(im8, 4, 18, 32, np.float64)
    b = oracle.hog_features(im8.astype(np.uint16) * 256, 4, 18, 32, np.float64)
    assert np.abs(a - b).max() < 2e-4


def _np_pyrdown(src, wt, finish):
    r, c, cn = src.shape
    dr, dc = (r + 1) // 2, (c + 1) // 2
    def refl(p, n):
        if n == 1:
            return 0
        while p < 0 or p >= n:
            p = -p if p < 0 else 2 * n - 2 - p
        return p
    s = src.astype(wt)
    rows = np.zeros((r, dc, cn), wt)
    for x in range(dc):
        x0, x1, x2, x3, x4 = refl(2 * x - 2, c), refl(2 * x - 1, c), 2 * x, refl(2 * x + 1, c), refl(2 * x + 2, c)
        rows[:, x] = ((s[:, x2] * wt(6) + (s[:, x1] + s[:, x3]) * wt(4)) + s[:, x0]) + s[:, x4]
    out = np.zeros((dr, dc, cn), wt)
    for y in range(dr):
        y0, y1, y2, y3, y4 = refl(2 * y - 2, r), refl(2 * y - 1, r), refl(2 * y, r), refl(2 * y + 1, r), refl(2 * y + 2, r)
        out[y] = ((rows[y2] * wt(6) + (rows[y1] + rows[y3]) * wt(4)) + rows[y0]) + rows[y4]
    return finish(out)


@pytest.mark.parametrize("IT", [np.uint16, np.float32, np.float64])
def test_pyramid_of_wider_depths_against_numpy(oracle, IT):
    """pyrDown (levels >= interval) against an independent numpy statement of the same taps and operation order;
    level 0 of the pyramid (resize to the same size) is the image itself; 16U-valued-as-8U equals the 8-bit pyramid."""
    rng = np.random.default_rng(3)
    if IT == np.uint16:
        im = rng.integers(0, 65536, (61, 77, 3)).astype(np.uint16)
    else:
        im = (rng.random((61, 77, 3)) * 255).astype(IT)
    imgs, scales = oracle.pyramid_images(im, 4, 3)
    assert imgs[0].dtype == IT and np.array_equal(imgs[0], im)
    for l in range(3, len(imgs)):
        if IT == np.uint16:
            want = _np_pyrdown(imgs[l - 3], np.int64, lambda v: ((v + 128) >> 8).astype(np.uint16))
        elif IT == np.float32:
            want = _np_pyrdown(imgs[l - 3], np.float32, lambda v: v * np.float32(1.0 / 256))
        else:
            want = _np_pyrdown(imgs[l - 3], np.float64, lambda v: v * (1.0 / 256))
        assert imgs[l].shape == want.shape and np.array_equal(imgs[l], want), l
    im8 = synth.synthetic_frame(9, 61, 77, 3)
    a, _ = oracle.pyramid_images(im8, 4, 3)
    b, _ = oracle.pyramid_images(im8.astype(np.uint16), 4, 3)
    for l in range(3, len(a)):              # the integer pyrDown is the same formula; (resized levels use float for 16U)
        if l % 3 == 0:
            assert np.array_equal(a[l], b[l].astype(np.uint8)) and b[l].max() < 256


def test_float_resize_against_numpy(oracle):
    """INTER_LINEAR on float pixels: dst = (S00*a0 + S01*a1)*b0 + (S10*a0 + S11*a1)*b1 with float coefficients
    (1-fx, fx), (1-fy, fy) of the same coordinate mapping as the 8-bit path."""
    rng = np.random.default_rng(4)
    src = (rng.random((37, 53, 1)) * 255).astype(np.float32)
    imgs, _ = oracle.pyramid_images(src, 4, 2)
    dst = imgs[1]
    dr, dc = dst.shape[:2]
    sr, sc = 37, 53
    want = np.zeros((dr, dc), np.float32)
    for dy in range(dr):
        fy = np.float32((dy + 0.5) * (1.0 / (dr / sr)) - 0.5)
        sy = int(np.floor(fy)); fy = np.float32(fy - np.float32(sy))
        y0, y1 = min(max(sy, 0), sr - 1), min(max(sy + 1, 0), sr - 1)
        b0, b1 = np.float32(1) - fy, fy
        for dx in range(dc):
            fx = np.float32((dx + 0.5) * (1.0 / (dc / sc)) - 0.5)
            sx = int(np.floor(fx)); fx = np.float32(fx - np.float32(sx))
            if sx < 0:
                fx, sx = np.float32(0), 0
            if sx >= sc - 1:
                r0, r1 = src[y0, sc - 1, 0], src[y1, sc - 1, 0]
            else:
                a0, a1 = np.float32(1) - fx, fx
                r0 = np.float32(src[y0, sx, 0] * a0) + np.float32(src[y0, sx + 1, 0] * a1)
                r1 = np.float32(src[y1, sx, 0] * a0) + np.float32(src[y1, sx + 1, 0] * a1)
            want[dy, dx] = np.float32(r0 * b0) + np.float32(r1 * b1)
    assert np.array_equal(dst[:, :, 0], want)
