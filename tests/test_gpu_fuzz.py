"""Randomised end-to-end parity: seeded random model geometries (tree shape, mixtures, components, sbin,
interval, filter size, linear deformation terms) x frame sizes (odd sizes, grey and colour, barely one
octave) through the fused detect path, every candidate bit for bit against the CPU oracle."""
import numpy as np
import pytest

from partsbaseddetector_amd import synth
from partsbaseddetector_amd import model as M

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    nparts = int(rng.integers(1, 9))
    pa = [0] + [int(rng.integers(1, p + 1)) for p in range(1, nparts)]      # 1-based parents, parent < child
    sbin = int(rng.choice([4, 4, 8, 6]))
    ksize = int(rng.choice([5, 5, 5, 3, 7]))
    model = M.synthetic_model(seed=seed, pa=pa, nmix=int(rng.integers(1, 7)), ncomponents=int(rng.integers(1, 4)),
                              ksize=ksize, sbin=sbin, interval=int(rng.choice([2, 5, 10])), thresh=-1e9,
                              linear_def=bool(rng.integers(0, 2)), anchor_range=int(rng.integers(0, 7)),
                              share_filters=bool(rng.integers(0, 2)), name=f"fuzz-{seed}")
    lo = 5 * sbin + 2 * sbin            # at least one level with a non-empty feature map
    rows, cols = int(rng.integers(lo + 8, 260)), int(rng.integers(lo + 8, 330))
    cn = int(rng.choice([1, 3, 3]))
    return model, synth.synthetic_frame(seed + 1, rows, cols, cn)


@pytest.mark.parametrize("seed", range(20))
def test_random_configuration(seed):
    from partsbaseddetector_amd import detector
    from oracle import oracle
    oracle.build()
    model, im = _random_case(seed)
    flat = model.flatten()
    want = oracle.detect(flat, im)
    if len(want) > 60:                  # keep the 60 best roots: exercises the strict `>` threshold too
        model.thresh = float(np.sort(np.array([w["score"] for w in want], np.float32))[-60])
        flat = model.flatten()
        want = oracle.detect(flat, im)
    det = detector.PartsBasedDetector(device=0)
    det.distributeModel(model)
    got = det.detect(im)
    det.hd.close()
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        assert (g.level, g.component, g.root[1], g.root[0]) == (w["level"], w["component"], w["root_y"], w["root_x"])
        assert np.array_equal(g.parts, w["parts"])
        assert np.float32(g.score()) == np.float32(w["score"])
