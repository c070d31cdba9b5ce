"""Pins the plain-C oracle (oracle/gs_oracle.c: double precision, explicit eigenvectors, true back-to-front backward)
to the fixtures the real reference produced.  Its analytic backward is independent of both torch autograd and the
product's front-to-back formulation."""
import numpy as np
import pytest

from oracle import c_oracle
from tests import util


def _run(d, grad=True):
    return c_oracle.render(d, d["H"], d["W"], d["fx"], d["fy"], d["cx"], d["cy"], grad_image=d["wrand"] if grad else None, **d["kwargs"])


@pytest.mark.parametrize("name", util.RENDER_CASES)
def test_image_and_gradients(name):
    d = util.load(name)
    st, img, g, (V, P) = _run(d)
    assert st == 0 and V == len(d["im_ids"]) and P == len(d["im_pair_gauss"])
    assert np.abs(img - d["image"]).max() < 1e-13
    for k in util.PARAMS:
        ref = d["grad_" + k]
        assert np.abs(g[k] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (k, np.abs(g[k] - ref).max())


@pytest.mark.parametrize("name", util.EMPTY_CASES)
def test_empty(name):
    d = util.load(name)
    st, img, g, _ = _run(d)
    assert st == 10 and np.abs(img).max() == 0 and all(np.abs(v).max() == 0 for v in g.values())


def test_offscreen_status():
    d = util.load("g10_offscreen")
    assert _run(d, grad=False)[0] == 11
