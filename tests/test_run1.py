"""the steps either side of the predict call (feature engineering, k-slab,
tropopause mask, unit conversion) as one device-resident pass, OHXBoosterRun1."""
import ctypes as C

import numpy as np
import pytest

from oracle import xgb_oracle as O
from quickchem_amd import capi, synth
from tests import helpers


def oracle_run1(model_image, st, **kw):
    lib = helpers.oracle_lib()
    b = capi.Booster(model_buffer=model_image, lib=lib)
    out = b.run1(st, **kw)
    b.free()
    return out


@pytest.mark.parametrize("dynamic", [True, False])
def test_run1_c_oracle_vs_numpy_oracle(small_model, dynamic):
    st = helpers.run1_state((6, 5, 24))
    a = oracle_run1(small_model.image, st, dynamic_k_range=dynamic)
    b = O.run1(O.load_model(small_model.image.tobytes()), st, dynamic)
    assert (a["k1"], a["k2"]) == (b["k1"], b["k2"]) and 1 <= a["k1"] <= a["k2"] == 24
    assert np.array_equal(helpers.bits(a["ndwet"]), helpers.bits(b["ndwet"]))
    k1 = a["k1"]
    assert np.all(a["oh_boost"][:, :, :k1 - 1] == 0)
    assert helpers.ulp_diff(a["oh_boost"][:, :, k1 - 1:], b["oh_boost"][:, :, k1 - 1:]).max() <= 2     # powf vs numpy
    assert np.allclose(a["oh"], b["oh"], rtol=1e-6, atol=0)
    # the mask: above the tropopause the answer is default_OH * NDWET * 1e-6, whatever the booster said
    pl = (st["ple_mod"][:, :, :-1] + st["ple_mod"][:, :, 1:]) * np.float32(0.5)
    above = ~(pl > st["tropp_mod"][:, :, None])
    want = ((st["default_oh"] * a["ndwet"]).astype(np.float32) * np.float32(1e-6)).astype(np.float32)
    assert above.any() and np.array_equal(helpers.bits(a["oh"][above]), helpers.bits(want[above]))


def test_run1_static_slab_asserts(small_model):
    st = helpers.run1_state((4, 4, 12))
    st["tropp_mod"][2, 1] = 3000.0
    with pytest.raises(capi.OhxError, match="Minimum tropopause pressure"):
        oracle_run1(small_model.image, st, dynamic_k_range=False)


@pytest.mark.gpu
@pytest.mark.parametrize("dynamic", [True, False])
@pytest.mark.parametrize("grid", [(4, 4, 72), (37, 11, 72), (24, 30, 137)])
def test_run1_gpu_vs_oracle(deep_model, dynamic, grid):
    """Feature engineering, slab, predict and post-processing in HBM against the CPU restatement:
    engineered features and NDWET bit-exact (checked through the margins), OH within 2 ulp (10**x)."""
    st = helpers.run1_state(grid, seed=grid[0])
    want = oracle_run1(deep_model.image, st, dynamic_k_range=dynamic)
    b = capi.Booster(model_buffer=deep_model.image)
    got = b.run1(st, dynamic_k_range=dynamic)
    assert (got["k1"], got["k2"]) == (want["k1"], want["k2"])
    assert np.array_equal(helpers.bits(got["ndwet"]), helpers.bits(want["ndwet"]))
    k1 = got["k1"]
    assert np.all(got["oh_boost"][:, :, :k1 - 1] == 0)
    assert helpers.ulp_diff(got["oh_boost"][:, :, k1 - 1:], want["oh_boost"][:, :, k1 - 1:]).max() <= 2
    pl = (st["ple_mod"][:, :, :-1] + st["ple_mod"][:, :, 1:]) * np.float32(0.5)
    above = ~(pl > st["tropp_mod"][:, :, None])
    assert np.array_equal(helpers.bits(got["oh"][above]), helpers.bits(want["oh"][above]))
    assert helpers.ulp_diff(got["oh"][~above], want["oh"][~above]).max() <= 3


@pytest.mark.gpu
def test_run1_gpu_engineered_features_bit_exact(small_model):
    """The engineered features themselves: run the prep on the GPU and read the margins of a
    booster whose every split is on an engineered feature - any bit of difference in a cumulative
    sum moves a cell across a threshold somewhere in 20k cells x 20 trees."""
    grid = (40, 25, 72)
    st = helpers.run1_state(grid, seed=5)
    model = O.load_model(small_model.image.tobytes())
    ref = O.run1(model, st, True)
    b = capi.Booster(model_buffer=small_model.image)
    got = b.run1(st, dynamic_k_range=True)
    k1 = ref["k1"]
    assert got["k1"] == k1
    want_boost = ref["oh_boost"]
    assert helpers.ulp_diff(got["oh_boost"][:, :, k1 - 1:], want_boost[:, :, k1 - 1:]).max() <= 2


@pytest.mark.gpu
def test_run1_gpu_error_paths(small_model):
    st = helpers.run1_state((4, 4, 12))
    b = capi.Booster(model_buffer=small_model.image)
    st2 = dict(st)
    st2["tropp_mod"] = st["tropp_mod"].copy()
    st2["tropp_mod"][0, 0] = 100.0
    with pytest.raises(capi.OhxError, match="Minimum tropopause pressure"):
        b.run1(st2, dynamic_k_range=False)
    st3 = dict(st)
    st3["no2"] = st["no2"].copy()
    st3["no2"][1, 1, 11] = np.inf
    with pytest.raises(capi.OhxError, match="inf"):
        b.run1(st3, dynamic_k_range=True)
