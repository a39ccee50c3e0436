"""Probe for a real libxgboost: the only route by which parity can leave
"unpinned".  QuickChem pins xgboost 1.6.0 EXACT (reference Shared/CMakeLists.txt:8); neither this container
nor the GPU image ships it, so these tests normally SKIP - loudly, naming what was looked for.  Where a
libxgboost is found (any version; the version is printed) they compare, bit for bit:
  * the oracle with the real library on the hand-computed vectors, config #1 (4x4x72) and a C12 batch,
    from a model file the PRODUCT's writer wrote (legacy binary and JSON);
  * a model file the REAL library wrote back (JSON, and legacy binary where the library still writes it),
    read by the product's reader and both oracles;
  * (-m gpu) the HIP path with the real library on the same batches.
The real library is driven through the same ctypes plumbing as the product (quickchem_amd/capi.py): it
exports the very symbols the product replaces."""
import os

import numpy as np
import pytest

from oracle import real_xgboost
from oracle import xgb_oracle as O
from quickchem_amd import capi, synth
from tests import helpers


@pytest.fixture(scope="module")
def real():
    lib, where = real_xgboost.find_libxgboost()
    if lib is None:
        pytest.skip(f"NO REAL libxgboost ON THIS MACHINE - PARITY STAYS UNPINNED (set OHX_LIBXGBOOST=/path/to/libxgboost.so "
                    f"to pin it).  Looked for: {where}")
    print(f"\nreal libxgboost {real_xgboost.version_of(lib)} at {where} (the reference pins 1.6.0 EXACT)")
    return lib


def _batches(deep_model):
    cases, hand_rows = helpers.load_hand_cases()
    hand = open(os.path.join(helpers.GOLDEN, "hand_forest.json"), "rb").read()
    g4 = synth.GRIDS["mock4x4"]
    g12 = synth.GRIDS["C12"]
    return [("hand vectors", hand, hand_rows, cases["missing"]),
            ("config #1 4x4x72", deep_model.image, synth.rows_cpu(g4, 0, g4[0] * g4[1] * g4[2]), synth.XX_MISS),
            ("C12 L72", deep_model.image, synth.rows_cpu(g12, 0, g12[0] * g12[1] * g12[2]), synth.XX_MISS)]


def _real_predict(lib, path, rows, missing, option_mask=0):
    b = capi.Booster(lib=lib)
    b.load_model(path)
    d = capi.DMatrix(rows, missing=missing, lib=lib)
    out = b.predict(d, option_mask=option_mask)
    d.free()
    return b, out


def _as_file(tmp_path, name, image, fmt):
    """A model image in `fmt` written by the PRODUCT's writer (csrc/forest_io.cpp), on disk."""
    src = np.frombuffer(bytes(image), dtype=np.uint8) if not isinstance(image, np.ndarray) else image
    p = tmp_path / (name + {"binary": ".model", "json": ".json", "ubj": ".ubj"}[fmt])
    p.write_bytes(synth.convert_model(src, fmt).tobytes())
    return str(p)


def test_real_libxgboost_against_the_oracle(real, tmp_path, deep_model):
    for label, image, rows, missing in _batches(deep_model):
        want = helpers.oracle_predict(synth.convert_model(np.frombuffer(bytes(image), dtype=np.uint8), "binary"), rows, missing)
        for fmt in ("binary", "json"):
            path = _as_file(tmp_path, label.split()[0] + fmt, image, fmt)
            booster, got = _real_predict(real, path, rows, missing)
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), (label, fmt)
            leaves = booster.predict(capi.DMatrix(rows[:256], missing=missing, lib=real), option_mask=16)
            mine = helpers.oracle_predict(synth.convert_model(np.frombuffer(bytes(image), dtype=np.uint8), "binary"),
                                          rows[:256], missing, option_mask=16)
            assert np.array_equal(leaves, mine), (label, fmt, "leaf ids")
            # a file the REAL library wrote, read back by the product's reader and by both oracles
            out_json = str(tmp_path / "real_wrote.json")
            booster.save_model(out_json)
            text = open(out_json, "rb").read()
            assert np.array_equal(helpers.bits(O.predict(O.load_model(text), rows, missing=missing)), helpers.bits(want))
            again = synth.convert_model(np.frombuffer(text, dtype=np.uint8), "binary")
            assert np.array_equal(helpers.bits(helpers.oracle_predict(again, rows, missing)), helpers.bits(want))
            out_bin = str(tmp_path / "real_wrote.model")
            try:
                booster.save_model(out_bin)          # newer releases refuse the deprecated binary format
            except capi.OhxError:
                continue
            raw = np.frombuffer(open(out_bin, "rb").read(), dtype=np.uint8)
            assert np.array_equal(helpers.bits(helpers.oracle_predict(raw, rows, missing)), helpers.bits(want))
            assert capi.Booster(out_bin).info()["num_nodes"] == capi.Booster(model_buffer=image).info()["num_nodes"]


@pytest.mark.gpu
def test_real_libxgboost_against_the_hip_path(real, tmp_path, deep_model):
    import torch
    assert torch.cuda.is_available()
    for label, image, rows, missing in _batches(deep_model):
        path = _as_file(tmp_path, label.split()[0], image, "binary")
        _, want = _real_predict(real, path, rows, missing)
        b = capi.Booster(path)
        for grid in (None, (4, 4, 0)) if label.startswith("config") else (None,):
            d = capi.DMatrix(rows, missing=missing)
            if grid:
                d.set_grid(*grid)
            assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want)), label
            d.free()


def test_the_probe_reports_what_it_looked_for():
    """Runs everywhere: the search itself must not crash, and an absent library must say where it looked."""
    lib, where = real_xgboost.find_libxgboost()
    assert lib is not None or ("xgboost" in where)
