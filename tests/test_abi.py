"""The C ABI library loads and exports every symbol include/ohxgb.h declares (CPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from quickchem_amd import capi
from tests import helpers


def header_symbols():
    text = open(os.path.join(helpers.ROOT, "include", "ohxgb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:XG|OHX)[A-Za-z0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.ABI_SYMBOLS)


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(helpers.PRODUCT_SO)
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_reference_bound_symbols_present_in_both_libraries():
    """The eleven symbols Shared/xgb_fortran_api.F90:19-119 binds."""
    prod, orc = C.CDLL(helpers.PRODUCT_SO), C.CDLL(helpers.ORACLE_SO)
    assert len(capi.REFERENCE_BOUND_SYMBOLS) == 11
    for name in capi.REFERENCE_BOUND_SYMBOLS:
        assert hasattr(prod, name) and hasattr(orc, name), name


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", helpers.PRODUCT_SO], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    assert "libamdhip64" in out


def test_reference_binding_module_links_against_the_product():
    """oracle/_ref holds the reference's own xgb_fortran_api.F90 compiled in place and linked,
    with a driver, against libohxgb.so: the drop-in claim at link level."""
    if not os.path.exists(helpers.DROPIN_HIP):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import subprocess
    out = subprocess.run(["ldd", helpers.DROPIN_HIP], stdout=subprocess.PIPE, text=True).stdout
    assert "libohxgb.so" in out


def test_handle_misuse_is_an_error_not_a_crash():
    lib = capi.load_library()
    out = C.c_uint64()
    assert lib.XGDMatrixNumRow(None, C.byref(out)) == -1
    assert b"invalid" in lib.XGBGetLastError()
    b = capi.Booster()
    h = b.handle
    b.free()
    assert lib.XGBoosterFree(h) == -1          # double free is caught by the magic word


def test_compute_without_gpu_fails_loudly():
    """No CPU fallback: where no HIP device is usable every compute entry point returns -1."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.OhxError, match="no CPU fallback"):
        capi.DMatrix(np.zeros((1, 27), dtype=np.float32), missing=-999.0)
    with pytest.raises(capi.OhxError, match="no CPU fallback"):
        capi.device_count()
