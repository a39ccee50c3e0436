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
    assert lib.XGBoosterFree(h) == -1          # double free: the handle is no longer in the live table
    junk = C.create_string_buffer(b"\x4f" * 256)
    assert lib.XGBoosterFree(C.cast(junk, C.c_void_p)) == -1 and lib.XGDMatrixFree(C.cast(junk, C.c_void_p)) == -1


def test_the_product_library_exports_the_header_and_nothing_else():
    """No benchmark-input generators, no test hooks, no stray globals: every defined, non-weak symbol in the dynamic
    table of libohxgb.so - functions (T), data (D, B, R) alike - is a name of include/ohxgb.h.  Left aside: weak
    C++ template instantiations (W / V / u, which the loader merges with the host's own) and the __hip_cuid_* markers
    hipcc emits per translation unit."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", helpers.PRODUCT_SO], stdout=subprocess.PIPE, text=True).stdout
    exported = []
    for ln in out.splitlines():
        kind, name = ln.split()[-2], ln.split()[-1]
        if kind in "WwVvu" or name.startswith("__hip_cuid_"):
            continue
        exported.append(name)
    assert sorted(exported) == header_symbols()


def test_stale_handles_under_address_sanitizer(tmp_path):
    """tools/handle_abuse.c against an AddressSanitizer build of the library's host side (device code
    unsanitised; no GPU needed): double frees, handles whose block the allocator has recycled, pointers that
    never were handles - each refused with -1, none of them read."""
    import shutil
    import subprocess
    hipcc, clang = "/opt/rocm/bin/hipcc", "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip("no ROCm toolchain")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("this clang has no shared AddressSanitizer runtime")
    src = os.path.join(helpers.ROOT, "quickchem_amd", "csrc")
    lib = tmp_path / "libohxgb_asan.so"
    r = subprocess.run([hipcc, "-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                        "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-x", "hip",
                        os.path.join(src, "capi.cpp"), os.path.join(src, "kernels.hip"), "-x", "c++",
                        os.path.join(src, "forest_io.cpp"), os.path.join(src, "flatten.cpp"), "-shared", "-o", str(lib)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = tmp_path / "handle_abuse"
    r = subprocess.run([clang, "-g", "-fsanitize=address", "-shared-libsan", os.path.join(helpers.ROOT, "tools", "handle_abuse.c"),
                        f"-L{tmp_path}", "-lohxgb_asan", f"-Wl,-rpath,{tmp_path}", f"-Wl,-rpath,{os.path.dirname(rt)}",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "handle_abuse: ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_unregistering_what_was_never_registered_is_not_an_error():
    """OHXUnregisterHost (include/ohxgb.h): a caller may say "this array is going away" about any array."""
    lib = capi.load_library()
    a = np.zeros(16, dtype=np.float32)
    assert lib.OHXUnregisterHost(a.ctypes.data) == 0 and lib.OHXUnregisterHost(None) == 0


def test_compute_without_gpu_fails_loudly():
    """No CPU fallback: where no HIP device is usable every compute entry point returns -1."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.OhxError, match="no CPU fallback"):
        capi.DMatrix(np.zeros((1, 27), dtype=np.float32), missing=-999.0)
    with pytest.raises(capi.OhxError, match="no CPU fallback"):
        capi.device_count()
