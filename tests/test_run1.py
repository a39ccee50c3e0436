"""the steps either side of the predict call (feature engineering, k-slab,
tropopause mask, unit conversion) as one device-resident pass, OHXBoosterRun1."""
import ctypes as C
import os

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
@pytest.mark.parametrize("grid,pieces,dynamic", [((37, 11, 72), 2, True), ((24, 30, 137), 3, False), ((40, 53, 72), 5, True),
                                                 ((144, 200, 72), 3, True), ((144, 200, 72), 4, False)])
def test_run1_in_pieces_is_run1(deep_model, grid, pieces, dynamic):
    """(r5) OH Run1 can walk a slab in ranges of j - the next range's feature engineering and the last one's mask and
    unit conversion on a second stream beside the walk (capi.cpp run1_device; never by itself - one piece is the default,
    pieces measured slower - but on request: ohx_run1_pieces > 1, here on grids whose j extent is no multiple of the
    brick's four, small enough for the trees-split-over-
    waves form and big enough for the ring kernel): every output, the DIAG dumps included, is what one piece gives, bit for
    bit - and that is checked against the oracle; with -999.0 and NaN among the inputs (the rows that hold them go
    through each piece's own second launch)."""
    st = helpers.run1_state(grid, seed=grid[1])
    rng = np.random.default_rng(3)
    for name in ("no2", "cloud", "ch2o"):
        a = st[name] = st[name].copy()
        mask = rng.random(a.shape) < 3e-4
        a[mask] = np.where(rng.random(int(mask.sum())) < 0.5, np.float32(-999.0), np.float32(np.nan))
    want = oracle_run1(deep_model.image, st, dynamic_k_range=dynamic, want_diag=True)
    outs = []
    for n in (1, pieces):
        b = capi.Booster(model_buffer=deep_model.image)
        b.set_param("ohx_kernel", "ring")
        b.set_param("ohx_run1_pieces", n)
        outs.append(b.run1(st, dynamic_k_range=dynamic, want_diag=True))
        b.free()
    one, many = outs
    assert (one["k1"], one["k2"]) == (many["k1"], many["k2"]) == (want["k1"], want["k2"])
    for name in one:
        if name in ("k1", "k2"):
            continue
        assert np.array_equal(helpers.bits(many[name]), helpers.bits(one[name])), name
    assert np.array_equal(helpers.bits(many["ndwet"]), helpers.bits(want["ndwet"]))
    k1 = many["k1"]
    assert np.all(many["oh_boost"][:, :, :k1 - 1] == 0)
    assert helpers.ulp_diff(many["oh_boost"][:, :, k1 - 1:], want["oh_boost"][:, :, k1 - 1:]).max() <= 2
    assert helpers.ulp_diff(many["oh"], want["oh"]).max() <= 3
    for name in ("diag_tauclwdn", "diag_taucliup", "diag_aodup", "diag_aoddn", "diag_aod", "diag_pl_bst", "diag_strato3"):
        assert np.array_equal(helpers.bits(many[name]), helpers.bits(want[name])), name


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


# ---- solar geometry (OH_GridCompMod.F90:401-466, 1444, 1905-1970) ----

JDAY_CASES = {20240101: 1, 20240229: 60, 20240301: 61, 20230301: 60, 21000301: 60, 20001231: 366, 19001231: 365,
              20230704: 185, 20241231: 366, 20230131: 31}   # worked out by hand from the calendar


def _lonlat(im=72, jm=37):
    lon = np.linspace(-np.pi, np.pi, im, endpoint=False, dtype=np.float32) + np.float32(0.01)
    lat = np.linspace(-np.pi / 2, np.pi / 2, jm, dtype=np.float32)
    lons, lats = np.meshgrid(lon, lat, indexing="ij")
    # also longitudes on the 0..2pi convention, which the reference folds back (:441-442)
    lons[::3, :] += np.float32(2 * np.pi)
    return lats.astype(np.float32), lons.astype(np.float32)


def _sza_tolerance(jday, lats):
    """acos near +-1: one ulp of cosz is worth sqrt(2*6e-8) rad = 0.02 degree; elsewhere 1e-3 degree."""
    dec = np.degrees(np.arcsin(0.3978 * np.sin(0.9863 * (jday - 80.0) * np.pi / 180.0)))
    near = np.abs(np.degrees(lats.astype(np.float64)) - dec) < 1.0
    return np.where(near, 0.05, 2e-3)


def test_solar_geometry_oracles_agree_and_match_the_closed_form(oracle_lib):
    import ctypes as C
    for nymd, want in JDAY_CASES.items():
        assert O.julian_day(nymd) == want and oracle_lib.oracle_julian_day(nymd) == want, nymd
    lats, lons = _lonlat()
    for jday in (1, 80, 172, 266, 355, 366):
        lat_np, sza_np = O.solar_geometry(jday, lats, lons, capi.DEG2RAD, capi.RAD2DEG)
        la, lo = np.ascontiguousarray(lats.T), np.ascontiguousarray(lons.T)
        lat_c, sza_c = np.empty_like(la), np.empty_like(la)
        oracle_lib.oracle_solar_geometry.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                                     C.c_float, C.c_void_p, C.c_void_p]
        assert oracle_lib.oracle_solar_geometry(jday, la.ctypes.data, lo.ctypes.data, lats.shape[0], lats.shape[1],
                                                float(capi.DEG2RAD), float(capi.RAD2DEG), lat_c.ctypes.data,
                                                sza_c.ctypes.data) == 0
        assert np.array_equal(lat_c.T, lat_np)                       # one float32 multiply
        tol = _sza_tolerance(jday, lats)
        assert (np.abs(sza_c.T.astype(np.float64) - sza_np) <= tol).all()
        # local noon: the hour angle is zero, so the zenith angle is |latitude - declination|
        dec = np.degrees(np.arcsin(0.3978 * np.sin(0.9863 * (jday - 80.0) * np.pi / 180.0)))
        closed = np.abs(np.degrees(lats.astype(np.float64)) - dec)
        assert (np.abs(sza_np - closed) <= 0.06).all()


@pytest.mark.gpu
def test_solar_geometry_gpu_against_the_oracle():
    """Tolerance, stated: float32 trig of the device library against the host's - 2e-3 degree, 0.05 degree
    within one degree of the subsolar latitude (acos near 1); LAT in degrees is one multiply, bit-exact."""
    for nymd, want in JDAY_CASES.items():
        assert capi.julian_day(nymd) == want
    lats, lons = _lonlat(144, 91)
    for jday in (1, 80, 172, 266, 366):
        lat_deg, sza = capi.solar_geometry(jday, lats, lons)
        lat_np, sza_np = O.solar_geometry(jday, lats, lons, capi.DEG2RAD, capi.RAD2DEG)
        assert np.array_equal(lat_deg, lat_np)
        assert (np.abs(sza.astype(np.float64) - sza_np) <= _sza_tolerance(jday, lats)).all()
        assert sza.min() >= 0.0 and sza.max() <= 180.0
    lib = capi.load_library()
    with pytest.raises(capi.OhxError, match="LATS"):
        capi.check(lib, lib.OHXSolarGeometry(80, None, None, 4, 4, 0.0, 0.0, None, None))


def test_run1_fortran_host_on_the_oracle(tmp_path, small_model):
    """quickchem_amd/fortran/oh_run1.F90 (bind(C) mirror of OHXRun1Args, the reference's variable names)
    driving OHXBoosterRun1: same library underneath, so the Fortran host must reproduce the Python
    host's result to the bit."""
    grid = (9, 7, 24)
    st = helpers.run1_state(grid, seed=3)
    state, model, out = tmp_path / "run1.bin", tmp_path / "oh.model", tmp_path / "out.bin"
    helpers.write_run1_state_file(state, st, True)
    model.write_bytes(small_model.image.tobytes())
    r = helpers.run_driver(helpers.RUN1_DRIVER_ORACLE, state, model, out)
    assert r.returncode == 0, r.stdout
    rc, k1, k2, oh, boost, ndwet = helpers.read_run1_output(out, *grid)
    ref = oracle_run1(small_model.image, st, dynamic_k_range=True)
    assert rc == 0 and (k1, k2) == (ref["k1"], ref["k2"])
    for got, name in ((oh, "oh"), (boost, "oh_boost"), (ndwet, "ndwet")):
        assert np.array_equal(helpers.bits(got), helpers.bits(ref[name])), name


@pytest.mark.gpu
def test_run1_fortran_host_on_the_gpu(tmp_path, small_model):
    grid = (40, 25, 72)
    st = helpers.run1_state(grid, seed=9)
    state, model = tmp_path / "run1.bin", tmp_path / "oh.model"
    helpers.write_run1_state_file(state, st, True)
    model.write_bytes(small_model.image.tobytes())
    outs = {}
    for tag, exe in (("gpu", helpers.RUN1_DRIVER_HIP), ("cpu", helpers.RUN1_DRIVER_ORACLE)):
        out = tmp_path / f"{tag}.bin"
        r = helpers.run_driver(exe, state, model, out)
        assert r.returncode == 0, r.stdout
        outs[tag] = helpers.read_run1_output(out, *grid)
    (rc, k1, k2, oh, boost, ndwet), (rc2, k1c, k2c, oh_c, boost_c, ndwet_c) = outs["gpu"], outs["cpu"]
    assert rc == 0 and rc2 == 0 and (k1, k2) == (k1c, k2c)
    assert np.array_equal(helpers.bits(ndwet), helpers.bits(ndwet_c))
    assert helpers.ulp_diff(boost[:, :, k1 - 1:], boost_c[:, :, k1 - 1:]).max() <= 2
    assert helpers.ulp_diff(oh, oh_c).max() <= 3
    # and the Python host on the same library: identical
    b = capi.Booster(model_buffer=small_model.image)
    py = b.run1(st, dynamic_k_range=True)
    assert np.array_equal(helpers.bits(oh), helpers.bits(py["oh"])) and (k1, k2) == (py["k1"], py["k2"])


def test_post_process_and_diag_dumps_of_the_oracle():
    """OHXOHPostProcess (the tail of Run1 on a tick that skips Boost) and the DIAG_* dumps, oracle against plain
    numpy float32 expressions in the reference's order (:1247-1257, 1444-1478, 1579-1595)."""
    grid = (5, 4, 18)
    st = helpers.run1_state(grid, seed=8)
    f32 = np.float32
    oh_ml = (st["default_oh"] * f32(3.0)).astype(f32)
    eps = float(f32(18.015) / f32(28.965))
    oh, ndwet = capi.oh_post_process(st["ple_mod"], st["t_mod"], st["q_mod"], st["tropp_mod"], st["default_oh"], oh_ml,
                                     epsilon=eps, lib=helpers.oracle_lib())
    pl = ((st["ple_mod"][:, :, :-1] + st["ple_mod"][:, :, 1:]) * f32(0.5)).astype(f32)
    tv = (st["t_mod"] * (f32(1.0) + st["q_mod"] / f32(eps)) / (f32(1.0) + st["q_mod"])).astype(f32)
    nd = ((f32(6.023e26) * pl) / (f32(8314.47) * tv)).astype(f32)
    pick = np.where(pl > st["tropp_mod"][:, :, None], oh_ml, st["default_oh"]).astype(f32)
    assert np.array_equal(helpers.bits(ndwet), helpers.bits(nd))
    assert np.array_equal(helpers.bits(oh), helpers.bits(((pick * nd).astype(f32) * f32(1e-6)).astype(f32)))


def test_diag_dumps_of_the_oracle(small_model):
    grid = (5, 4, 18)
    st = helpers.run1_state(grid, seed=8)
    f32 = np.float32
    out = oracle_run1(small_model.image, st, dynamic_k_range=True, want_diag=True)
    km = grid[2]
    thick = (st["zle_bst"][:, :, :-1] - st["zle_bst"][:, :, 1:]).astype(f32)
    sc = st["scacoef"][0] + st["scacoef"][1]
    for a in st["scacoef"][2:]:
        sc = (sc + a).astype(f32)
    aod = (thick * sc).astype(f32)
    assert np.array_equal(helpers.bits(out["diag_aod"]), helpers.bits(aod))
    up = np.zeros(grid, f32)
    run = np.zeros(grid[:2], f32)
    for k in range(km):
        run = (run + st["tauclw"][:, :, k]).astype(f32)
        up[:, :, k] = run
    assert np.array_equal(helpers.bits(out["diag_tauclwup"]), helpers.bits(up))
    dn = np.zeros(grid, f32)
    for k in range(km):
        s = np.zeros(grid[:2], f32)
        for kk in range(k, km):
            s = (s + aod[:, :, kk]).astype(f32)
        dn[:, :, k] = s
    assert np.array_equal(helpers.bits(out["diag_aoddn"]), helpers.bits(dn))
    assert np.array_equal(out["diag_strato3"], (st["gmito3"] - st["gmitto3"]).astype(f32))
    assert np.array_equal(out["diag_pl_bst"], ((st["ple_bst"][:, :, :-1] + st["ple_bst"][:, :, 1:]) * f32(0.5)).astype(f32))


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [(4, 4, 72), (37, 11, 40)])
def test_post_process_and_diag_dumps_on_the_gpu(small_model, grid):
    """The same two on the MI355X against the oracle: every dump and the post-processed OH bit for bit."""
    st = helpers.run1_state(grid, seed=grid[1])
    eps = float(np.float32(18.015) / np.float32(28.965))
    want = oracle_run1(small_model.image, st, dynamic_k_range=True, epsilon=eps, want_diag=True)
    b = capi.Booster(model_buffer=small_model.image)
    got = b.run1(st, dynamic_k_range=True, epsilon=eps, want_diag=True)
    for name in capi.RUN1_DIAG_3D + ["diag_strato3", "ndwet"]:
        assert np.array_equal(helpers.bits(got[name]), helpers.bits(want[name])), name
    # asking for the dumps changes nothing else
    plain = b.run1(st, dynamic_k_range=True, epsilon=eps)
    assert np.array_equal(helpers.bits(plain["oh"]), helpers.bits(got["oh"]))
    oh_ml = want["oh_boost"]
    a = capi.oh_post_process(st["ple_mod"], st["t_mod"], st["q_mod"], st["tropp_mod"], st["default_oh"], oh_ml, epsilon=eps)
    c = capi.oh_post_process(st["ple_mod"], st["t_mod"], st["q_mod"], st["tropp_mod"], st["default_oh"], oh_ml, epsilon=eps,
                             lib=helpers.oracle_lib())
    assert np.array_equal(helpers.bits(a[0]), helpers.bits(c[0])) and np.array_equal(helpers.bits(a[1]), helpers.bits(c[1]))
    with pytest.raises(capi.OhxError, match="NULL"):
        capi.check(b.lib, b.lib.OHXOHPostProcess(4, 4, 4, 1.0, 1.0, 1.0, None, None, None, None, None, None, None, None))


@pytest.mark.gpu
def test_host_forms_with_registered_arrays(small_model):
    """ohx_register_host: the host forms of OHXBoosterRun1, OHXOHPostProcess and OHXBoosterPredictFields on arrays that
    stay at their addresses from tick to tick (as MAPL's do) - registered with the driver at the first tick, gathered
    into one copy launch from then on - give what they give on pageable arrays, bit for bit, tick after tick; arrays
    that are not 16-byte aligned take the scalar lanes of the copy kernel; OHXReleaseScratch() forgets the
    registrations and the next tick makes them again."""
    import torch
    torch.cuda.set_device(0)
    grid = (37, 11, 40)
    st = helpers.run1_state(grid, seed=3)
    b = capi.Booster(model_buffer=small_model.image)
    call = b.run1_prepare(st, dynamic_k_range=True, want_diag=True)
    # one input at an address that is 4 but not 16 bytes aligned
    shifted = np.zeros(call["keep"]["co"].size + 1, dtype=np.float32)[1:]
    shifted[:] = call["keep"]["co"].reshape(-1)
    assert shifted.ctypes.data % 16 != 0
    call["keep"]["co_shifted"] = shifted
    call["args"].co = shifted.ctypes.data
    plain = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.run1_call(call).items()}
    b.set_param("ohx_register_host", "1")
    try:
        for tick in range(3):
            got = b.run1_call(call)
            for k, v in plain.items():
                if isinstance(v, np.ndarray):
                    assert np.array_equal(helpers.bits(got[k]), helpers.bits(v)), (tick, k)
                else:
                    assert got[k] == v
            if tick == 1:
                capi.check(b.lib, b.lib.OHXReleaseScratch())
        # (r5) one array leaves on its own - the caller is about to free it - and comes back: the tick after registers it
        # again; an array the library never saw, and NULL, are not errors; an array passed twice (the model's T is also
        # Boost's T under ONLINE_INST) crosses once and changes nothing
        capi.check(b.lib, b.lib.OHXUnregisterHost(call["keep"]["t_mod"].ctypes.data))
        capi.check(b.lib, b.lib.OHXUnregisterHost(np.zeros(8, dtype=np.float32).ctypes.data))
        capi.check(b.lib, b.lib.OHXUnregisterHost(None))
        got = b.run1_call(call)
        assert np.array_equal(helpers.bits(got["oh"]), helpers.bits(plain["oh"]))
        same_t = b.run1_prepare(dict(st, t_bst=st["t_mod"], qv=st["q_mod"], ple_bst=st["ple_mod"]), dynamic_k_range=True)
        same_t["args"].t_bst = same_t["args"].t_mod
        same_t["args"].qv = same_t["args"].q_mod
        same_t["args"].ple_bst = same_t["args"].ple_mod
        aliased = b.run1_call(same_t)
        want_aliased = oracle_run1(small_model.image, dict(st, t_bst=st["t_mod"], qv=st["q_mod"], ple_bst=st["ple_mod"]),
                                   dynamic_k_range=True)
        assert np.array_equal(helpers.bits(aliased["ndwet"]), helpers.bits(want_aliased["ndwet"]))
        assert helpers.ulp_diff(aliased["oh"], want_aliased["oh"]).max() <= 3
        # the kernels' flag words come back on the copy-back launch of a registered tick: an inf among the inputs is still an
        # error (xgboost's "Input data contains `inf` or `nan`"), and the tick after it is clean again
        no2 = call["keep"]["no2"]
        spot = np.unravel_index(no2.size - 3, no2.shape)          # (the bottom level: inside every slab)
        was = no2[spot]
        no2[spot] = np.inf
        with pytest.raises(capi.OhxError, match="inf"):
            b.run1_call(call)
        no2[spot] = was
        got = b.run1_call(call)
        assert np.array_equal(helpers.bits(got["oh"]), helpers.bits(plain["oh"]))
        # the post-processing of a tick that skips Boost, host form
        im, jm, km = grid
        flat = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32).T)      # noqa: E731
        ins = [flat(st[k]) for k in ("ple_mod", "t_mod", "q_mod", "tropp_mod", "default_oh")] + [flat(plain["oh_boost"])]
        outs = {}
        for mode in ("1", "0"):
            b.set_param("ohx_register_host", mode)
            oh, nd = np.zeros(im * jm * km, dtype=np.float32), np.zeros(im * jm * km, dtype=np.float32)
            for _ in range(2):
                capi.check(b.lib, b.lib.OHXOHPostProcess(im, jm, km, 6.023e26, 8314.47, 18.015 / 28.965,
                                                         *[a.ctypes.data for a in ins], oh.ctypes.data, nd.ctypes.data))
            outs[mode] = (oh, nd)
        assert np.array_equal(helpers.bits(outs["1"][0]), helpers.bits(outs["0"][0]))
        assert np.array_equal(helpers.bits(outs["1"][1]), helpers.bits(outs["0"][1]))
        # the fused call, host form
        g2 = synth.GRIDS["C12"]
        pl, tropp, fields = helpers.synth_state(g2)
        ff = [np.ascontiguousarray(f.T) for f in fields]
        res = {}
        for mode in ("1", "0"):
            b.set_param("ohx_register_host", mode)
            oh = np.zeros(g2[0] * g2[1] * g2[2], dtype=np.float32)
            margin = np.zeros(g2[0] * g2[1] * (g2[2] - 4), dtype=np.float32)
            for _ in range(2):
                b.predict_fields(ff, synth.IS2D, synth.PL_FEATURE, *g2, 5, g2[2], synth.XX_MISS, oh, ohscale=0.85, margin=margin)
            res[mode] = (oh, margin)
        # (registered, a rank-sized block's k-slab of every field and the outputs cross in one copy launch each way)
        assert np.array_equal(helpers.bits(res["1"][0]), helpers.bits(res["0"][0])) and np.any(res["1"][0] != 0)
        assert np.array_equal(helpers.bits(res["1"][1]), helpers.bits(res["0"][1])) and np.any(res["1"][1] != 0)
        assert np.all(res["1"][0][:g2[0] * g2[1] * 4] == 0)              # the levels above the slab are not touched
    finally:
        b.set_param("ohx_register_host", "0")
        b.free()


@pytest.mark.gpu
def test_two_oh_instances_tick_from_two_threads(small_model, deep_model):
    """Two OH instances in one process (QuickChem_GridComp.rc:22), each driven by its own thread: OHXBoosterRun1's host form
    on registered arrays, twenty ticks each on blocks of different size.  The two boosters share the library's two streams,
    the host registry and the copy kernel; every tick of either must be what the same booster gives alone."""
    import threading
    jobs = []
    for k, (model, grid) in enumerate(((small_model, (37, 11, 40)), (deep_model, (48, 24, 72)))):
        st = helpers.run1_state(grid, seed=40 + k)
        b = capi.Booster(model_buffer=model.image)
        call = b.run1_prepare(st, dynamic_k_range=True, want_boost=True, want_ndwet=True)
        alone = {key: v.copy() for key, v in b.run1_call(call).items() if isinstance(v, np.ndarray)}
        jobs.append((b, call, alone))
    jobs[0][0].set_param("ohx_register_host", "1")          # (process-wide)
    errors = []

    def run(b, call, alone):
        try:
            for tick in range(20):
                got = b.run1_call(call)
                for key, v in alone.items():
                    if not np.array_equal(helpers.bits(got[key]), helpers.bits(v)):
                        errors.append((tick, key))
        except Exception as e:                      # noqa: BLE001 - reported by the assert below
            errors.append(repr(e))
    try:
        threads = [threading.Thread(target=run, args=j) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        jobs[0][0].set_param("ohx_register_host", "0")
        capi.check(jobs[0][0].lib, jobs[0][0].lib.OHXReleaseScratch())
    assert errors == []


KNOB_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from quickchem_amd import capi, synth
from tests import helpers
model = synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"])
b = capi.Booster(model_buffer=model.image)
st = helpers.run1_state((37, 11, 40), seed=3)
call = b.run1_prepare(st, dynamic_k_range=True, want_boost=True, want_ndwet=True, want_diag=True)
plain = {k: v.copy() for k, v in b.run1_call(call).items() if isinstance(v, np.ndarray)}
import os
engine = os.environ.get("OHX_TEST_COPY_ENGINE")
if engine:
    b.set_param("ohx_copy_engine", engine)
b.set_param("ohx_register_host", "1")
# (auto: two ticks of warm-up, then K K K K D D D D K K K K D D D D and a verdict - 24 ticks see every phase of it)
for tick in range(24 if engine == "auto" else 4):
    got = b.run1_call(call)
    for k, v in plain.items():
        assert np.array_equal(got[k].view(np.uint32), v.view(np.uint32)), (tick, k)
choice, trials, picked = b.copy_engine_choice()
if engine == "auto":
    assert choice in (0, 1) and trials == 1 and picked == choice, (choice, trials, picked)
else:
    assert choice == -1 or engine is None, (engine, choice)
b.set_param("ohx_register_host", "0")
print("KNOBS_OK")
"""


@pytest.mark.gpu
def test_the_host_tick_under_each_of_its_experiment_knobs():
    """The knobs of the host form's tick (include/ohxgb.h part 3; each read once per process, hence a child per setting):
    whatever order the lists cross in, whoever waits for them and however they are moved, a registered tick gives what a
    pageable one gives, bit for bit."""
    import subprocess
    import sys
    for env in ({"OHX_RUN1_GATE": "1"}, {"OHX_RUN1_SLAB_IN_PLACE": "1"}, {"OHX_RUN1_GATE": "1", "OHX_RUN1_SLAB_IN_PLACE": "1"},
                {"OHX_RUN1_STREAMS": "1"}, {"OHX_COPY_POST_BLOCKS": "0"}, {"OHX_COPY_POST_BLOCKS": "8", "OHX_COPY_BACK_BLOCKS": "0"},
                {"OHX_RUN1_STREAMS": "1", "OHX_RUN1_SLAB_IN_PLACE": "1", "OHX_COPY_BACK_BLOCKS": "32"},
                # (r6) ohx_copy_engine: every list and the copy-back by the DMA engines, by copy kernels, and auto's trial
                {"OHX_TEST_COPY_ENGINE": "dma"}, {"OHX_TEST_COPY_ENGINE": "kernel"}, {"OHX_TEST_COPY_ENGINE": "auto"},
                {"OHX_TEST_COPY_ENGINE": "dma", "OHX_RUN1_STREAMS": "1"}):
        r = subprocess.run([sys.executable, "-c", KNOB_CHILD, helpers.ROOT], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=600, cwd=helpers.ROOT)
        assert r.returncode == 0 and "KNOBS_OK" in r.stdout, (env, r.stdout[-1500:], r.stderr[-3000:])
