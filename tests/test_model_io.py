"""Host logic of XGBoosterLoadModel/SaveModel: legacy binary <-> JSON (CPU)."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import xgb_oracle as O
from quickchem_amd import capi, synth
from tests import helpers


def test_binary_json_round_trip(small_model):
    js = synth.convert_model(small_model.image, "json")
    back = synth.convert_model(js, "binary")
    assert np.array_equal(back, small_model.image)
    doc = json.loads(js.tobytes())
    assert doc["learner"]["gradient_booster"]["name"] == "gbtree"
    assert int(doc["learner"]["learner_model_param"]["num_feature"]) == 27
    assert len(doc["learner"]["gradient_booster"]["model"]["trees"]) == small_model.num_trees


def test_json_and_binary_predict_the_same(small_model):
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 2048)
    js = synth.convert_model(small_model.image, "json")
    a = O.predict(O.load_model(small_model.image.tobytes()), rows, missing=synth.XX_MISS)
    b = O.predict(O.load_model(js.tobytes()), rows, missing=synth.XX_MISS)
    assert np.array_equal(helpers.bits(a), helpers.bits(b))


def test_binary_layout_sizes(small_model):
    """136-byte learner param, two length-prefixed names, 160-byte gbtree param, then per tree
    148 + 36 * num_nodes bytes, then tree_info."""
    img = small_model.image.tobytes()
    assert img[:4] == b"binf"
    off = 4 + 136
    n, = struct.unpack_from("<Q", img, off)
    assert img[off + 8:off + 8 + n] == b"reg:squarederror"
    off += 8 + n
    n, = struct.unpack_from("<Q", img, off)
    assert img[off + 8:off + 8 + n] == b"gbtree"
    off += 8 + n
    num_trees, = struct.unpack_from("<i", img, off)
    assert num_trees == small_model.num_trees
    off += 160
    total = 0
    for _ in range(num_trees):
        num_nodes = struct.unpack_from("<37i", img, off)[1]
        total += num_nodes
        off += 148 + 36 * num_nodes
    assert total == small_model.num_nodes
    assert off + 4 * num_trees == len(img)


def test_file_round_trip_through_the_abi(tmp_path, small_model):
    """XGBoosterLoadModel / XGBoosterSaveModel need no GPU: host logic only."""
    p_bin, p_json, p_bin2 = tmp_path / "m.model", tmp_path / "m.json", tmp_path / "m2.bin"
    p_bin.write_bytes(small_model.image.tobytes())
    b = capi.Booster(str(p_bin))
    b.save_model(str(p_json))
    b2 = capi.Booster(str(p_json))
    b2.save_model(str(p_bin2))
    assert p_bin2.read_bytes() == small_model.image.tobytes()
    info = b2.info()
    assert info["num_trees"] == small_model.num_trees and info["num_nodes"] == small_model.num_nodes
    assert info["num_slots"] >= info["num_nodes"] // 3 and info["packed"] == 2 and info["num_feature"] == 27
    b2.set_param("ohx_kernel", "packed2")
    info = b2.info()
    assert info["num_slots"] >= info["num_nodes"] and info["packed"] == 1


@pytest.mark.parametrize("mutation", ["truncate", "bad_child", "ubj", "garbage_json", "nonadjacent"])
def test_malformed_models_fail_loudly(tmp_path, small_model, mutation):
    img = bytearray(small_model.image.tobytes())
    path = tmp_path / "m.model"
    if mutation == "truncate":
        img = img[:len(img) // 2]
    elif mutation == "bad_child":
        # first tree, root node: cleft far out of range
        off = 4 + 136 + 8 + 16 + 8 + 6 + 160 + 148
        struct.pack_into("<i", img, off + 4, 10**8)
        struct.pack_into("<i", img, off + 8, 10**8 + 1)
    elif mutation == "nonadjacent":
        off = 4 + 136 + 8 + 16 + 8 + 6 + 160 + 148
        l, r = struct.unpack_from("<ii", img, off + 4)
        struct.pack_into("<ii", img, off + 4, r, l)          # right == left - 1
    elif mutation == "ubj":
        path = tmp_path / "m.ubj"
    elif mutation == "garbage_json":
        path = tmp_path / "m.json"
        img = bytearray(b'{"learner": {"oops": 1}}')
    path.write_bytes(bytes(img))
    b = capi.Booster()
    with pytest.raises(capi.OhxError):
        b.load_model(str(path))


def test_missing_file_and_unloaded_booster(tmp_path):
    b = capi.Booster()
    with pytest.raises(capi.OhxError, match="cannot open"):
        b.load_model(str(tmp_path / "nope.model"))
    with pytest.raises(capi.OhxError, match="no model"):
        b.save_model(str(tmp_path / "x.model"))


def test_layout_parameters_change_slots_not_nodes(deep_model):
    b = capi.Booster(model_buffer=deep_model.image)
    sup = b.info()
    assert sup["packed"] == 2 and sup["node_bytes"] == 16 * sup["num_slots"] and sup["num_slots"] % 4 == 0
    b.set_param("ohx_kernel", "packed2")
    base = b.info()
    b.set_param("ohx_line_slots", 0)
    bfs = b.info()
    assert bfs["num_nodes"] == base["num_nodes"] == deep_model.num_nodes
    assert bfs["num_slots"] == bfs["num_nodes"]                 # breadth-first only: no padding
    assert base["num_slots"] < 1.35 * base["num_nodes"]         # line packing wastes little
    assert base["max_depth"] == deep_model.max_depth == 18
    b.set_param("ohx_kernel", "wide")
    assert b.info()["packed"] == 0 and b.info()["node_bytes"] == 16 * bfs["num_slots"]
    with pytest.raises(capi.OhxError):
        b.set_param("ohx_kernel", "nope")


def test_format_selection_falls_back_when_a_tree_is_too_big_for_super_nodes():
    """The super-node format addresses at most 16 384 grandchild groups per tree; a bigger tree makes
    the booster fall back to the packed 8-byte format (host logic, no GPU needed to see it)."""
    big = synth.make_model(num_trees=2, max_depth=24, sample_log2=20, min_leaf=1, grid=synth.GRIDS["C48"])
    b = capi.Booster(model_buffer=big.image)
    info = b.info()
    assert big.num_nodes > 3 * 65536 and info["packed"] == 1, (big.num_nodes, info)
    small = synth.make_model(num_trees=2, max_depth=8, sample_log2=14, min_leaf=4, grid=synth.GRIDS["C12"])
    assert capi.Booster(model_buffer=small.image).info()["packed"] == 2


def test_ubjson_round_trip_and_predictions(small_model, tmp_path):
    """UBJSON (what xgboost >= 1.6 writes for ".ubj"): product writer -> product reader and the oracle's
    own independent reader; same trees, same predictions, byte-identical legacy image after the trip."""
    ubj = synth.convert_model(small_model.image, "ubj")
    assert ubj[:2].tobytes() == b"{L" and len(ubj) < len(synth.convert_model(small_model.image, "json"))
    assert np.array_equal(synth.convert_model(ubj, "binary"), small_model.image)
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 2048)
    a = O.predict(O.load_model(small_model.image.tobytes()), rows, missing=synth.XX_MISS)
    b = O.predict(O.load_model(ubj.tobytes()), rows, missing=synth.XX_MISS)
    assert np.array_equal(helpers.bits(a), helpers.bits(b))
    # through the ABI, by file extension
    p_ubj, p_bin = tmp_path / "m.ubj", tmp_path / "m.bin"
    bst = capi.Booster(model_buffer=small_model.image)
    bst.save_model(str(p_ubj))
    assert p_ubj.read_bytes() == ubj.tobytes()
    capi.Booster(str(p_ubj)).save_model(str(p_bin))
    assert p_bin.read_bytes() == small_model.image.tobytes()
    # a hand-made document with standard small-int markers and unoptimised arrays is read too
    doc = (b'{i\x07learner{i\x13learner_model_param{i\nbase_scoreSi\x045E-1i\x0bnum_featureSi\x011i\tnum_classSi\x010}'
           b'i\tobjective{i\x04nameSi\x10reg:squarederror}'
           b'i\x10gradient_booster{i\x04nameSi\x06gbtreei\x05model{i\x12gbtree_model_param{i\tnum_treesSi\x011}'
           b'i\ttree_info[i\x00]i\x05trees[{i\ntree_param{i\tnum_nodesSi\x013i\x0bnum_featureSi\x011}'
           b'i\rleft_children[i\x01i\xffi\xff]i\x0eright_children[i\x02i\xffi\xff]i\x07parents[l\x7f\xff\xff\xffi\x00i\x00]'
           b'i\rsplit_indices[i\x00i\x00i\x00]i\x10split_conditions[d\x3f\x80\x00\x00d\xbf\x00\x00\x00d\x40\x00\x00\x00]'
           b'i\x0cdefault_left[TFF]}]}}}}')
    m = O.load_model(doc)
    assert np.array_equal(O.predict(m, np.float32([[0.5], [1.0], [np.nan]])), np.float32([0.0, 2.5, 0.0]))
    got = helpers.oracle_predict(synth.convert_model(doc, "binary"), np.float32([[0.5], [1.0], [np.nan]]), float("nan"))
    assert np.array_equal(got, np.float32([0.0, 2.5, 0.0]))


def test_model_readers_survive_mutated_files_under_sanitizers(tmp_path, small_model):
    """tools/fuzz_models.cpp, built with AddressSanitizer + UBSan (CPU only): thousands of truncated, bit-flipped,
    spliced legacy-binary / JSON / UBJSON images go through the readers and the flatteners; each is either
    accepted or refused with an error - no crash, no out-of-bounds access, no runaway allocation."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "quickchem_amd", "csrc")
    exe = tmp_path / "fuzz_models"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                            "-fno-sanitize-recover=undefined", "-fopenmp", "-I", src,
                            os.path.join(root, "tools", "fuzz_models.cpp"), os.path.join(src, "forest_io.cpp"),
                            os.path.join(src, "flatten.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    seed_model = synth.make_model(num_trees=6, max_depth=6, sample_log2=12, min_leaf=2, grid=synth.GRIDS["C12"])
    model = tmp_path / "seed.model"
    model.write_bytes(seed_model.image.tobytes())
    for seed in (11, 12):
        r = subprocess.run([str(exe), str(model), "4000", str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "0 crashes" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def _hand_doc():
    return json.load(open(os.path.join(helpers.GOLDEN, "hand_forest.json")))


def _trees_of(image):
    doc = json.loads(synth.convert_model(np.frombuffer(image, dtype=np.uint8), "json").tobytes())
    return doc, doc["learner"]["gradient_booster"]["model"]["trees"]


@pytest.mark.parametrize("variant", ["v16_plain", "v16_binf_attrs_metrics", "v081_bin"])
def test_legacy_files_from_an_independent_writer(variant):
    """The production models are legacy-binary files written by XGBoost 1.6.0 (".model") and by 0.81 (".bin")
    (OH_instance_OH.rc:17-20).  No such file exists here, so the layout is written a third time, in
    tests/helpers.legacy_image, straight from the published structs; the product's reader and the C oracle's
    must both read it, and the hand-computed answers of tests/golden/hand_cases.json must come out."""
    doc = _hand_doc()
    kw = {"v16_plain": dict(version=(1, 6)),
          "v16_binf_attrs_metrics": dict(version=(1, 6), binf=True, attributes=[("best_iteration", "99"), ("a", "")],
                                         metrics=["rmse", "mae"]),
          "v081_bin": dict(version=(0, 0), objective="reg:linear")}[variant]
    img = helpers.legacy_image(doc, **kw)
    cases, rows = helpers.load_hand_cases()
    got = helpers.oracle_predict(np.frombuffer(img, dtype=np.uint8), rows, cases["missing"])
    assert np.array_equal(got, np.float32(cases["margin"]))
    out, trees = _trees_of(img)
    want = doc["learner"]["gradient_booster"]["model"]["trees"]
    for a, b in zip(trees, want):
        for key in ("left_children", "right_children", "split_indices", "default_left", "parents"):
            assert [int(x) for x in a[key]] == [int(x) for x in b[key]], key
        assert np.array_equal(np.float32(a["split_conditions"]), np.float32(b["split_conditions"]))
    assert out["learner"]["objective"]["name"] == kw.get("objective", "reg:squarederror")
    if "attributes" in kw:
        assert out["learner"]["attributes"] == dict(kw["attributes"])
    info = capi.Booster(model_buffer=img).info()
    assert info["num_trees"] == 5 and info["num_nodes"] == 11 and info["num_feature"] == 3


@pytest.mark.parametrize("damage", ["garbage_trailer", "truncated_trailer", "num_deleted_lie", "flags_without_trailer"])
def test_what_prediction_does_not_need_cannot_fail_the_load(damage, capfd):
    """Attributes, metric names and the num_deleted counter are not needed to predict (the oracle's reader
    never looks at them): a production file whose trailer this library mis-reads must still load."""
    doc = _hand_doc()
    if damage == "garbage_trailer":
        img = helpers.legacy_image(doc, attributes=[("k", "v")], trailer=b"\xff" * 37)
    elif damage == "truncated_trailer":
        full = helpers.legacy_image(doc, attributes=[("k", "v")], metrics=["rmse"])
        img = full[:-5]
    elif damage == "num_deleted_lie":
        img = helpers.legacy_image(doc, num_deleted_lie=3)
    else:
        img = helpers.legacy_image(doc, attributes=[("k", "v")], metrics=["rmse"], trailer=b"")
    b = capi.Booster(model_buffer=img)
    assert b.info()["num_nodes"] == 11
    assert "[libohxgb] warning: model file:" in capfd.readouterr().err
    cases, rows = helpers.load_hand_cases()
    got = helpers.oracle_predict(np.frombuffer(img, dtype=np.uint8), rows, cases["missing"])
    assert np.array_equal(got, np.float32(cases["margin"]))


def test_poisson_files_carry_max_delta_step_between_attributes_and_metrics():
    doc = _hand_doc()
    img = helpers.legacy_image(doc, binf=True, objective="count:poisson", attributes=[("k", "v")],
                               max_delta_step="0.7", metrics=["poisson-nloglik"], base_score=0.5)
    again = synth.convert_model(np.frombuffer(img, dtype=np.uint8), "binary").tobytes()
    assert again == img                      # every section found where it is, and written back in the same order


def test_margins_start_from_prob_to_margin_of_the_objective():
    """xgboost 1.6.0 keeps the user's base_score in the file and starts margins from
    obj->ProbToMargin(base_score); a binary file from xgboost < 1.0 already holds the margin.  Identity for
    the OH model; pinned here for the others so that option_mask = 1 is not silently off (both oracles;
    the GPU library against them in tests/test_gpu_parity.py)."""
    doc = _hand_doc()
    cases, rows = helpers.load_hand_cases()
    for t in doc["learner"]["gradient_booster"]["model"]["trees"][:2]:
        t["split_conditions"] = [0.0]        # drop the +-1e8 stumps: keep the start value visible
    lmp = doc["learner"]["learner_model_param"]
    ident = O.predict(O.load_model(json.dumps(doc).encode()), rows, missing=cases["missing"])
    for objective, base, start in (("binary:logistic", 0.25, -np.log(np.float32(3.0))),
                                   ("count:poisson", 0.5, np.log(np.float32(0.5))),
                                   ("reg:gamma", 2.0, np.log(np.float32(2.0)))):
        doc["learner"]["objective"]["name"] = objective
        lmp["base_score"] = repr(base)
        m = O.load_model(json.dumps(doc).encode())
        assert abs(float(m.base_score) - float(start)) < 1e-6
        got = O.predict(m, rows, missing=cases["missing"])
        shifted = (ident - np.float32(float(_hand_doc()["learner"]["learner_model_param"]["base_score"]))) + m.base_score
        assert np.allclose(got, shifted, atol=1e-5)
        # the C oracle, through a 1.6 binary image and through a pre-1.0 one (margin stored as is)
        new = helpers.oracle_predict(np.frombuffer(helpers.legacy_image(doc, version=(1, 6)), dtype=np.uint8), rows,
                                     cases["missing"], option_mask=1)
        assert np.array_equal(helpers.bits(new), helpers.bits(got))
        old = helpers.oracle_predict(np.frombuffer(helpers.legacy_image(doc, version=(0, 0)), dtype=np.uint8), rows,
                                     cases["missing"], option_mask=1)
        m.base_score = np.float32(base)
        assert np.array_equal(helpers.bits(old), helpers.bits(O.predict(m, rows, missing=cases["missing"])))
