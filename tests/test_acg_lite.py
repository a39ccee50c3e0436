"""tools/acg_lite.py - the stand-in for MAPL's code generator that oracle/Makefile uses to compile the reference's own
parent grid component in place - and tests/golden/make_oh_specs.py, which pins the OH spec table with the same parser."""
import json
import os
import subprocess
import sys

import pytest

from tests import helpers

sys.path.insert(0, os.path.join(helpers.ROOT, "tools"))
import acg_lite  # noqa: E402

SPECS = """schema_version: 2.0.0
component: TOY

category: IMPORT
#-------------------------------------------------------------
 NAME   | UNITS | DIMS | VLOC | COND            | LONG NAME
#-------------------------------------------------------------
 A      | K     | xyz  | C    |                 | a centre field
 B      | Pa    | xyz  | E    | self%want_b     | an edge field   # only sometimes
# C     | 1     | xy   | N    |                 | commented out

category: EXPORT
 NAME   | UNITS | DIMS | VLOC | UNGRIDDED | LONG NAME
 D      | 1     | xy   | N    | size(self%w) | a 2-D field with bins

category: INTERNAL
 NAME | UNITS | DIMS | VLOC | RESTART | ADD2EXPORT | FRIENDLYTO | LONG NAME
 E    | kg kg-1 | xyz | C | MAPL_RestartOptional | T | DYNAMICS | an internal field
"""


def test_acg_lite_reads_a_state_spec_table_and_writes_the_five_headers(tmp_path):
    rc = tmp_path / "TOY_StateSpecs.rc"
    rc.write_text(SPECS)
    specs = acg_lite.parse_specs(str(rc))
    assert specs["component"] == "TOY"
    assert [r["short_name"] for r in specs["IMPORT"]] == ["A", "B"]
    assert specs["IMPORT"][1] == {"short_name": "B", "units": "Pa", "dims": "MAPL_DimsHorzVert",
                                  "vlocation": "MAPL_VLocationEdge", "condition": "self%want_b", "long_name": "an edge field"}
    assert specs["EXPORT"][0]["ungridded_dims"] == "size(self%w)" and specs["EXPORT"][0]["dims"] == "MAPL_DimsHorzOnly"
    assert specs["INTERNAL"][0]["add2export"] is True and specs["INTERNAL"][0]["friendlyto"] == "DYNAMICS"
    files = acg_lite.headers(specs)
    assert sorted(files) == ["TOY_DeclarePointer___.h", "TOY_Export___.h", "TOY_GetPointer___.h", "TOY_Import___.h",
                             "TOY_Internal___.h"]
    imp = files["TOY_Import___.h"]
    assert imp.count("call MAPL_AddImportSpec(GC,") == 2 and "if (self%want_b) then" in imp and "VLOCATION=MAPL_VLocationEdge" in imp
    assert "UNGRIDDED_DIMS=[size(self%w)]" in files["TOY_Export___.h"]
    assert "RESTART=MAPL_RestartOptional" in files["TOY_Internal___.h"] and "ADD2EXPORT=.true." in files["TOY_Internal___.h"]
    decl = files["TOY_DeclarePointer___.h"]
    assert "real, pointer, dimension(:,:,:) :: A" in decl and "real, pointer, dimension(:,:,:) :: D" in decl     # 2-D + bins
    get = files["TOY_GetPointer___.h"]
    assert "call MAPL_GetPointer(IMPORT, A, 'A', __RC__)" in get and "call MAPL_GetPointer(INTERNAL, E, 'E', __RC__)" in get
    # the command line the Makefile uses
    out = tmp_path / "h"
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "acg_lite.py"), str(rc), "--outdir", str(out), "--json"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == specs and sorted(os.listdir(out)) == sorted(files)
    # what it refuses
    bad = tmp_path / "bad.rc"
    bad.write_text(SPECS.replace("2.0.0", "1.0.0"))
    with pytest.raises(SystemExit):
        acg_lite.parse_specs(str(bad))
    bad.write_text(SPECS.replace(" A      | K     | xyz  | C    |                 | a centre field", " A | K | xyz"))
    with pytest.raises(SystemExit):
        acg_lite.parse_specs(str(bad))


def test_the_committed_oh_spec_table_is_what_the_reference_says_today(tmp_path):
    """Where the reference is at hand (this container), tests/golden/oh_specs.json is regenerated and must be the
    committed file: the golden cannot drift from the files it was read from."""
    if not os.path.exists("/root/reference/OH_GridComp/OH_StateSpecs.rc"):
        pytest.skip("no /root/reference here")
    committed = json.load(open(os.path.join(helpers.GOLDEN, "oh_specs.json")))
    src = open(os.path.join(helpers.GOLDEN, "make_oh_specs.py")).read()
    patched = tmp_path / "make_oh_specs.py"
    patched.write_text(src.replace('path = os.path.join(ROOT, "tests", "golden", "oh_specs.json")',
                                   f'path = {str(tmp_path / "oh_specs.json")!r}')
                       .replace("ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))",
                                f"ROOT = {helpers.ROOT!r}"))
    r = subprocess.run([sys.executable, str(patched)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.load(open(tmp_path / "oh_specs.json")) == committed
    # and it is a table, not text of the reference: names and flags only
    assert set(committed) == {"made_by", "state_specs", "conditional_imports", "conditions", "data_instance", "mapl_defaults"}
    assert len(committed["state_specs"]["EXPORT"]) == 32 and len(committed["conditional_imports"]["IMPORT_24"]) == 16
