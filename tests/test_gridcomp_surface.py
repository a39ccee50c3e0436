"""The GridComp shell is written against MAPL and ESMF by their own names (north_star: "keeping
the MAPL/ESMF GridComp surface ... so it drops into GEOS unchanged"): quickchem_amd/fortran/oh_gridcomp.F90 does
`use ESMF`, `use MAPL`, includes MAPL_Generic.h, and calls nothing of the mock in quickchem_amd/fortran/mapl_lite/
that is not a real MAPL/ESMF name.  (The parent above it is not in this repository at all: it is the reference's own
QuickChem_GridCompMod.F90, compiled in place against the same mock - oracle/Makefile `ref` - which is the other half
of "the mock speaks MAPL".)  What the shell runs on here (the mock) is tested by tests/test_gridcomp.py; this file pins
the surface.
Reference: OH_GridComp/OH_GridCompMod.F90:1,13-14,516-605,693-797,855-897,1147-1185,1195,1820,1860-1886;
QuickChem_GridCompMod.F90:1,116-192,244-273,324-339,392-419,457-474,531."""
import os
import re

from tests import helpers

FORTRAN = os.path.join(helpers.ROOT, "quickchem_amd", "fortran")
SHELL = [os.path.join(FORTRAN, "oh_gridcomp.F90"), os.path.join(FORTRAN, "oh_standalone_cap.F90")]


def test_the_parent_is_not_in_this_repository():
    """QuickChem_GridCompMod is QuickChem's own file (the re-typed parent was a copy): nothing under the
    package defines that module; the test driver uses it from oracle/_ref."""
    for dirpath, _, names in os.walk(os.path.join(helpers.ROOT, "quickchem_amd")):
        for name in names:
            if name.lower().endswith((".f90", ".f")):
                text = open(os.path.join(dirpath, name), errors="replace").read()
                assert not re.search(r"^\s*module\s+QuickChem_GridCompMod\b", text, flags=re.I | re.M), name

# every MAPL_ / ESMF_ identifier the reference's two grid components use (the lines above), plus MAPL_VLocationNone
# (MAPL's constant for 2-D fields, which the reference leaves to the default)
REAL_API = {
    "ESMF_ALARM", "ESMF_ALARMISRINGING", "ESMF_ALARMRINGEROFF", "ESMF_CLOCK", "ESMF_CLOCKGET", "ESMF_CONFIG",
    "ESMF_CONFIGCREATE", "ESMF_CONFIGDESTROY", "ESMF_CONFIGFINDLABEL", "ESMF_CONFIGGETATTRIBUTE", "ESMF_CONFIGGETLEN",
    "ESMF_CONFIGLOADFILE", "ESMF_GRID", "ESMF_GRIDCOMP", "ESMF_GRIDCOMPGET", "ESMF_GRIDCOMPRUN", "ESMF_MAXPATHLEN",
    "ESMF_MAXSTR", "ESMF_METHOD_INITIALIZE", "ESMF_METHOD_RUN", "ESMF_STATE", "ESMF_SUCCESS", "ESMF_TIME", "ESMF_TIMEGET",
    "ESMF_USERCOMPGETINTERNALSTATE", "ESMF_USERCOMPSETINTERNALSTATE", "ESMF_FIELD",
    "MAPL_ADDCHILD", "MAPL_ADDEXPORTSPEC", "MAPL_ADDIMPORTSPEC", "MAPL_ADDINTERNALSPEC", "MAPL_AM_I_ROOT", "MAPL_AVOGAD",
    "MAPL_DEGREES_TO_RADIANS", "MAPL_DIMSHORZONLY", "MAPL_DIMSHORZVERT", "MAPL_EPSILON", "MAPL_GENERIC",
    "MAPL_GENERICINITIALIZE", "MAPL_GENERICSETSERVICES", "MAPL_GET", "MAPL_GETOBJECTFROMGC", "MAPL_GETPOINTER",
    "MAPL_GETRESOURCE", "MAPL_GRIDCOMPSETENTRYPOINT", "MAPL_GRIDGET", "MAPL_MAXMIN", "MAPL_METACOMP", "MAPL_PACKTIME",
    "MAPL_RADIANS_TO_DEGREES", "MAPL_RESTARTOPTIONAL", "MAPL_RESTARTSKIP", "MAPL_RUNIV", "MAPL_VLOCATIONCENTER",
    "MAPL_VLOCATIONEDGE", "MAPL_VLOCATIONNONE", "MAPL_STRINGTEMPLATE", "MAPL_H2OMW", "MAPL_AIRMW",
}


def code_lines(path):
    for line in open(path):
        if line.lstrip().startswith("#"):
            yield line
        else:
            yield line.split("!")[0]


def test_the_shell_calls_nothing_of_the_mock_by_a_mock_name():
    for path in SHELL:
        text = "\n".join(code_lines(path))
        assert not re.search(r"\b(ml_|esmfl_|mapll_)\w*|%add_spec|%get_|mapl_lite", text, flags=re.I), path
        assert not re.search(r"%p%", text), f"{path} looks inside a mock handle"


def test_every_mapl_and_esmf_name_in_the_shell_is_a_real_one():
    for path in SHELL:
        used = set()
        for line in code_lines(path):
            used |= {m.upper() for m in re.findall(r"\b(?:ESMF|MAPL)_\w+", line, flags=re.I)}
        assert used <= REAL_API, (path, sorted(used - REAL_API))


def test_the_shell_uses_esmf_and_mapl_and_the_generic_header():
    for path in SHELL:
        lines = list(code_lines(path))
        assert lines[0].strip() == '#include "MAPL_Generic.h"', path
        uses = {m.lower() for line in lines for m in re.findall(r"^\s*use\s+(\w+)", line, flags=re.I)}
        assert {"esmf", "mapl"} <= uses, (path, uses)
        assert uses <= {"esmf", "mapl", "oh_xgb_predict", "oh_run1", "oh_gridcompmod"}, (path, uses)
    # and the mock sits in a directory of its own, under the real module names
    mock = os.path.join(FORTRAN, "mapl_lite")
    assert re.search(r"^module ESMF\b", open(os.path.join(mock, "ESMF.F90")).read(), flags=re.M)
    assert re.search(r"^module MAPL\b", open(os.path.join(mock, "MAPL.F90")).read(), flags=re.M)
    header = open(os.path.join(mock, "MAPL_Generic.h")).read()
    for macro in ("__Iam__", "__RC__", "__STAT__", "VERIFY_", "_ASSERT", "RETURN_"):
        assert re.search(rf"^#define {re.escape(macro)}\b", header, flags=re.M), macro
