#!/usr/bin/env python3
"""Writes tests/golden/oh_specs.json: what the reference's OH_GridComp registers in SetServices - every import,
export and internal field with its metadata - read from the reference's own files:

  * OH_GridComp/OH_StateSpecs.rc (the table MAPL's code generator turns into OH_Import___.h / OH_Export___.h /
    OH_Internal___.h, included at OH_GridCompMod.F90:642,644,786), through tools/acg_lite.parse_specs;
  * OH_GridComp/OH_GridCompMod.F90: the data-driven instance's two specs (:611-634) and the conditional imports written
    with the ADD_IMPORT_* macros (:653-783): the macro definitions are read for what each one sets, every use for
    (short name, long name, units) and for the block it stands in (always / IMPORT_INST / IMPORT_24 /
    IMPORT_PRECOMPUTED).

The output is DATA (names, units, flags), not source text.  tests/test_gridcomp.py compares it with what the product's
OH_GridCompMod::SetServices registers in the mock, for every OH_data_source.  Run here (needs /root/reference):
    python tests/golden/make_oh_specs.py
"""
from __future__ import annotations

import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import acg_lite  # noqa: E402

REF = os.environ.get("OHX_REFERENCE", "/root/reference")
F90 = os.path.join(REF, "OH_GridComp", "OH_GridCompMod.F90")
RC = os.path.join(REF, "OH_GridComp", "OH_StateSpecs.rc")

SHORTHAND = {}      # __HO__ -> {"dims": "MAPL_DimsHorzOnly"} ...  (read from the file)
KEYWORDS = {"DIMS": "dims", "VLOCATION": "vlocation", "RESTART": "restart", "REFRESH_INTERVAL": "refresh_interval",
            "AVERAGING_INTERVAL": "averaging_interval", "UNGRIDDED_DIMS": "ungridded_dims", "SHORT_NAME": "short_name",
            "LONG_NAME": "long_name", "UNITS": "units", "ADD2EXPORT": "add2export"}


def split_args(text):
    """top-level comma split (commas inside brackets, parentheses and quotes stay)"""
    out, depth, quote, cur = [], 0, None, ""
    for ch in text:
        if quote:
            cur += ch
            if ch == quote:
                quote = None
            continue
        if ch in "'\"":
            quote = ch
        elif ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def keyword_args(args):
    """['SHORT_NAME=sss', '__HV__', 'restart = MAPL_RestartSkip', ...] -> dict by our key names"""
    spec = {}
    for a in args:
        if a in SHORTHAND:
            spec.update(SHORTHAND[a])
            continue
        if a in ("__RC__", "GC", "gc"):
            continue
        m = re.match(r"(\w+)\s*=\s*(.+)$", a, flags=re.S)
        if not m:
            raise SystemExit(f"cannot read spec argument {a!r}")
        key = KEYWORDS.get(m.group(1).upper())
        if key is None:
            raise SystemExit(f"unknown spec keyword in {a!r}")
        spec[key] = m.group(2).strip()
    return spec


def tidy(spec):
    out = {}
    for k, v in spec.items():
        if isinstance(v, str):
            v = v.strip()
            if len(v) >= 2 and v[0] in "'\"" and v[-1] == v[0]:
                v = v[1:-1]
        if k in ("refresh_interval", "averaging_interval"):
            v = int(eval(v, {"__builtins__": {}}))            # 60*60*24
        if k == "add2export":
            v = str(v).lower() in (".true.", "t", "true")
        out[k] = v
    return out


def main():
    lines = open(F90).read().split("\n")

    # ---- the shorthand macros (#define __HO__ DIMS=MAPL_DimsHorzOnly ...) and the ADD_IMPORT_* macros
    macros = {}
    for ln in lines:
        m = re.match(r"#define\s+(__\w+__)\s+(.+)$", ln)
        if m and m.group(1) not in ("__RC__",):
            SHORTHAND[m.group(1)] = {}
            for part in split_args(m.group(2)):
                SHORTHAND[m.group(1)].update(keyword_args([part]))
    for ln in lines:
        m = re.match(r"#define\s+(ADD_IMPORT_\w+)\((\w+),(\w+),(\w+)\)\s+call\s+MAPL_AddImportSpec\((.*)\)\s*$", ln)
        if m:
            name, a1, a2, a3, body = m.groups()
            spec = keyword_args(split_args(body))
            assert (spec.pop("short_name"), spec.pop("long_name"), spec.pop("units")) == (a1, a2, a3), ln
            macros[name] = spec
    assert len(macros) >= 9, sorted(macros)

    # ---- every use of one, with the block it stands in
    blocks = {"always": [], "IMPORT_INST": [], "IMPORT_24": [], "IMPORT_PRECOMPUTED": []}
    conditions = {}
    block = "always"
    for no, ln in enumerate(lines, 1):
        code = ln.split("!")[0]
        m = re.match(r"\s*(IMPORT_\w+)\s*:\s*IF\s*\((.*)$", code, flags=re.I)
        if m:
            block = m.group(1)
            cond, j = m.group(2), no
            while cond.rstrip().endswith("&"):                # continued condition
                cond = cond.rstrip()[:-1] + " " + lines[j].split("!")[0].strip()
                j += 1
            conditions[block] = re.sub(r"\s+", " ", re.sub(r"\)\s*THEN\s*$", "", cond.strip(), flags=re.I)).strip()
            continue
        if re.match(r"\s*END\s+IF\s+IMPORT_\w+", code, flags=re.I):
            block = "always"
            continue
        m = re.match(r"\s*(ADD_IMPORT_\w+)\s*\((.*)\)\s*$", code)
        if m and not ln.lstrip().startswith("#"):
            short, long_name, units = split_args(m.group(2))
            spec = dict(macros[m.group(1)], short_name=short, long_name=long_name, units=units)
            blocks[block].append(dict(tidy(spec), line=no))

    # ---- the data-driven instance (:611-634): two explicit calls, one of them in a loop over the bins
    data = []
    text = "\n".join(lines)
    start = text.index("if (data_driven) then", text.index("IMPORT STATE"))
    end = text.index("end if ! (data_driven)")
    chunk = re.sub(r"&\s*\n(\s*!.*\n)*\s*", " ", text[start:end])          # join continuation lines (drop comment lines between)
    for m in re.finditer(r"call\s+MAPL_Add(Internal|Import)Spec\s*\((.*?)__RC__\)", chunk, flags=re.S | re.I):
        args = [a for a in split_args(m.group(2)) if a]
        spec = tidy(keyword_args(args))
        spec["state"] = m.group(1).upper()
        if "trim(field_name)" in spec["short_name"]:
            # inside `do i = 1, self%nbins` with `write (field_name, '(A, I0.3)') '', i` (:624-625): one import per bin
            assert re.search(r"write\s*\(field_name,\s*'\(A,\s*I0\.3\)'\)\s*'',\s*i", chunk)
            fmt = lambda e: re.sub(r"'\s*//\s*trim\(field_name\)\s*(//\s*')?", "{bin:03d}", e).strip("'")
            spec["short_name"], spec["long_name"] = fmt(spec["short_name"]), fmt(spec["long_name"])
            spec["per"] = "bin = 1 .. nbins"
        data.append(spec)
    assert len(data) == 2, data

    specs = acg_lite.parse_specs(RC)
    out = {
        "made_by": "tests/golden/make_oh_specs.py from OH_GridComp/OH_StateSpecs.rc and OH_GridComp/OH_GridCompMod.F90 "
                   "(GEOS-ESM/QuickChem); data only: names, units and flags of the MAPL specs",
        "state_specs": {k: specs[k] for k in ("IMPORT", "EXPORT", "INTERNAL")},
        "conditional_imports": blocks,
        "conditions": conditions,
        "data_instance": data,
        "mapl_defaults": {"vlocation": "MAPL_VLocationNone", "restart": "MAPL_RestartOptional",
                          "refresh_interval": 0, "averaging_interval": 0, "add2export": False},
    }
    path = os.path.join(ROOT, "tests", "golden", "oh_specs.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    n = sum(len(v) for v in blocks.values())
    print(f"{path}: {len(specs['IMPORT'])} + {len(specs['EXPORT'])} + {len(specs['INTERNAL'])} state-spec rows, "
          f"{n} conditional imports in {len(blocks)} blocks, {len(data)} specs of the data instance")


if __name__ == "__main__":
    main()
