#!/usr/bin/env python3
"""acg_lite.py -- a small stand-in for MAPL's automatic code generator (MAPL_GridCompSpecs_ACG.py), which GEOS runs
over a component's *_StateSpecs.rc at build time (mapl_acg() in the reference's CMakeLists.txt:10-12 and
OH_GridComp/CMakeLists.txt:9-11) to produce the headers the grid components include:

    <Comp>_Import___.h  <Comp>_Export___.h  <Comp>_Internal___.h     MAPL_Add*Spec calls     (OH_GridCompMod.F90:642,644,786;
    <Comp>_DeclarePointer___.h  <Comp>_GetPointer___.h                                        QuickChem_GridCompMod.F90:182,383,411)

MAPL is not in this image, so when the reference's own, unmodified parent QuickChem_GridCompMod.F90 is compiled in
place (oracle/Makefile, target `ref`) its three headers are produced by this script from the reference's
QuickChem_StateSpecs.rc (which has no entries: the headers hold a comment).  The same parser turns a spec file into
JSON (--json): tests/golden/make_oh_specs.py builds the table the OH shell's SetServices is pinned against from
OH_StateSpecs.rc with it.

Schema 2.0.0 as the two files use it: `category: IMPORT|EXPORT|INTERNAL`, a header row of column labels separated
by `|`, then one row per variable; `#` starts a comment; aliases xyz/xy/z, C/E/N (the files' own legend).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

DIMS = {"xyz": "MAPL_DimsHorzVert", "xy": "MAPL_DimsHorzOnly", "z": "MAPL_DimsVertOnly"}
VLOC = {"C": "MAPL_VLocationCenter", "E": "MAPL_VLocationEdge", "N": "MAPL_VLocationNone"}
COLUMN_KEY = {"NAME": "short_name", "UNITS": "units", "DIMS": "dims", "VLOC": "vlocation", "LONG NAME": "long_name",
              "COND": "condition", "UNGRIDDED": "ungridded_dims", "RESTART": "restart", "ADD2EXPORT": "add2export",
              "FRIENDLYTO": "friendlyto", "NUM_SUBTILES": "num_subtiles"}


def parse_specs(path):
    """-> {"component": str, "IMPORT": [row, ...], "EXPORT": [...], "INTERNAL": [...]}; a row maps the MAPL keyword
    of every non-empty column to its value (aliases resolved)."""
    out = {"component": None, "IMPORT": [], "EXPORT": [], "INTERNAL": []}
    category, columns = None, None
    for raw in open(path):
        line = raw.split("#", 1)[0].rstrip()
        if not line.strip():
            continue
        head = line.strip()
        if head.startswith("schema_version:"):
            if head.split(":", 1)[1].strip() != "2.0.0":
                raise SystemExit(f"{path}: schema {head} is not 2.0.0")
            continue
        if head.startswith("component:"):
            out["component"] = head.split(":", 1)[1].strip()
            continue
        if head.startswith("category:"):
            category = head.split(":", 1)[1].strip().upper()
            if category not in ("IMPORT", "EXPORT", "INTERNAL"):
                raise SystemExit(f"{path}: unknown category {category}")
            columns = None
            continue
        cells = [c.strip() for c in line.split("|")]
        if category is None:
            raise SystemExit(f"{path}: a table row before any category: {line!r}")
        if columns is None:                                   # the header row of the category
            columns = cells
            unknown = [c for c in columns if c not in COLUMN_KEY]
            if unknown:
                raise SystemExit(f"{path}: unknown column label(s) {unknown}")
            continue
        if len(cells) != len(columns):
            raise SystemExit(f"{path}: {len(cells)} cells under {len(columns)} columns: {line!r}")
        row = {}
        for label, cell in zip(columns, cells):
            if cell == "":
                continue
            key = COLUMN_KEY[label]
            if key == "dims":
                cell = DIMS[cell]
            elif key == "vlocation":
                cell = VLOC[cell]
            elif key == "add2export":
                cell = cell.upper() in ("T", ".TRUE.", "TRUE")
            row[key] = cell
        out[category].append(row)
    return out


def _spec_call(category, row):
    which = {"IMPORT": "MAPL_AddImportSpec", "EXPORT": "MAPL_AddExportSpec", "INTERNAL": "MAPL_AddInternalSpec"}[category]
    args = [f"SHORT_NAME='{row['short_name']}'"]
    if "long_name" in row:
        args.append(f"LONG_NAME='{row['long_name']}'")
    if "units" in row:
        args.append(f"UNITS='{row['units']}'")
    if "dims" in row:
        args.append(f"DIMS={row['dims']}")
    if "vlocation" in row:
        args.append(f"VLOCATION={row['vlocation']}")
    if "ungridded_dims" in row:
        args.append(f"UNGRIDDED_DIMS=[{row['ungridded_dims']}]")
    if "restart" in row:
        args.append(f"RESTART={row['restart']}")
    if row.get("add2export"):
        args.append("ADD2EXPORT=.true.")
    if "friendlyto" in row:
        args.append(f"FRIENDLYTO='{row['friendlyto']}'")
    body = f"call {which}(GC, &\n     " + ", &\n     ".join(args) + ", __RC__)\n"
    if "condition" in row:
        return f"if ({row['condition']}) then\n{body}end if\n"
    return body


def _rank(row):
    n = {"MAPL_DimsHorzVert": 3, "MAPL_DimsHorzOnly": 2, "MAPL_DimsVertOnly": 1}[row.get("dims", "MAPL_DimsHorzOnly")]
    return n + (1 if "ungridded_dims" in row else 0)


def headers(specs):
    """-> {file name: text} for the five headers of the component"""
    comp = specs["component"]
    note = f"!  generated by tools/acg_lite.py from {comp}_StateSpecs.rc\n"
    files = {}
    for cat, stem in (("IMPORT", "Import"), ("EXPORT", "Export"), ("INTERNAL", "Internal")):
        files[f"{comp}_{stem}___.h"] = note + "".join(_spec_call(cat, r) for r in specs[cat])
    decl, get = note, note
    for cat in ("IMPORT", "EXPORT", "INTERNAL"):
        for r in specs[cat]:
            name = r["short_name"]
            decl += f"real, pointer, dimension({','.join(':' * _rank(r))}) :: {name}\n"
            line = f"call MAPL_GetPointer({cat}, {name}, '{name}', __RC__)\n"
            get += f"if ({r['condition']}) then\n{line}end if\n" if "condition" in r else line
    files[f"{comp}_DeclarePointer___.h"] = decl
    files[f"{comp}_GetPointer___.h"] = get
    return files


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("specs", help="a *_StateSpecs.rc file")
    ap.add_argument("--outdir", help="write the component's five headers here")
    ap.add_argument("--json", action="store_true", help="print the parsed table as JSON")
    args = ap.parse_args()
    specs = parse_specs(args.specs)
    if args.json:
        json.dump(specs, sys.stdout, indent=1)
        print()
    if args.outdir:
        os.makedirs(args.outdir, exist_ok=True)
        for name, text in headers(specs).items():
            with open(os.path.join(args.outdir, name), "w") as f:
                f.write(text)


if __name__ == "__main__":
    main()
