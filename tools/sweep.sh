#!/bin/bash
# usage: tools/sweep.sh <outfile> <common bench args> -- "<variant args>" "<variant args>" ...
# Runs bench.py once per variant and prints: value (Mcells/s), kernel ms, variant.
out=$1; shift
common=()
while [ "$1" != "--" ]; do common+=("$1"); shift; done
shift
: > "$out"
for v in "$@"; do
  line=$(timeout -k 10 300 python bench.py "${common[@]}" $v 2>/dev/null | tail -1)
  python - "$v" "$line" >> "$out" <<'PY'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    r = d.get('roofline') or {}
    if 'kernel_ms' in r:
        print(f"{d['value']/1e6:9.1f} Mcells/s  kernel {r['kernel_ms']:8.2f} ms  frac {r['frac']:.4f}  | {v}")
    else:
        print(f"{d['value']/1e6:9.1f} Mcells/s  step   {d['ms_per_step']:8.2f} ms               | {v}")
except Exception as e:
    print(f"FAILED | {v} | {line[:200]}")
PY
done
cat "$out"
