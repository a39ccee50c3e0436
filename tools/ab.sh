#!/bin/bash
# Same-device A/B of library builds: tools/ab.sh <out.log> <rounds> "<label>|<libdir or ->|<bench args>" ...
# Each variant runs from a private copy of the tree with its libohxgb.so swapped in.
out=$1; rounds=$2; shift 2
: > "$out"
for r in $(seq 1 "$rounds"); do
  for spec in "$@"; do
    IFS='|' read -r label lib args <<< "$spec"
    root=/tmp/ab_$label
    if [ ! -d "$root" ]; then
      mkdir -p "$root" && cp -r bench.py BASELINE.json quickchem_amd oracle profiles "$root"/ || exit 1
      [ "$lib" != "-" ] && cp "$lib"/libohxgb.so "$root"/quickchem_amd/lib/libohxgb.so
    fi
    line=$(cd "$root" && python bench.py --steps 10 --warmup 3 --no-verify $args 2>/dev/null | grep '^{"metric"')
    echo "$label $(python -c "import sys,json; d=json.loads(sys.argv[1]); print('%.3f ms  %.1f Mcells/s' % (d['ms_per_step'], d['value']/1e6))" "$line")" >> "$out"
  done
done
cat "$out"
