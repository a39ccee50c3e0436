set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
V=tools/bin/variants
# 1. whole-batch cross-check (wide kernel) + oracle on the first 2^18 rows, once per build / mode
for spec in "product_dyn|-|--param ohx_dynamic_tiles=1" "sstep6|$V/sstep6|" "sstep6_miss|$V/sstep6|--missing-ppm 100" "dyn_miss|-|--param ohx_dynamic_tiles=1 --missing-ppm 100"; do
  IFS='|' read -r label lib args <<< "$spec"
  root=/tmp/chk_$label; mkdir -p $root && cp -r bench.py BASELINE.json quickchem_amd oracle profiles $root/
  [ "$lib" != "-" ] && cp $lib/libohxgb.so $root/quickchem_amd/lib/libohxgb.so
  (cd $root && timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 4 $args 2>&1 | tail -3 | cut -c1-400) > $O/verify_$label.log 2>&1
  echo "$label: $(grep -o '"verified": [a-z]*' $O/verify_$label.log | head -1) $(grep -o 'bench:.*' $O/verify_$label.log | head -1)"
done
# 2. same-device A/B, three interleaved rounds
tools/ab.sh $O/ab.txt 3 \
  "r03base|$V/r03base|" \
  "product|-|" \
  "dyn|-|--param ohx_dynamic_tiles=1" \
  "dyn_lpr3|-|--param ohx_dynamic_tiles=1 --param ohx_launches_per_residency=3" \
  "dyn_lpr4|-|--param ohx_dynamic_tiles=1 --param ohx_launches_per_residency=4" \
  "lpr3|-|--param ohx_launches_per_residency=3" \
  "sstep4|$V/sstep4|" \
  "sstep6|$V/sstep6|" \
  "sstep8|$V/sstep8|" \
  "sstep6_off|$V/sstep6|--param ohx_scalar_step=0" \
  "miss_static|-|--missing-ppm 100" \
  "miss_dyn|-|--param ohx_dynamic_tiles=1 --missing-ppm 100" \
  "miss_dyn_lpr4|-|--param ohx_dynamic_tiles=1 --param ohx_launches_per_residency=4 --missing-ppm 100" > /dev/null
sort $O/ab.txt
