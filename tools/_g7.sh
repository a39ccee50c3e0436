set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
V=tools/bin/variants
for v in lsort4; do
  root=/tmp/chk_$v; mkdir -p $root && cp -r bench.py BASELINE.json quickchem_amd oracle profiles $root/ && cp $V/$v/libohxgb.so $root/quickchem_amd/lib/libohxgb.so
  for extra in "" "--missing-ppm 100"; do (cd $root && timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 4 --no-pcie $extra > $OLDPWD/$O/verify_$v.log 2>&1; echo "$v [$extra] rc=$? $(grep -o '"verified": [a-z]*' $OLDPWD/$O/verify_$v.log) $(grep -o '"ms_per_step": [0-9.]*' $OLDPWD/$O/verify_$v.log) $(grep -o 'bench:.*' $OLDPWD/$O/verify_$v.log | head -1)"); done
done
tools/ab.sh $O/ab.txt 3 "product|-|" "lsort2|$V/lsort2|" "lsort4|$V/lsort4|" > /dev/null
sort $O/ab.txt
