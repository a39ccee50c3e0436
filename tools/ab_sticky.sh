#!/bin/bash
# one GPU session for the sticky-leaf layout: parity tests on a private copy of the tree with the variant's libraries,
# a verified bench run, then same-device A/B against the product
R=$PWD
O=$R/gpurun_out/sticky
mkdir -p $O
rm -rf /tmp/st && mkdir -p /tmp/st && cp -r bench.py BASELINE.json quickchem_amd oracle tests include __graft_entry__.py profiles tools /tmp/st/ || exit 1
cp tools/bin/variants/sticky/*.so /tmp/st/quickchem_amd/lib/ || exit 1
(cd /tmp/st && timeout -k 10 600 python -m pytest tests/test_random_forests.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1) || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
(cd /tmp/st && timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-pcie --no-rank-ticks --cpu-seconds 3 > $O/bench_verified.json 2> $O/bench_verified.err) || { tail -20 $O/bench_verified.err; exit 1; }
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_verified.json") if l.startswith('{"metric"')][0])
print("verified bench:", d["ms_per_step"], d.get("verified"), d.get("verified_against"), d["config"]["booster"])
PY
timeout -k 10 900 tools/ab.sh $O/ab.log 3 "product|-|" "sticky|tools/bin/variants/sticky|" "product_d6|-|--depth 6" "sticky_d6|tools/bin/variants/sticky|--depth 6" "product_fields|-|--path fields" "sticky_fields|tools/bin/variants/sticky|--path fields"
