set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
for g in 1 2; do timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --param ohx_tile_group=$g > $O/verify_g$g.log 2>&1; echo "group $g rc=$? $(grep -o '"verified": [a-z]*' $O/verify_g$g.log) $(grep -o '"ms_per_step": [0-9.]*' $O/verify_g$g.log)"; done
tools/ab.sh $O/ab.txt 3 "line4|-|" "ij2x2|-|--param ohx_tile_group=1" "ik2x2|-|--param ohx_tile_group=2" > /dev/null
sort $O/ab.txt
