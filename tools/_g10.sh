set -o pipefail
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 300 python3 tools/latency_rows.py > $O/latency.json 2> $O/latency.err; echo "latency rc=$?"
python3 -c "
import json; l=json.load(open('$O/latency.json'))
for n,e in l['rows'].items(): print(n, {k:(round(v['p50_us']),round(v['p95_us'])) for k,v in e.items()})"
tools/sweep.sh $O/sweep.txt --steps 10 --warmup 3 --cpu-seconds 0 --no-verify -- "" "--grid C48" "--grid C48 --param ohx_tree_split=off" "--grid C90" "--grid C90 --param ohx_tree_split=off"
