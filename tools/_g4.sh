set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 300 python3 tools/latency_rows.py > $O/latency.json 2> $O/latency.err; echo "latency rc=$?"
python3 -c "
import json; l=json.load(open('$O/latency.json'))
for n,e in l['rows'].items(): print(n, {k:(round(v['p50_us']),round(v['p95_us'])) for k,v in e.items()})"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_random_forests.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value']/1e9, d['ms_per_step']); print(d['pcie_inclusive']); print(d['cpu_baseline']['ticks_s'])"
