set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; grep '^{"metric"' $O/bench_default.log | tail -1 > $O/bench_default.json; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); c=d['cpu_baseline']; print(d['value']/1e9, d['ms_per_step']); print({k:c[k] for k in ('value','cores','ticks_s','first_tick_s','load_s')}); print(c['one_thread'])"
