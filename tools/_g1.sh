set -o pipefail
mkdir -p gpurun_out/r03b
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_run1.py tests/test_gridcomp.py -x -q -m gpu > gpurun_out/r03b/tests.log 2>&1; echo "tests rc=$?" 
tail -3 gpurun_out/r03b/tests.log
tools/sweep.sh gpurun_out/r03b/sweep.txt --steps 10 --warmup 3 --cpu-seconds 0 --no-verify -- "--path fields" "--path run1" "" "--path fields" "--path run1"
