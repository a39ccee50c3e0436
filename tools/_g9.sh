set -o pipefail
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
tools/sweep.sh $O/sweep.txt --steps 10 --warmup 3 --cpu-seconds 0 --no-verify -- "--path fields" "--path run1" ""
