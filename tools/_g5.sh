set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_random_forests.py tests/test_sklearn_crosscheck.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for ppm in 100 1000 10000; do timeout -k 10 300 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --missing-ppm $ppm > $O/verify_$ppm.log 2>&1; echo "ppm $ppm rc=$? $(grep -o '"verified": [a-z]*' $O/verify_$ppm.log) $(grep -o '"ms_per_step": [0-9.]*' $O/verify_$ppm.log)"; done
tools/ab.sh $O/ab.txt 3 \
  "clean|-|" \
  "m100|-|--missing-ppm 100" "m100_off|-|--missing-ppm 100 --param ohx_defer_missing=off" \
  "m300|-|--missing-ppm 300" \
  "m1000|-|--missing-ppm 1000" "m1000_off|-|--missing-ppm 1000 --param ohx_defer_missing=off" \
  "m10000|-|--missing-ppm 10000" "m10000_off|-|--missing-ppm 10000 --param ohx_defer_missing=off" > /dev/null
sort $O/ab.txt
