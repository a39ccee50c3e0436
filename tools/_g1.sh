#!/bin/bash
mkdir -p gpurun_out/fill
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_run1.py tests/test_gridcomp.py -m gpu -x -q -k "fields or run1 or gridcomp or defer" > gpurun_out/fill/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/fill/tests.log
grep -q "failed\|error" gpurun_out/fill/tests.log && exit 1
bash tools/ab.sh gpurun_out/fill/ab.txt 2 "base_fields|tools/bin/variants/base|--path fields" "new_fields|-|--path fields" "base_run1|tools/bin/variants/base|--path run1" "new_run1|-|--path run1"
