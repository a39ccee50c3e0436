set -o pipefail
O=gpurun_out/r03k; mkdir -p $O
V=tools/bin/variants
tools/ab.sh $O/ab.txt 3 "serial|-|--path fields" "group3|$V/fill3|--path fields" "group9|$V/fill9|--path fields" "run1_serial|-|--path run1" "run1_group3|$V/fill3|--path run1" > /dev/null
sort $O/ab.txt
