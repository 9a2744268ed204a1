#!/bin/bash
# step_ab.sh "ENV=VAL ..." ... : bench.py step time for a list of environment settings x configurations
B="--also-other 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --steps 100"
for e in "$@"; do
  for cfg in "--tagged 1" "--tagged 1 --batch 2048" "--tagged 0" "--tagged 0 --batch 8192"; do
    out=$(env $e timeout -k 10 200 python bench.py $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4))")
    echo "$e | $cfg : $out ms"
  done
done
