#!/bin/bash
# step_ab1.sh "ENV=VAL ..." ... : tagged B=1024 step time for a list of environment settings
B="--also-other 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --steps 100 --tagged 1"
for e in "$@"; do
    out=$(env $e timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4))")
    echo "$e : $out ms"
done
