#!/bin/bash
# step_ab_cfg.sh "ENV=VAL ..." ... : bench.py step time for a list of environment settings x the tagged configurations (1024, config 3)
B="--also-other 0 --also-large 0 --also-configs 0 --kernels 0 --cpu-seconds 0 --windows 3 --steps 100 --tagged 1"
for e in "$@"; do
  for cfg in "--batch 1024" "--batch 2048 --hparams kuairand"; do
    out=$(env $e timeout -k 10 200 python bench.py $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4))")
    echo "$e | $cfg : $out ms"
  done
done
