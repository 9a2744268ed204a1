#!/bin/bash
O=gpurun_out
for q in 4 0 12 6; do
HIDVAE_GRAPH_QUEUES=$q python -m pytest tests -m gpu -x -q > $O/r3f_tests_q$q.log 2>&1; echo "full suite, queues=$q rc=$? $(tail -1 $O/r3f_tests_q$q.log | cut -c1-120)" | tee -a $O/r3f_summary.log
done
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50"
for q in 4 5 6 8 12; do HIDVAE_GRAPH_QUEUES=$q python bench.py $B > $O/r3f_q.json 2>/dev/null; echo "tagged queues=$q: $(cat $O/r3f_q.json)" | tee -a $O/r3f_summary.log; done
