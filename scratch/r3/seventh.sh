#!/bin/bash
O=gpurun_out
for q in 3 5; do
HIDVAE_GRAPH_QUEUES=$q python -m pytest tests -m gpu -x -q > $O/r3g_tests_q$q.log 2>&1; echo "full suite, queues=$q rc=$? $(tail -1 $O/r3g_tests_q$q.log | cut -c1-120)" | tee -a $O/r3g_summary.log
done
GPU_MAX_HW_QUEUES=8 HIDVAE_GRAPH_QUEUES=8 python -m pytest tests -m gpu -x -q > $O/r3g_tests_hw8q8.log 2>&1; echo "full suite, GPU_MAX_HW_QUEUES=8 queues=8 rc=$? $(tail -1 $O/r3g_tests_hw8q8.log | cut -c1-120)" | tee -a $O/r3g_summary.log
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50"
for q in 3 3 5; do HIDVAE_GRAPH_QUEUES=$q python bench.py $B > $O/r3g_q.json 2>/dev/null; echo "tagged queues=$q: $(cat $O/r3g_q.json)" | tee -a $O/r3g_summary.log; done
GPU_MAX_HW_QUEUES=8 HIDVAE_GRAPH_QUEUES=8 python bench.py $B > $O/r3g_q.json 2>/dev/null; echo "tagged GPU_MAX_HW_QUEUES=8 queues=8: $(cat $O/r3g_q.json)" | tee -a $O/r3g_summary.log
GPU_MAX_HW_QUEUES=8 HIDVAE_GRAPH_QUEUES=4 python bench.py $B > $O/r3g_q.json 2>/dev/null; echo "tagged GPU_MAX_HW_QUEUES=8 queues=4: $(cat $O/r3g_q.json)" | tee -a $O/r3g_summary.log
HIDVAE_GRAPH_QUEUES=3 python bench.py $B --batch 2048 > $O/r3g_q.json 2>/dev/null; echo "B2048 tagged queues=3: $(cat $O/r3g_q.json)" | tee -a $O/r3g_summary.log
HIDVAE_GRAPH_QUEUES=3 python bench.py --also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 > $O/r3g_q.json 2>/dev/null; echo "untagged queues=3: $(cat $O/r3g_q.json)" | tee -a $O/r3g_summary.log
HIDVAE_GRAPH_QUEUES=3 python bench.py --also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --dist 1 > $O/r3g_q.json 2>/dev/null; echo "untagged dist1 queues=3: $(grep value $O/r3g_q.json)" | tee -a $O/r3g_summary.log
