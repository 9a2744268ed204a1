#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r3c_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r3c_tests.log)" | tee -a $O/r3c_summary.log
for nw in 16 12 8; do HIDVAE_RQ_IDS_NW=$nw HIDVAE_RQ_PF32_NW=$nw python scratch/r3/ids_bench.py 2>&1 | grep items= | tee -a $O/r3c_summary.log; done
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50"
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py $B > $O/r3c_q.json 2>/dev/null; echo "$label: $(cat $O/r3c_q.json)" | tee -a $O/r3c_summary.log
}
run "tagged default(q12,fork-first)" X=1
run "tagged q16" DEBUG_HIP_FORCE_GRAPH_QUEUES=16
run "tagged q24" DEBUG_HIP_FORCE_GRAPH_QUEUES=24
run "tagged q32" DEBUG_HIP_FORCE_GRAPH_QUEUES=32
run "tagged q12 batch1" DEBUG_HIP_GRAPH_BATCH_SIZE=1
run "tagged q12 batch16" DEBUG_HIP_GRAPH_BATCH_SIZE=16
run "tagged q12 batch256" DEBUG_HIP_GRAPH_BATCH_SIZE=256
run "tagged q12 pktcap0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "tagged q12 pktcap1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "tagged q12 side_stream" HIDVAE_SIDE_STREAM=1
run "tagged q12 grouped" HIDVAE_TAG_GROUPED=1
run "tagged q12 tagstreams1" HIDVAE_TAG_STREAMS=1
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50 --batch 2048"
run "B2048 tagged default" X=1
run "B2048 tagged side_stream" HIDVAE_SIDE_STREAM=1
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3"
run "untagged default" X=1
run "untagged side_stream" HIDVAE_SIDE_STREAM=1
