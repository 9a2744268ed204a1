#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -x -q --deselect tests/test_dp_gpu.py > $O/r3d_tests_nodp.log 2>&1; echo "tests(no dp) rc=$? $(tail -1 $O/r3d_tests_nodp.log)" | tee -a $O/r3d_summary.log
HIDVAE_GRAPH_QUEUES=0 python -m pytest tests/test_dp_gpu.py -m gpu -x -q > $O/r3d_dp_q0.log 2>&1; echo "dp tests, runtime default queues rc=$? $(tail -1 $O/r3d_dp_q0.log)" | tee -a $O/r3d_summary.log
HIDVAE_GRAPH_QUEUES=4 python -m pytest tests/test_dp_gpu.py -m gpu -x -q > $O/r3d_dp_q4.log 2>&1; echo "dp tests, 4 queues rc=$? $(tail -1 $O/r3d_dp_q4.log)" | tee -a $O/r3d_summary.log
HIDVAE_DP_GRAPH_COLLECTIVES=0 python -m pytest tests/test_dp_gpu.py -m gpu -x -q -k "between_graphs or two_ranks" > $O/r3d_dp_q12_old.log 2>&1; echo "dp tests, 12 queues, collectives between graphs rc=$? $(tail -1 $O/r3d_dp_q12_old.log)" | tee -a $O/r3d_summary.log
python -m pytest tests/test_dp_gpu.py -m gpu -x -q -k "in_the_graph and untagged" > $O/r3d_dp_q12_in.log 2>&1; echo "dp tests, 12 queues, in graph, untagged rc=$? $(tail -1 $O/r3d_dp_q12_in.log)" | tee -a $O/r3d_summary.log
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3"
python bench.py $B --dist 1 > $O/r3d_q.json 2>/dev/null; echo "untagged dist1 in-graph q12 rc=$?: $(cat $O/r3d_q.json)" | tee -a $O/r3d_summary.log
HIDVAE_DP_GRAPH_COLLECTIVES=0 python bench.py $B --dist 1 > $O/r3d_q.json 2>/dev/null; echo "untagged dist1 between graphs q12 rc=$?: $(cat $O/r3d_q.json)" | tee -a $O/r3d_summary.log
HIDVAE_DP_GRAPH_COLLECTIVES=0 python bench.py $B --dist 1 --tagged 1 --steps 50 > $O/r3d_q.json 2>/dev/null; echo "tagged dist1 between graphs q12 rc=$?: $(cat $O/r3d_q.json)" | tee -a $O/r3d_summary.log
python bench.py $B --dist 1 --tagged 1 --steps 50 > $O/r3d_q.json 2>/dev/null; echo "tagged dist1 in graph q12 rc=$?: $(cat $O/r3d_q.json)" | tee -a $O/r3d_summary.log
python scratch/r3/ids_bench.py 2>&1 | grep items= | sed "s/^/with fixup: /" | tee -a $O/r3d_summary.log
HIDVAE_RQ_NOFIX=1 python scratch/r3/ids_bench.py 2>&1 | grep items= | sed "s/^/NO fixup launch: /" | tee -a $O/r3d_summary.log
HIDVAE_RQ_PF32_NW=12 python scratch/r3/ids_bench.py 2>&1 | grep items= | sed "s/^/full forms 12 waves: /" | tee -a $O/r3d_summary.log
python tools/step_phases.py --tagged 1 > $O/r3d_phases_b1024.log 2>&1; grep -v amdgpu.ids $O/r3d_phases_b1024.log | head -60
python tools/step_phases.py --tagged 1 --batch 2048 > $O/r3d_phases_b2048.log 2>&1
