#!/bin/bash
# round 3, first GPU call: tests, DP fixed cost with the collectives inside the graph, ids-only VQ variants, tagged-step structure
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r3a_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/r3a_summary.log
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3"
python bench.py $B > $O/r3a_plain.json 2> $O/r3a_plain.err; echo "plain: $(cat $O/r3a_plain.json)" | tee -a $O/r3a_summary.log
python bench.py $B --dist 1 > $O/r3a_dist_ingraph.json 2> $O/r3a_dist_ingraph.err; echo "dist in-graph: $(cat $O/r3a_dist_ingraph.json)" | tee -a $O/r3a_summary.log
HIDVAE_DP_GRAPH_COLLECTIVES=0 python bench.py $B --dist 1 > $O/r3a_dist_old.json 2> $O/r3a_dist_old.err; echo "dist old: $(cat $O/r3a_dist_old.json)" | tee -a $O/r3a_summary.log
HIDVAE_DP_OVERLAP=1 python bench.py $B --dist 1 > $O/r3a_dist_ingraph_ov.json 2> $O/r3a_dist_ingraph_ov.err; echo "dist in-graph overlapped: $(cat $O/r3a_dist_ingraph_ov.json)" | tee -a $O/r3a_summary.log
python bench.py $B --tagged 1 --steps 50 > $O/r3a_tag_plain.json 2> $O/r3a_tag_plain.err; echo "tagged plain: $(cat $O/r3a_tag_plain.json)" | tee -a $O/r3a_summary.log
python bench.py $B --tagged 1 --steps 50 --dist 1 > $O/r3a_tag_dist.json 2> $O/r3a_tag_dist.err; echo "tagged dist in-graph (overlap by size): $(cat $O/r3a_tag_dist.json)" | tee -a $O/r3a_summary.log
HIDVAE_DP_OVERLAP=0 python bench.py $B --tagged 1 --steps 50 --dist 1 > $O/r3a_tag_dist_noov.json 2> $O/r3a_tag_dist_noov.err; echo "tagged dist in-graph no overlap: $(cat $O/r3a_tag_dist_noov.json)" | tee -a $O/r3a_summary.log
for nw in 16 12 8 0; do HIDVAE_RQ_IDS_NW=$nw python scratch/r3/ids_bench.py >> $O/r3a_ids.log 2>&1; done
cat $O/r3a_ids.log | tee -a $O/r3a_summary.log
python tools/step_gantt.py --tagged 1 --each 1 > $O/r3a_gantt_tagged.log 2>&1
python tools/step_gantt.py --tagged 0 --each 1 > $O/r3a_gantt_untagged.log 2>&1
python tools/step_timeline.py --tagged 1 --each 1 --top 60 > $O/r3a_timeline_tagged.log 2>&1
head -12 $O/r3a_gantt_tagged.log
tail -5 $O/r3a_tests.log
