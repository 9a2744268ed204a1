#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r3j_tests.log 2>&1; echo "full suite rc=$? $(tail -1 $O/r3j_tests.log | cut -c1-150)" | tee -a $O/r3j_summary.log
grep -E "^FAILED|^ERROR" $O/r3j_tests.log | head -30 | tee -a $O/r3j_summary.log
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50"
python bench.py $B > $O/r3j_q.json 2>$O/r3j_q.err; echo "tagged B1024: $(cat $O/r3j_q.json)" | tee -a $O/r3j_summary.log
python bench.py $B --batch 2048 > $O/r3j_q.json 2>/dev/null; echo "tagged B2048: $(cat $O/r3j_q.json)" | tee -a $O/r3j_summary.log
python tools/step_timeline.py --tagged 1 --top 12 > $O/r3j_timeline_tagged.log 2>&1
tail -3 $O/r3j_q.err
