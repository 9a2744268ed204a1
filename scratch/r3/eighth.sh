#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r3h_tests.log 2>&1; echo "full suite rc=$? $(tail -1 $O/r3h_tests.log | cut -c1-150)" | tee -a $O/r3h_summary.log
grep -E "^FAILED|^ERROR" $O/r3h_tests.log | head -30 | tee -a $O/r3h_summary.log
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50"
python bench.py $B > $O/r3h_q.json 2>$O/r3h_q.err; echo "tagged B1024: $(cat $O/r3h_q.json)" | tee -a $O/r3h_summary.log
python bench.py $B --batch 2048 > $O/r3h_q.json 2>/dev/null; echo "tagged B2048: $(cat $O/r3h_q.json)" | tee -a $O/r3h_summary.log
python tools/step_phases.py --tagged 1 > $O/r3h_phases_b1024.log 2>&1
python tools/step_timeline.py --tagged 1 --top 30 > $O/r3h_timeline_tagged.log 2>&1
grep -v amdgpu.ids $O/r3h_phases_b1024.log | head -40
