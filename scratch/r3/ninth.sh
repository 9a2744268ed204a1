#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r3i_tests.log 2>&1; echo "full suite rc=$? $(tail -1 $O/r3i_tests.log | cut -c1-150)" | tee -a $O/r3i_summary.log
grep -E "^FAILED|^ERROR" $O/r3i_tests.log | head -30 | tee -a $O/r3i_summary.log
