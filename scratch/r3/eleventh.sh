#!/bin/bash
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r3k_tests.log 2>&1; echo "full suite rc=$? $(tail -1 $O/r3k_tests.log | cut -c1-150)" | tee -a $O/r3k_summary.log
grep -E "^FAILED|^ERROR" $O/r3k_tests.log | head -30 | tee -a $O/r3k_summary.log
( time python bench.py > $O/r3k_bench.json 2> $O/r3k_bench.err ) 2> $O/r3k_bench.time; tail -3 $O/r3k_bench.time | tee -a $O/r3k_summary.log
python - <<'PY' | tee -a gpurun_out/r3k_summary.log
import json
d=json.load(open('gpurun_out/r3k_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus','dtype')}, d['config'])
print('roofline', {k:v for k,v in d['roofline'].items() if k!='kernel'})
print('untagged', {k:d['untagged_core_step'][k] for k in ('value','ms_per_step','launches')})
print('large', {k:d['large_batch_step'][k] for k in ('value','ms_per_step')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'other cpu', d['untagged_core_step']['cpu_baseline']['value'])
for k in d['kernels']: print(round(k['us'],1), round(k['frac'],3), k['kernel'][:90])
print('in_step', d['in_step']['launches'], d['in_step']['gemm_class'])
PY
tail -5 $O/r3k_bench.err
