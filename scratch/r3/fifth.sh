#!/bin/bash
O=gpurun_out
HIDVAE_GRAPH_QUEUES=0 python -m pytest tests/test_train_gpu.py -m gpu -x -q > $O/r3e_train_q0.log 2>&1; echo "train tests, runtime default queues rc=$? $(tail -1 $O/r3e_train_q0.log)" | tee -a $O/r3e_summary.log
python -m pytest tests/test_train_gpu.py -m gpu -x -q > $O/r3e_train_q12.log 2>&1; echo "train tests, 12 queues rc=$? $(tail -1 $O/r3e_train_q12.log | cut -c1-100)" | tee -a $O/r3e_summary.log
HIDVAE_GRAPH_QUEUES=4 python -m pytest tests/test_train_gpu.py -m gpu -x -q > $O/r3e_train_q4.log 2>&1; echo "train tests, 4 queues rc=$? $(tail -1 $O/r3e_train_q4.log | cut -c1-100)" | tee -a $O/r3e_summary.log
python tools/step_phases.py --tagged 1 > $O/r3e_phases_b1024.log 2>&1
python tools/step_phases.py --tagged 1 --ahead 1 > $O/r3e_phases_b1024_idle.log 2>&1
python tools/step_phases.py --tagged 0 > $O/r3e_phases_untagged.log 2>&1
grep -v amdgpu.ids $O/r3e_phases_b1024.log
