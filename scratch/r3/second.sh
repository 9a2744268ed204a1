#!/bin/bash
O=gpurun_out
B="--also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 --tagged 1 --steps 50"
for q in default 2 3 6 8 12; do
  if [ $q = default ]; then python bench.py $B > $O/r3b_q.json 2>/dev/null; else DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python bench.py $B > $O/r3b_q.json 2>/dev/null; fi
  echo "queues=$q: $(cat $O/r3b_q.json)" | tee -a $O/r3b_summary.log
done
DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python tools/step_gantt.py --tagged 1 --each 1 > $O/r3b_gantt_q8.log 2>&1
head -12 $O/r3b_gantt_q8.log | tee -a $O/r3b_summary.log
for q in default 8; do
  if [ $q = default ]; then python bench.py $B --batch 2048 > $O/r3b_q.json 2>/dev/null; else DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python bench.py $B --batch 2048 > $O/r3b_q.json 2>/dev/null; fi
  echo "B=2048 queues=$q: $(cat $O/r3b_q.json)" | tee -a $O/r3b_summary.log
done
for q in default 8; do
  if [ $q = default ]; then python bench.py --also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 > $O/r3b_q.json 2>/dev/null; else DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python bench.py --also-tagged 0 --also-large 0 --kernels 0 --cpu-seconds 0 --windows 3 > $O/r3b_q.json 2>/dev/null; fi
  echo "untagged queues=$q: $(cat $O/r3b_q.json)" | tee -a $O/r3b_summary.log
done
python scratch/r3/dump_graph.py > $O/r3b_dump.log 2>&1; tail -2 $O/r3b_dump.log
for d in 0 1 2 3 8 11; do HIDVAE_RQ_DBG=$d python scratch/r3/ids_bench.py 2>&1 | grep "items=1048576" | sed "s/^/dbg=$d /" | tee -a $O/r3b_summary.log; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/r3b_ids_sq -- python3 $R/scratch/r3/ids_prof.py > $R/$O/r3b_ids_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/r3b_ids_sq2 -- python3 $R/scratch/r3/ids_prof.py > $R/$O/r3b_ids_sq2.log 2>&1
cd $R
python3 tools/pmc_table.py $O/r3b_ids_sq > $O/r3b_ids_sq.csv 2>&1; python3 tools/pmc_table.py $O/r3b_ids_sq2 > $O/r3b_ids_sq2.csv 2>&1
rm -rf $O/r3b_ids_sq $O/r3b_ids_sq2
cat $O/r3b_ids_sq.csv $O/r3b_ids_sq2.csv | grep -i "pf32\|name" | cut -c1-600
python -m pytest tests/test_dp_gpu.py tests/test_properties_gpu.py -m gpu -x -q 2>&1 | tail -3
