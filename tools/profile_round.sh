#!/bin/bash
# rocprofv3 evidence passes of a round (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/).
#   bash tools/profile_round.sh r02
# Counters go in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section); the program itself follows `--`.
set -e
TAG=${1:-rXX}
note() { echo "$(date +%T) $1" >> $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_progress.log; echo "$1"; }
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -- python3 $R/bench.py --cpu-seconds 0 --steps 100 --warmup 10 --also-large 0 --also-other 0 --tagged 0 --kernels 0 --windows 1 > $OUT/step.log 2>&1
note "pass step done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_tagged -- python3 $R/bench.py --cpu-seconds 0 --steps 50 --warmup 5 --also-large 0 --also-other 0 --kernels 0 --windows 1 --tagged 1 > $OUT/step_tagged.log 2>&1
note "pass step_tagged done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_config3 -- python3 $R/bench.py --cpu-seconds 0 --steps 50 --warmup 5 --also-large 0 --also-other 0 --kernels 0 --windows 1 --batch 2048 --tagged 1 --hparams kuairand > $OUT/step_config3.log 2>&1
note "pass step_config3 (B=2048, tagged, kuairand hyper-parameters) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_config5 -- python3 $R/bench.py --cpu-seconds 0 --steps 100 --warmup 10 --also-large 0 --also-other 0 --tagged 0 --kernels 0 --windows 1 --batch 4096 --levels 4 --codes 1024 > $OUT/step_config5.log 2>&1
note "pass step_config5 (B=4096, 4x1024) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_config4 -- python3 $R/bench.py --cpu-seconds 0 --steps 100 --warmup 10 --also-large 0 --also-other 0 --tagged 0 --kernels 0 --windows 1 --batch 8192 > $OUT/step_config4.log 2>&1
note "pass step_config4 (B=8192 shard) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/quoted -- python3 $R/tools/profile_kernels.py > $OUT/quoted.log 2>&1
note "pass quoted done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/profile_kernels.py > $OUT/fetch.log 2>&1
note "pass fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/profile_kernels.py > $OUT/write.log 2>&1
note "pass write done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $R/tools/profile_kernels.py > $OUT/sq.log 2>&1
note "pass sq done"
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $R/tools/profile_kernels.py > $OUT/tcc.log 2>&1
note "pass tcc done"
# L1 / texture-addresser passes, at most 2 counters each (a wider TCP/TA pass aborted inside rocprofv3 with signal 6 in round 2): the logs are kept either way
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcp -- python3 $R/tools/profile_kernels.py > $OUT/${TAG}_tcp_pass.log 2>&1 && python3 $R/tools/pmc_table.py $OUT/tcp > $OUT/${TAG}_quoted_kernels_pmc_tcp.csv || note "tcp pass failed (see ${TAG}_tcp_pass.log)"
note "pass tcp done"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $OUT/ta -- python3 $R/tools/profile_kernels.py > $OUT/${TAG}_ta_pass.log 2>&1 && python3 $R/tools/pmc_table.py $OUT/ta > $OUT/${TAG}_quoted_kernels_pmc_ta.csv || note "ta pass failed (see ${TAG}_ta_pass.log)"
note "pass ta done"
# fold into small CSVs (the raw traces are large)
python3 $R/tools/pmc_summary.py $OUT/fetch $OUT/write > $OUT/${TAG}_quoted_kernels_pmc_hbm_traffic.csv
python3 $R/tools/pmc_table.py $OUT/sq > $OUT/${TAG}_quoted_kernels_pmc_sq.csv
python3 $R/tools/pmc_table.py $OUT/tcc > $OUT/${TAG}_quoted_kernels_pmc_l2.csv
for d in step step_tagged step_config3 step_config4 step_config5 quoted; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${TAG}_${d}_kernel_stats.csv
done
# keep only the folded files in the merge-back (<= 64 MiB)
rm -rf $OUT/step $OUT/step_tagged $OUT/step_config3 $OUT/step_config4 $OUT/step_config5 $OUT/quoted $OUT/fetch $OUT/write $OUT/sq $OUT/tcc $OUT/tcp $OUT/ta
ls -la $OUT
