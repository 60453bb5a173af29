#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root; writes gpurun_out/prof/).
# One profiler per pass: kernel trace + stats, then the PMC passes on their own (never combined with other traces).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-graph-launch-timing"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $B --no-extras > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B --graph 0 --steps 3 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B --graph 0 --steps 3 --warmup 1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sqa -- $B --graph 0 --steps 3 --warmup 1 > $OUT/sqa.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sqb -- $B --graph 0 --steps 3 --warmup 1 > $OUT/sqb.log 2>&1
cd $ROOT
python3 tools/rocpd_summary.py $(ls $OUT/trace/*/*_results.db | head -1) 32 $OUT/kernel_stats.md > /dev/null   # 1 eager + 1 after capture + 5 warm-up + 20 timed + 5 span-timing passes
python3 tools/timeline2.py $(ls $OUT/trace/*/*_results.db | head -1) > $OUT/timeline.txt 2>&1 || true
python3 tools/hbm_traffic.py $(ls $OUT/fetch/*/*_counter_collection.csv | head -1) $(ls $OUT/write/*/*_counter_collection.csv | head -1) 7 $OUT/hbm_traffic.md "HBM traffic per kernel launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x 2: MI355X_MICROARCH.md); config: bf16 B=8 T=16000 L=30 R=64 S=256 wt=${SRWN_FUSE_WT:-1}"   # 1 + 3 + 3 passes
python3 tools/pmc_table.py $OUT/sq_counters.md group_,rowgemm,colgemm,wgrad,headchain $(ls $OUT/sqa/*/*_counter_collection.csv | head -1) $(ls $OUT/sqb/*/*_counter_collection.csv | head -1) > /dev/null
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -n 1 $OUT/bench.json
