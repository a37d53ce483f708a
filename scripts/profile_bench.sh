#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (one run per pass).
# Usage: scripts/profile_bench.sh <tag> [bench.py arguments]    -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-run}
shift || true
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --no-cpu-baseline --no-extras $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps ${STEPS_STATS:-10} --warmup ${WARMUP_STATS:-2} > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $B --steps ${STEPS_PMC:-3} --warmup ${WARMUP_PMC:-1} > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq pass failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $B --steps ${STEPS_PMC:-3} --warmup ${WARMUP_PMC:-1} > $OUT/pmc_sq2.log 2>&1 || { echo "pmc sq2 pass failed"; tail -5 $OUT/pmc_sq2.log; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps ${STEPS_PMC:-3} --warmup ${WARMUP_PMC:-1} > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch pass failed"; tail -5 $OUT/pmc_fetch.log; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps ${STEPS_PMC:-3} --warmup ${WARMUP_PMC:-1} > $OUT/pmc_write.log 2>&1 || { echo "pmc write pass failed"; tail -5 $OUT/pmc_write.log; }
python3 scripts/summarize_prof.py $OUT ${CONFIG_NAME:-readme} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
