#!/bin/bash
# PMC passes over scripts/lone_wave.py (one wave alone on the GPU): instruction mix of the critical path.
# Usage: scripts/profile_lone.sh <tag>   -> gpurun_out/prof_<tag>/summary.txt   (WBRT_LIB is honoured)
set -u
TAG=${1:-lone}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
B="python3 scripts/lone_wave.py"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq pass failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || { echo "pmc sq2 pass failed"; tail -5 $OUT/pmc_sq2.log; exit 1; }
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
