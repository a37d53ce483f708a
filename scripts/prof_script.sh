#!/bin/bash
# Profiles a scratch script on the GPU box: kernel trace + stats, then two PMC passes (one run each).
# Usage: scripts/prof_script.sh <tag> <script.py> [arguments]    -> gpurun_out/prof_<tag>/...
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/prof_$TAG; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 "$@" > $O/stats.log 2>&1 || { echo stats failed; tail -5 $O/stats.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc -- python3 "$@" > $O/pmc.log 2>&1 || { echo pmc failed; tail -5 $O/pmc.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- python3 "$@" > $O/pmc2.log 2>&1 || { echo pmc2 failed; tail -5 $O/pmc2.log; }
echo done $TAG
