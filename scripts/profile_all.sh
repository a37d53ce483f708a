#!/bin/bash
# rocprofv3 summaries of every BASELINE configuration (kernel trace + PMC passes) -> gpurun_out/prof_r3_<config>/
# usage: scripts/profile_all.sh [configs...]   (default: readme readme_tolerant mesh uhd stress)
set -u
CONFIGS=${@:-readme readme_tolerant mesh uhd stress}
for c in $CONFIGS; do
  case $c in
    readme)          CONFIG_NAME=readme STEPS_STATS=10 WARMUP_STATS=2 STEPS_PMC=3 WARMUP_PMC=1 bash scripts/profile_bench.sh r3_readme || exit 1 ;;
    readme_tolerant) CONFIG_NAME=readme_tolerant STEPS_STATS=10 WARMUP_STATS=2 STEPS_PMC=3 WARMUP_PMC=1 bash scripts/profile_bench.sh r3_readme_tolerant --arith tolerant || exit 1 ;;
    mesh)            CONFIG_NAME=mesh STEPS_STATS=6 WARMUP_STATS=1 STEPS_PMC=2 WARMUP_PMC=1 bash scripts/profile_bench.sh r3_mesh --config mesh || exit 1 ;;
    uhd)             CONFIG_NAME=uhd STEPS_STATS=2 WARMUP_STATS=1 STEPS_PMC=1 WARMUP_PMC=0 bash scripts/profile_bench.sh r3_uhd --config uhd || exit 1 ;;
    stress)          CONFIG_NAME=stress STEPS_STATS=1 WARMUP_STATS=1 STEPS_PMC=1 WARMUP_PMC=0 bash scripts/profile_bench.sh r3_stress --config stress || exit 1 ;;
  esac
done
