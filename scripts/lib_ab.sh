#!/bin/bash
# A/B of library builds on the headline frame (cold, strict and tolerant): usage scripts/lib_ab.sh <rounds> <variant names under
# opencl-ray-tracer_amd/variants, or "default">
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset WBRT_LIB; else export WBRT_LIB=$PWD/opencl-ray-tracer_amd/variants/$v.so; fi
    s=$(timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
    t=$(timeout -k 10 100 python3 bench.py --no-cpu-baseline --no-extras --arith tolerant 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
    echo "$v strict $s tolerant $t"
  done
done
