#!/bin/bash
# A/B of library builds on the two walk workloads: f1_car 1920x1080x16 and the stress scene 1920x1080x16.
# usage: scripts/walk_ab.sh [variant names under opencl-ray-tracer_amd/variants, or "default"]
for v in "$@"; do
  if [ "$v" = default ]; then unset WBRT_LIB; else export WBRT_LIB=$PWD/opencl-ray-tracer_amd/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 200 python scripts/bvh_check.py 1920 16 2>&1 | grep -E "mode=[12]|differing" || exit 1
  timeout -k 10 200 python scripts/stress_quick.py 2>&1 | grep -E "proven|capped|identical" || exit 1
done
