#!/bin/bash
# A/B of library builds on the full mesh configuration (BASELINE configs[2]: cold and carried-over frame times from bench.py).
for v in "$@"; do
  if [ "$v" = default ]; then unset WBRT_LIB; else export WBRT_LIB=$PWD/opencl-ray-tracer_amd/variants/$v.so; fi
  timeout -k 10 300 python bench.py --config mesh --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$v', 'cold', d['ms_per_step'], 'carried', c.get('ms_per_step_schedule_carried_over'), 'capped', c.get('ms_per_step_in_WBRT_MESH_BVH_CAPPED'))" || exit 1
done
