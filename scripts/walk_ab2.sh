#!/bin/bash
# f1_car scene in the proven and capped walks at 16 and at 100 spp for several library builds (kernel times of bvh_check.py)
for v in "$@"; do
  if [ "$v" = default ]; then unset WBRT_LIB; else export WBRT_LIB=$PWD/opencl-ray-tracer_amd/variants/$v.so; fi
  for spp in 16 100; do
    echo "== $v $spp spp"; timeout -k 10 200 python scripts/bvh_check.py 1920 $spp 2>&1 | grep -E "mode=[12]" || exit 1
  done
done
