#!/bin/bash
# Bench several builds of the library side by side: tools/variant_bench.sh NAME...
# (each NAME is raytracer_amd/csrc/librtamd_NAME.so; "base" is the product library)
mkdir -p gpurun_out
for v in "$@"; do
  lib=raytracer_amd/csrc/librtamd_$v.so
  [ "$v" = base ] && lib=raytracer_amd/csrc/librtamd.so
  RTAMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/variant_$v.json 2> gpurun_out/variant_$v.err || { echo "$v failed"; tail -3 gpurun_out/variant_$v.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/variant_%s.json" % sys.argv[1]))
print("%-12s %8.1f Mrays/s  kernel %.2f ms  mean %s" % (sys.argv[1], d["value"], d["roofline"]["kernel_ms"], d["config"]["frame_mean_rgb"]))
PY
done
