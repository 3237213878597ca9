#!/bin/bash
# Parameter sweeps on the bench frame after the shadow tree learned to split triangles: bash tools/r3_sweep.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3_d
mkdir -p "$OUT"
cd "$R"
L=raytracer_amd/csrc
{
echo "== round limits of the shadow walk (hot-only builds)"
timeout -k 10 300 python tools/quick_bench.py $L/librtamd_hot.so $L/librtamd_loads2.so $L/librtamd_loads4.so $L/librtamd_skips2.so || exit 1
for leaf in 12 16 20 24; do
  for step in 0.5 1.0 2.0; do
    echo "== shadow tree leaf rows $leaf, step cost $step"
    RTAMD_SWEEP_LEAF=$leaf RTAMD_SWEEP_STEP=$step timeout -k 10 200 python tools/quick_bench.py --steps 5 $L/librtamd_hot.so || exit 1
  done
done
} 2>&1 | tee "$OUT/sweep.txt"
