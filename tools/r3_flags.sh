#!/bin/bash
# compiler-flag variants of the hot kernel, and the merged shadow-tree call on the general kernels
set -e
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_g
mkdir -p $OUT
L=raytracer_amd/csrc
timeout -k 10 300 python tools/quick_bench.py --steps 10 $L/librtamd_base.so $L/librtamd_sched_max-ilp.so $L/librtamd_sched_max-memory-clause.so $L/librtamd_sched_iterative-ilp.so $L/librtamd_sched_iterative-minreg.so $L/librtamd_nounroll.so $L/librtamd_o2.so $L/librtamd_licm.so $L/librtamd_hoist.so $L/librtamd_prio_walk1.so $L/librtamd_prio_walk3.so $L/librtamd_prio_leaf1.so $L/librtamd_prio_leaf3.so $L/librtamd_base.so > $OUT/compiler_flags.txt 2>&1
cat $OUT/compiler_flags.txt
timeout -k 10 300 python tools/config_rates.py --no-pmc --frames 7 "j dragon" "mirror teapot" "J dragon" "K teapot" "2 cornellbox" > $OUT/rates_before_merge.txt 2>&1
timeout -k 10 300 python tools/config_rates.py --no-pmc --frames 7 --lib $L/librtamd_merged.so "j dragon" "mirror teapot" "J dragon" "K teapot" "2 cornellbox" > $OUT/rates_merged.txt 2>&1
python - <<'PY'
import json
for n in ("before_merge","merged"):
    print(n)
    for l in open("gpurun_out/r3_g/rates_%s.txt" % n):
        if l.startswith("{"):
            r=json.loads(l); print("  %-60s %8.3f ms %9.1f Mrays/s %s" % (r["config"], r["ms_per_frame"], r["Mrays_per_s"], r["frame_digest"]))
PY
