#!/bin/bash
# Collect rocprofv3 evidence for bench.py's workload on the GPU box: a kernel trace with
# stats plus PMC passes, one counter block budget per run (8 SQ, 4 TCC, 2 GRBM per pass).
# Usage (through gpurun): bash tools/profile_pmc.sh <tag> [bench args...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" $ARGS > "$OUT/kt.log" 2>&1 || exit 1
pass() {
    name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAVES
pass sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_FLAT SQ_INSTS_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "trace_kernel" not in k:
                continue
            args = k[k.index("<") + 1:k.rindex(">")].replace(" ", "").split(",")    # MODE, COUNT, EXT, TREES
            key = "counting" if len(args) > 1 and args[1] in ("true", "1") else "timed"
            e = res.setdefault(key, {"dispatches": {}})
            e["vgpr"], e["sgpr"], e["scratch"], e["grid"], e["wg"] = r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"]
            e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for key, e in res.items():
    e.pop("dispatches")
    for c, v in list(e.items()):
        if isinstance(v, list):
            e[c] = sum(v) / len(v)      # average per dispatch
for f in glob.glob(out + "/kt/**/*_kernel_stats.csv", recursive=True):
    res["kernel_stats"] = [r for r in csv.DictReader(open(f))][:4]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res.get("timed", {}), indent=1))
PY
