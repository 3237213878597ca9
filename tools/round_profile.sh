#!/bin/bash
# Evidence for profiles/<tag>: GPU tests, the bench line (with its live counter passes), the
# rocprofv3 kernel trace of the same bench command, the phase split.  Run through gpurun:
#   bash tools/round_profile.sh <tag>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1 || { tail -5 "$OUT/gpu_tests.log"; exit 1; }
tail -1 "$OUT/gpu_tests.log"
timeout -k 10 400 python bench.py --save-pmc > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cp profiles/pmc_dragon_1080p_16spp_ao_arealight.json "$OUT/pmc.json"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --no-pmc --no-cpu-baseline > "$OUT/bench_line_profiled.json" 2> "$OUT/kt.err" || { tail -5 "$OUT/kt.err"; exit 1; }
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/kt"
cd "$R"
timeout -k 10 200 python tools/phase_profile.py > "$OUT/phase_cycles.json" 2>/dev/null || exit 1
python - "$OUT" <<'PY'
import json, sys, csv
out = sys.argv[1]
b = json.load(open(out + "/bench_line.json"))
print("value %.1f %s, kernel %.3f ms, binding %s" % (b["value"], b["unit"], b["roofline"]["kernel_ms"], b["roofline"]["binding"]))
for k, v in b["roofline"]["ceilings"].items():
    print("  %-5s frac %.3f" % (k, v["frac"]))
for r in list(csv.DictReader(open(out + "/kernel_stats.csv")))[:3]:
    print("  rocprof: %s calls %s avg %.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
