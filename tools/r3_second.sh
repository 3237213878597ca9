#!/bin/bash
# Second evidence pass of round 3 (through gpurun): bash tools/r3_second.sh TAG
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd "$R"
echo "== GPU tests"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1 || { tail -30 "$OUT/gpu_tests.log"; exit 1; }
tail -1 "$OUT/gpu_tests.log"
echo "== bench"
timeout -k 10 400 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python - "$OUT/bench_line.json" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print("value %.1f %s, ms/step %.3f, kernel %.3f ms" % (b["value"], b["unit"], b["ms_per_step"], b["roofline"]["kernel_ms"]))
print({k: round(v["frac"], 3) for k, v in b["roofline"]["ceilings"].items()}, b["roofline"]["executed"]["per_ray"])
PY
echo "== the worlds of the sweeps whose samples differed: product library, then the checking build that sums a path in the reference's order (-DRT_REF_ORDER -DRT_NO_REFILL)"
WORLDS=fuzz_232,fuzz_294,fuzz_1134,fuzzmesh_42,fuzzmesh_762,fuzz_2511,fuzzmesh_1160,fuzz_1337,fuzzmesh_47
timeout -k 10 300 python tests/fuzz_sweep.py --worlds $WORLDS > "$OUT/knife_edge_worlds_product.log" 2>&1; grep -a "FAILED\|worlds," "$OUT/knife_edge_worlds_product.log"
RTAMD_LIB=$R/raytracer_amd/csrc/librtamd_reforder_cramer.so timeout -k 10 300 python tests/fuzz_sweep.py --worlds $WORLDS > "$OUT/knife_edge_worlds_reference_order.log" 2>&1; grep -a "FAILED\|worlds," "$OUT/knife_edge_worlds_reference_order.log"
echo "== configurations"
timeout -k 10 600 python tools/config_rates.py --out "$OUT/rates.json" > "$OUT/rates.log" 2>&1 || { tail -5 "$OUT/rates.log"; exit 1; }
python - "$OUT/rates.json" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    print("%-60s %8.3f ms %9.1f Mrays/s lanes %s valu %s %s" % (r["config"][:60], r["ms_per_frame"], r["Mrays_per_s"], r.get("lane_utilisation"), r.get("valu_busy"), r.get("kernel", "")[:40]))
PY
