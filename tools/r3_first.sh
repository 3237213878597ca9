#!/bin/bash
# First evidence pass of round 3 (through gpurun): bash tools/r3_first.sh
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3_a
mkdir -p "$OUT"
cd "$R"
echo "== frames of the emitter-beside-the-box world: product library, library without the near-limit guard"
timeout -k 10 200 python tools/frame_digest.py panel_between_floor_and_box raytracer_amd/csrc/librtamd.so raytracer_amd/csrc/librtamd_prefix.so > "$OUT/fused_guard_frames.txt" 2>&1 || { cat "$OUT/fused_guard_frames.txt"; exit 1; }
cat "$OUT/fused_guard_frames.txt"
echo "== GPU tests"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1 || { tail -30 "$OUT/gpu_tests.log"; exit 1; }
tail -1 "$OUT/gpu_tests.log"
echo "== bench"
timeout -k 10 400 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python - "$OUT/bench_line.json" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print("value %.1f %s, ms/step %.3f, kernel %.3f ms, cpu %s" % (b["value"], b["unit"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["cpu_baseline"]))
print(b["ranks"])
PY
echo "== phases"
timeout -k 10 200 python tools/phase_profile.py > "$OUT/phase_cycles.json" 2> "$OUT/phase.err" || { tail -5 "$OUT/phase.err"; exit 1; }
python - "$OUT/phase_cycles.json" <<'PY'
import json, sys
p = json.load(open(sys.argv[1]))
print(json.dumps(p.get("shadow_walks"), indent=1)); print(p.get("shadow_rounds"))
PY
echo "== two ranks over gloo on one GPU"
RTAMD_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/gloo2.json" 2> "$OUT/gloo2.err" || { tail -20 "$OUT/gloo2.err"; exit 1; }
python - "$OUT/gloo2.json" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print("gloo2 value %.1f, ms/step %.3f, ranks %s" % (b["value"], b["ms_per_step"], b["ranks"]))
PY
echo "== what FETCH_SIZE reports for known reads"
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- "$R/tools/calib/fetch_calib" > "$OUT/calib_fetch.log" 2>&1 || { tail -5 "$OUT/calib_fetch.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/calib_rdreq" -- "$R/tools/calib/fetch_calib" > "$OUT/calib_rdreq.log" 2>&1 || { tail -5 "$OUT/calib_rdreq.log"; exit 1; }
python3 "$R/tools/pmc_summary.py" "$OUT/calib_fetch" "$OUT/calib_rdreq" > "$OUT/fetch_calibration.json"
cat "$OUT/fetch_calibration.json"
echo "== the same request counters for the bench kernel"
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$OUT/bench_rdreq" -- python3 "$R/bench.py" --pmc-probe --steps 2 --warmup 1 > "$OUT/bench_rdreq.log" 2>&1 || { tail -5 "$OUT/bench_rdreq.log"; exit 1; }
python3 "$R/tools/pmc_summary.py" "$OUT/bench_rdreq" > "$OUT/bench_requests.json"
cat "$OUT/bench_requests.json"
rm -rf "$OUT/calib_fetch" "$OUT/calib_rdreq" "$OUT/bench_rdreq"
