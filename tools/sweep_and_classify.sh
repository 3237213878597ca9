#!/bin/bash
# A random-world sweep with the product library, then its failing worlds again under the checking builds
# (-DRT_REF_ORDER, then -DRT_TRI_LU, then -DRT_REF_ORDER -DRT_NO_REFILL); what no build explains is printed last.
#   bash tools/sweep_and_classify.sh TAG FIRST LAST MESH_FIRST MESH_LAST      (through gpurun; the three
#   checking libraries are built beforehand with tools/build_variant.sh: reforder_cramer, lu, reforder_norefill)
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
L=$PWD/raytracer_amd/csrc
timeout -k 10 ${SWEEP_SECONDS:-900} python -u tests/fuzz_sweep.py "$@" > $OUT/sweep.log 2>&1
tail -1 $OUT/sweep.log
names() { grep '^FAILED' "$1" | awk '{print $2}' | paste -s -d, ; }
W=$(names $OUT/sweep.log)
grep -c "frame differs" $OUT/sweep.log | sed 's/^/worlds whose timed frame differs from the reference-walk kernel: /'
for lib in reforder_cramer lu reforder_norefill; do
    [ -z "$W" ] && break
    RTAMD_LIB=$L/librtamd_$lib.so timeout -k 10 300 python -u tests/fuzz_sweep.py --worlds $W > $OUT/failing_$lib.log 2>&1
    echo "under $lib: $(tail -1 $OUT/failing_$lib.log)"
    W=$(names $OUT/failing_$lib.log)
done
echo "explained by no checking build: ${W:-none}"
