#!/bin/bash
# classify the failing worlds of the 6000-world teapot sweep with the checking builds; try the plane-origin test on a build with the old sign test
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_i
mkdir -p $OUT
W=fuzzmesh_5907,fuzzmesh_6029,fuzzmesh_6054,fuzzmesh_6357,fuzzmesh_6528,fuzzmesh_6927,fuzzmesh_7020,fuzzmesh_7151,fuzzmesh_7590,fuzzmesh_7865,fuzzmesh_7914,fuzzmesh_8310,fuzzmesh_10001,fuzzmesh_10134,fuzzmesh_10193,fuzzmesh_10305,fuzzmesh_10322,fuzzmesh_10620,fuzzmesh_10704,fuzzmesh_11010,fuzzmesh_11304,fuzzmesh_11676
echo "== reference order build"
RTAMD_LIB=$PWD/raytracer_amd/csrc/librtamd_reforder_cramer.so timeout -k 10 400 python -u tests/fuzz_sweep.py --worlds $W > $OUT/failing_worlds_reference_order.log 2>&1
grep "^FAILED\|worlds," $OUT/failing_worlds_reference_order.log | paste -s -d' ' | cut -c1-400
echo "== LU build"
RTAMD_LIB=$PWD/raytracer_amd/csrc/librtamd_lu.so timeout -k 10 400 python -u tests/fuzz_sweep.py --worlds fuzzmesh_5907,fuzzmesh_6029,fuzzmesh_8310,fuzzmesh_11010 > $OUT/failing_worlds_lu.log 2>&1
grep "^FAILED\|worlds," $OUT/failing_worlds_lu.log | paste -s -d' ' | cut -c1-400
echo "== plane-origin test, product"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "start_on_the_planes or origin_on_a_split" --timeout 200 --timeout-method thread 2>&1 | tail -3
echo "== plane-origin test, old sign test"
RTAMD_LIB=$PWD/raytracer_amd/csrc/librtamd_oldsign.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "start_on_the_planes" --timeout 200 --timeout-method thread 2>&1 | grep "AssertionError\|passed\|failed" | cut -c1-300
