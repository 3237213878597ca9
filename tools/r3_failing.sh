#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3_g
mkdir -p $OUT
W=fuzz_4973,fuzz_5292,fuzz_6054,fuzz_6183,fuzz_6576,fuzz_7686,fuzz_7884,fuzz_8364,fuzz_9348,fuzzmesh_1929,fuzzmesh_2031,fuzzmesh_2340,fuzzmesh_2559,fuzzmesh_2583,fuzzmesh_2807,fuzzmesh_2924,fuzzmesh_3161,fuzzmesh_3188,fuzzmesh_3417,fuzzmesh_3663,fuzzmesh_4404,fuzzmesh_5175,fuzzmesh_5181,fuzzmesh_5234
timeout -k 10 300 python -u tests/fuzz_sweep.py --worlds $W > $OUT/failing_worlds_product.log 2>&1
grep "^FAILED\|^AssertionError\|worlds," $OUT/failing_worlds_product.log | paste - - | cut -c1-200
echo "== reference order build"
RTAMD_LIB=$PWD/raytracer_amd/csrc/librtamd_reforder_cramer.so timeout -k 10 400 python -u tests/fuzz_sweep.py --worlds $W > $OUT/failing_worlds_reference_order.log 2>&1
grep "^FAILED\|^AssertionError\|worlds," $OUT/failing_worlds_reference_order.log | paste - - | cut -c1-200
echo "== around fuzzmesh_2924, three times"
for i in 1 2 3; do
timeout -k 10 200 python -u tests/fuzz_sweep.py 0 0 2905 2930 > $OUT/around_2924_$i.log 2>&1
grep "^FAILED\|^AssertionError\|worlds," $OUT/around_2924_$i.log | paste - - | cut -c1-200
done
