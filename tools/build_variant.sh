#!/bin/bash
# Build an experiment variant of the library holding the benchmark's whole-frame kernel only:
#   tools/build_variant.sh NAME [-DFLAG ...]   ->  raytracer_amd/csrc/librtamd_NAME.so  (seconds, not minutes)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off ${RT_SCOPE:--DRT_ONLY_HOT} "$@" \
    -o raytracer_amd/csrc/librtamd_$name.so raytracer_amd/csrc/rt_kernels.hip raytracer_amd/csrc/rt_kdbuild.hip raytracer_amd/csrc/rt_shadowtree.cpp
