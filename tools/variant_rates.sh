#!/bin/bash
# tools/variant_rates.sh "<config label substrings>" NAME...: tools/config_rates.py (no counters) for several library builds
sel=$1; shift
for v in "$@"; do
  echo "== $v"
  timeout -k 10 300 python tools/config_rates.py --no-pmc --lib raytracer_amd/csrc/librtamd_$v.so $sel || exit 1
done
