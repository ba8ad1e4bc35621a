#!/bin/bash
# A/B of matching-library builds on one box: tools/exp/ab_match.sh lib1.so lib2.so ...  (two rounds each)
set -e
for round in 1 2; do
  for lib in "$@"; do
    STEREO_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/match_time.py 2>/dev/null | tail -1
  done
done
