#!/bin/bash
# A/B helper for timing on ONE gpurun box (boxes differ by several percent, so a change is
# only judged against the previous build inside the same call).  Before the call, on the CPU:
#   mkdir -p ab_tmp; copy the files of the two builds to ab_tmp/<name>_old.<ext> / _new.<ext>
#   for libcamcal.so, rig.py, _lib.py, engine.py
# then on the box:   tools/ab_swap.sh old; <timing>; tools/ab_swap.sh new; <timing>
set -e
cp ab_tmp/libcamcal_$1.so cameracalibration_amd/csrc/libcamcal.so
for f in rig _lib engine; do
  [ -f ab_tmp/${f}_$1.py ] && cp ab_tmp/${f}_$1.py cameracalibration_amd/$f.py
done
exit 0
