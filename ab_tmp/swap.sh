#!/bin/bash
# usage: swap.sh old|new
cp ab_tmp/libcamcal_$1.so cameracalibration_amd/csrc/libcamcal.so
cp ab_tmp/rig_$1.py cameracalibration_amd/rig.py
cp ab_tmp/_lib_$1.py cameracalibration_amd/_lib.py
cp ab_tmp/engine_$1.py cameracalibration_amd/engine.py
