#!/bin/bash
# Round profiles (run on the GPU box from the repo root): rocprofv3 kernel trace of the bench command and
# the PMC passes of tools/pmc_run.py, one counter group per run.  Usage: tools/profile_round.sh r02_k
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $root/bench.py --no-cpu-baseline --no-solves > $out/bench_under_profiler.json 2> $out/bench_under_profiler.err || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o pmc -- python3 $root/tools/pmc_run.py > $out/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; exit 1; }
done
cd $root

ls $out
