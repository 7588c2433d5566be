#!/bin/bash
# PC sampling of the bench workload on the GPU box (rocprofv3 beta feature; stochastic = hardware sampling with stall reasons).
#   tools/pc_sample.sh <tag> [bench args...]   -> gpurun_out/<tag>_pcsamp/
tag=$1; shift
. "$(dirname "$0")/_single_process_guard.sh"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_pcsamp
rm -rf $out; mkdir -p $out
for M in stochastic host_trap; do
  U=cycles; I=1048576
  if [ $M = host_trap ]; then U=time; I=1; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $U --pc-sampling-method $M --pc-sampling-interval $I --kernel-trace --output-format csv -d $out/$M -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-configs "$@" > $out/$M.log 2>&1
  echo "$M rc=$?"; tail -3 $out/$M.log; find $out/$M -type f | head; 
done
