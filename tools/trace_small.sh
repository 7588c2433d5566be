#!/bin/bash
# Run on the MI355X box: kernel trace of tools/measure_small.py -> gpurun_out/<tag>_small_kernel_stats.csv and the raw trace
# (start / end of every launch: the gaps between the kernels of one small build).   tools/trace_small.sh <tag>
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_smalltrace -- python3 tools/measure_small.py > $out/${tag}_small_traced.json 2> $out/${tag}_small_traced.err
cp $out/${tag}_smalltrace/*/*kernel_stats.csv $out/${tag}_small_kernel_stats.csv
cp $out/${tag}_smalltrace/*/*kernel_trace.csv $out/${tag}_small_kernel_trace.csv
rm -rf $out/${tag}_smalltrace
head -14 $out/${tag}_small_kernel_stats.csv | cut -c1-160
