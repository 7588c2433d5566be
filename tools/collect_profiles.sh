#!/bin/bash
# Run on the MI355X box (gpurun): kernel-trace statistics and the two PMC passes of the bench workload.
#   tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_{kernel_stats.csv,pmc.json,bench.json}
. "$(dirname "$0")/_single_process_guard.sh"
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-configs > $out/${tag}_trace.log 2>&1
cp $out/${tag}_trace/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-configs > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-configs > $out/${tag}_write.log 2>&1
python3 tools/summarise_pmc.py $out/${tag}_fetch $out/${tag}_write > $out/${tag}_pmc.json
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rm -rf $out/${tag}_trace $out/${tag}_fetch $out/${tag}_write
head -12 $out/${tag}_kernel_stats.csv | cut -c1-200
tail -1 $out/${tag}_bench.json | cut -c1-400
