#!/bin/bash
# Everything the round's profiles/ set is made of, in one call on the MI355X box:   tools/collect_round.sh r06_c
#   <tag>_kernel_stats.csv, <tag>_pmc.json, <tag>_bench.json   (tools/collect_profiles.sh: kernel trace, the two PMC passes, the default bench line)
#   <tag>_sq.txt                                               (tools/pmc_sq.sh: SQ counters per kernel)
#   <tag>_timeline.txt                                         (tools/trace_s2.sh: start / duration / gap of the kernels of one steady-state build)
#   <tag>_bench_S3_100M.json, <tag>_bench_S5_20M.json          (configs[2] and configs[4] at full size on one GPU)
#   <tag>_small.json                                           (tools/measure_small.py: configs[0]-sized frames, eager and replayed)
#   <tag>_first_builds.txt                                     (tools/first_build_probe.py: first builds of fresh and of warmed handles)
tag=${1:-round}
out=gpurun_out
bash tools/collect_profiles.sh $tag > $out/${tag}_collect.log 2>&1
bash tools/pmc_sq.sh $tag > /dev/null 2>&1
bash tools/trace_s2.sh ${tag}_tl > /dev/null 2>&1; mv $out/${tag}_tl_timeline.txt $out/${tag}_timeline.txt; rm -f $out/${tag}_tl_kernel_stats.csv $out/${tag}_tl_tr.log
python3 bench.py --workload S3 --points 100000000 --steps 10 --no-cpu-baseline --no-extras --no-configs > $out/${tag}_bench_S3_100M.json 2> $out/${tag}_s3.err
python3 bench.py --workload S5 --steps 10 --no-cpu-baseline --no-extras --no-configs > $out/${tag}_bench_S5_20M.json 2> $out/${tag}_s5.err
python3 tools/measure_small.py > $out/${tag}_small.json 2> /dev/null
python3 tools/first_build_probe.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_first_builds.txt
tail -1 $out/${tag}_bench.json | cut -c1-300
head -9 $out/${tag}_kernel_stats.csv | cut -c1-60,170-
for f in S3_100M S5_20M; do python3 -c "
import json,sys
d=json.loads([l for l in open('$out/${tag}_bench_$f.json') if l.startswith('{')][0]); print('$f', d['ms_per_step'], d['config']['nodes'], d['retries_in_timed_region'], {k:round(v,3) for k,v in d['phase_ms'].items() if v>0.01})"; done
