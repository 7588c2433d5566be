# A/B of the bucket kernels on the GPU box: tools/ab_bucket.sh "4 3 2" "S2|S3 --points 32000000|S5"
set -x
KS=${1:-"4 3"}
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_t.log 2>&1; tail -3 gpurun_out/r02_t.log | cut -c1-300
for K in $KS; do
 for W in "S2" "S3 --points 32000000" "S5"; do
  tag=$(echo $W | cut -d' ' -f1)
  GNDT_BUCKET_KERNEL=$K python3 bench.py --workload $W --steps 10 --no-cpu-baseline --no-extras --stamps > gpurun_out/r02_ab_k${K}_${tag}.json 2> gpurun_out/r02_ab_k${K}_${tag}.err
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r02_ab_k${K}_${tag}.json").read().strip().splitlines()[-1])
print("K$K $tag ms", d["ms_per_step"], "nodes", d["config"]["nodes"], d["phase_ms"])
PY
  tail -1 gpurun_out/r02_ab_k${K}_${tag}.err | cut -c1-330
 done
done
