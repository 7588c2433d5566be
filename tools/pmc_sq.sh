#!/bin/bash
# SQ counters per kernel for one bench.py invocation (run on the GPU box):  tools/pmc_sq.sh <tag> [bench args...]
# Two passes of 8 SQ counters each (rocprofv3 --pmc only, no tracing beside it), summarised per kernel name.
. "$(dirname "$0")/_single_process_guard.sh"
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
i=1
for P in "$P1" "$P2"; do
  rm -rf $out/${tag}_sq$i
  rocprofv3 --pmc $P --output-format csv -d $out/${tag}_sq$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-configs "$@" > $out/${tag}_sq$i.log 2>&1
  i=$((i+1))
done
python3 - "$out/${tag}_sq1" "$out/${tag}_sq2" <<'PY' | tee $out/${tag}_sq.txt
import collections, csv, glob, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if not k.startswith("gndt::"): continue
            k = k[6:80]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
    c = {a: tot[k][a] / max(n[k][a], 1) for a in tot[k]}
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e5: continue
    pct = lambda a: 100.0 * c.get(a, 0) / wc
    print(f"{k}\n   wave_cycles {wc:.3g}  parked(WAIT_ANY) {pct('SQ_WAIT_ANY'):.0f}%  issue-stall(WAIT_INST_ANY) {pct('SQ_WAIT_INST_ANY'):.0f}%  active {pct('SQ_ACTIVE_INST_ANY'):.0f}%"
          f"  [valu {pct('SQ_ACTIVE_INST_VALU'):.0f}% lds {pct('SQ_ACTIVE_INST_LDS'):.0f}% sca {pct('SQ_ACTIVE_INST_SCA'):.0f}%]  busy_cycles {c.get('SQ_BUSY_CYCLES',0):.3g}\n"
          f"   insts: valu {c.get('SQ_INSTS_VALU',0):.3g} salu {c.get('SQ_INSTS_SALU',0):.3g} lds {c.get('SQ_INSTS_LDS',0):.3g} vmem {c.get('SQ_INSTS_VMEM',0):.3g}"
          f"  lds_idx_active {c.get('SQ_LDS_IDX_ACTIVE',0):.3g} lds_bank_conflict {c.get('SQ_LDS_BANK_CONFLICT',0):.3g} wait_inst_lds {c.get('SQ_WAIT_INST_LDS',0):.3g}")
PY
rm -rf $out/${tag}_sq1 $out/${tag}_sq2
