#!/bin/bash
# One gpurun call = one invocation of this script on the GPU box:  scripts/gpu_round.sh <tag> <stage> [<stage> ...]
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
# Stages:
#   tests           python -m pytest tests -m gpu
#   smoke           __graft_entry__.smoke()
#   bench[:ARGS]    python bench.py ARGS            (ARGS with ',' for spaces; output bench_<n>.json / .err)
#   pmc[:ARGS]      four rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / L2 requests / VALU) + a --kernel-trace --stats pass over
#                   `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats ARGS`, reduced to per-kernel CSVs
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
n=0
for st in "$@"; do
  name=${st%%:*}; args=""; [ "$st" != "$name" ] && args=$(echo "${st#*:}" | tr ',' ' ')
  case $name in
    tests) timeout 1500 python -m pytest tests -m gpu -q -x $args 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench) n=$((n+1)); echo "== bench $args"; timeout 1200 python bench.py $args > $O/bench_$n.json 2> $O/bench_$n.err; echo "$args" > $O/bench_$n.args
           grep -E "per-launch|host ABI|xrl host|cpu reference|Error|error" $O/bench_$n.err | cut -c1-1500; python -c "
import json,sys
try:
    d=json.load(open('$O/bench_$n.json')); print({k:d.get(k) for k in ('value','ms_per_step','value_host_abi','parity')}); print(d['roofline'].get('per_kernel_ms_per_step'))
except Exception as e: print('no json', e)
" ;;
    pmc) cd /tmp && export TMPDIR=/tmp
         B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats $args"
         timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
         timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
         timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- $B > $O/pmc_l2.log 2>&1
         timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
         timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi --no-stats $args > $O/ktrace.log 2>&1
         python - $O <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "xrl::" in r["Kernel_Name"]]
        with open(f"{O}/{d}.csv", "w", newline="") as out:
            w = csv.DictWriter(out, fieldnames=["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
            w.writeheader(); w.writerows(rows)
        agg = collections.defaultdict(list)
        for r in rows: agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()): print(d, k, "n=%d mean=%.5g" % (len(v), sum(v) / len(v)))
    os.system(f"rm -rf {O}/{d}")
for f in glob.glob(f"{O}/ktrace/**/*kernel_stats.csv", recursive=True):
    os.system(f"cp {f} {O}/kernel_stats.csv"); print(open(f).read()[:3000])
os.system(f"rm -rf {O}/ktrace")
PY
         cd $R ;;
    pmcset) # pmcset:COUNTER,COUNTER,...[@BENCHARG,BENCHARG...]  one rocprofv3 --pmc pass with the given counters, per-kernel means printed
         n=$((n+1)); cs=${args%%@*}; ba=""; [ "$args" != "$cs" ] && ba=${args#*@}
         cd /tmp && export TMPDIR=/tmp
         timeout 300 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $O/pmcset_$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats $ba > $O/pmcset_$n.log 2>&1
         python - $O/pmcset_$n <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "xrl::" in r["Kernel_Name"]]
    with open(d + ".csv", "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
        w.writeheader(); w.writerows(rows)
    agg = collections.defaultdict(list)
    for r in rows: agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(k, "n=%d mean=%.5g" % (len(v), sum(v) / len(v)))
os.system(f"rm -rf {d}")
PY
         tail -3 $O/pmcset_$n.log; cd $R ;;
    *) echo "unknown stage $name" ;;
  esac
done
ls $O
