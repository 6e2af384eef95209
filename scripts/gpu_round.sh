#!/bin/bash
# One gpurun call = one invocation of this script on the GPU box:  scripts/gpu_round.sh <tag> <stage> [<stage> ...]
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
# Stages:
#   tests[:ARGS]    python -m pytest -m gpu ARGS   (default: tests; ',' separates arguments, '+' is a space inside one, e.g. tests:tests/test_gpu_fuzz.py,-k,fuzz+or+goldens)
#   smoke           __graft_entry__.smoke()
#   bench[:ARGS]    python bench.py ARGS            (ARGS with ',' for spaces; output bench_<n>.json / .err)
#   so[:NAME]       the following stages load pecos_amd/lib/NAME (a kernel-tuning variant built by scripts/build_variant.sh) instead of libxrl_amd.so; empty = back to the default
#   probe[:ARGS]    python scripts/host_abi_probe.py ARGS   (fresh process; per-call times + the library's stage timing lines)
#   ab:ROWS@CONFIG@STEPS@SET|SET|...   scripts/ab.py on the first ROWS rows (0 = all) of CONFIG: option sets (k=v+k=v, "" = defaults) timed in one process,
#                   outputs compared bit for bit with the first set's; XRL_SO=<variant .so> selects a library build
#   pmc[:ARGS]      two rocprofv3 --pmc passes (FETCH_SIZE + SQ counters | WRITE_SIZE + L2 hit / miss / requests) + a --kernel-trace --stats pass over
#                   `bench.py --steps 4 --warmup 6 --no-cpu-baseline --no-host-abi --no-stats ARGS` (last 4 steps counted), reduced to per-kernel CSVs and one pmc_traffic entry
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
n=0
for st in "$@"; do
  name=${st%%:*}; args=""; [ "$st" != "$name" ] && args=$(echo "${st#*:}" | tr ',' ' ')
  case $name in
    tests) targs=(); if [ "$st" != "$name" ]; then IFS=',' read -ra tl <<< "${st#*:}"; for x in "${tl[@]}"; do targs+=("${x//+/ }"); done; fi   # (',' separates arguments, '+' stands for a space inside one: -k,fuzz+or+goldens)
           [ ${#targs[@]} -eq 0 ] && targs=(tests)
           timeout ${XRL_TESTS_TIMEOUT:-1500} python -m pytest -m gpu -q -x "${targs[@]}" 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench) n=$((n+1)); echo "== bench $args"; timeout ${XRL_BENCH_TIMEOUT:-600} python bench.py $args > $O/bench_$n.json 2> $O/bench_$n.err; echo "$args" > $O/bench_$n.args
           grep -E "per-launch|host ABI|xrl host|cpu reference|Error|error" $O/bench_$n.err | cut -c1-1500; python -c "
import json,sys
try:
    d=json.load(open('$O/bench_$n.json')); print({k:d.get(k) for k in ('value','ms_per_step','value_host_abi','parity')}); print(d['roofline'].get('per_kernel_ms_per_step'))
except Exception as e: print('no json', e)
" ;;
    so) if [ -n "$args" ]; then export PECOS_XRL_AMD_SO=$R/pecos_amd/lib/$args; else unset PECOS_XRL_AMD_SO; fi; echo "== library: ${PECOS_XRL_AMD_SO:-default}" ;;
    probe) n=$((n+1)); timeout 600 python scripts/host_abi_probe.py $args > $O/probe_$n.json 2> $O/probe_$n.err; grep -E "probe|xrl host" $O/probe_$n.err | cut -c1-400; cat $O/probe_$n.json ;;
    ab) n=$((n+1)); IFS='@' read -r ab_rows ab_cfg ab_steps ab_sets <<< "${st#*:}"
        IFS='|' read -ra ab_list <<< "$ab_sets"; ab_args=(); for x in "${ab_list[@]}"; do ab_args+=("$(echo "$x" | tr '+' ',')"); done
        [ ${#ab_args[@]} -eq 0 ] && ab_args=("")
        ab_scale=1.0; case $ab_cfg in *:*) ab_scale=${ab_cfg#*:}; ab_cfg=${ab_cfg%%:*} ;; esac   # CONFIG:SCALE
        echo "== ab rows=$ab_rows $ab_cfg x$ab_scale"; AB_ROWS=$ab_rows timeout 900 python scripts/ab.py $ab_cfg $ab_scale $ab_steps "${ab_args[@]}" > $O/ab_$n.log 2>&1; grep -E "ms/step|Error|error" $O/ab_$n.log | cut -c1-600 ;;
    pmc) # pmc:BENCHARG,BENCHARG,...   two counter passes (TCC + TCP + SQ blocks have separate slots) + one --kernel-trace --stats pass; writes
         # pmc_<n>_{fetch,write,l2,sq}.csv, kernel_stats_<n>.csv and the pmc_traffic entry pmc_entry_<n>.json (scripts/pmc_traffic.py)
         n=$((n+1)); cd /tmp && export TMPDIR=/tmp XRL_STEP_MARKER=1
         B="python $R/bench.py --steps 4 --warmup 6 --no-cpu-baseline --no-host-abi --no-stats --no-extra --parity-rows 0 $args"
         timeout ${XRL_PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_a -- $B > $O/pmc_${n}_a.log 2>&1
         timeout ${XRL_PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_b -- $B > $O/pmc_${n}_b.log 2>&1
         timeout ${XRL_PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python $R/bench.py --steps 10 --warmup 6 --no-cpu-baseline --no-host-abi --no-stats --no-extra --parity-rows 0 $args > $O/ktrace_$n.log 2>&1
         python - $O $n "$args" <<'PY'
import csv, glob, os, sys, collections
O, n, args = sys.argv[1], sys.argv[2], sys.argv[3].split()
rows = []
for d in ("pmc_a", "pmc_b"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "xrl::" in r["Kernel_Name"] or "step_marker" in r["Kernel_Name"]]
    os.system(f"rm -rf {O}/{d}")
split = {"fetch": ("FETCH_SIZE",), "write": ("WRITE_SIZE",), "l2": ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum")}
for name, ctrs in list(split.items()) + [("sq", None)]:
    known = sum(split.values(), ())
    sel = [r for r in rows if (r["Counter_Name"] in ctrs if ctrs else r["Counter_Name"] not in known)]
    with open(f"{O}/pmc_{n}_{name}.csv", "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
        w.writeheader(); w.writerows(sel)
    os.system(f"cp {O}/pmc_{n}_{name}.csv {O}/pmc_{name}.csv")
agg = collections.defaultdict(list)
for r in rows: agg[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(k, "n=%d mean=%.5g" % (len(v), sum(v) / len(v)))
for f in glob.glob(f"{O}/ktrace/**/*kernel_stats.csv", recursive=True):
    os.system(f"cp {f} {O}/kernel_stats_{n}.csv"); print(open(f).read()[:2500])
os.system(f"rm -rf {O}/ktrace")
cfg, scale, opts = "amazon-670k", "1.0", []
for i, a in enumerate(args):
    if a == "--config": cfg = args[i + 1]
    if a == "--scale": scale = args[i + 1]
    if a == "--opt": opts.append(args[i + 1])
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.system(f"cd {R} && python scripts/pmc_traffic.py {O} {O}/pmc_entry_{n}.json 1.0 {cfg} {scale} {','.join(opts)}")
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
