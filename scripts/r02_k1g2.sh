#!/bin/bash
# K1G: XCD-aware workgroup order on/off and register-tile shapes (dense-768, N = 100 k; scale 0.1 -> 96-column parents,
# 0.05 -> 64, 0.15 -> 128), the dense host-ABI path, then PMC passes of the default shapes.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_k1g2; rm -rf $O; mkdir -p $O
cd $R
timeout 240 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "dense or golden or fuzz or synthetic" 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
grep -q passed $O/pytest.log || exit 1
run() { # scale variant grouped extra...
  B="python bench.py --config dense-768 --scale $1 --rows 100000 --steps 10 --warmup 2 --no-cpu-baseline ${@:4}"
  timeout 150 $B --opt k1g_variant=$2 --opt k1g_grouped=$3 > $O/b_s$1_v$2_g$3.json 2> $O/b_s$1_v$2_g$3.err
  echo "scale $1 v$2 grouped=$3: $(grep per-launch $O/b_s$1_v$2_g$3.err | sed 's/k0_prolongate\[[0-9]\]=[0-9.]*//g; s/k1_sort_items\[[0-9]\]=[0-9.]*//g; s/k2_topk\[[0-9]\]=[0-9.]*//g' | tr -s ' ') $(grep 'host ABI' $O/b_s$1_v$2_g$3.err | cut -c1-90) $(python -c "import json,sys; print(json.loads(open('$O/b_s$1_v$2_g$3.json').read().splitlines()[-1])['ms_per_step'])" 2>/dev/null)"
}
{ run 0.1 0 1; run 0.1 0 0 --no-host-abi; run 0.1 4 1 --no-host-abi; run 0.1 5 1 --no-host-abi; run 0.1 3 1 --no-host-abi; } 2>&1 | tee $O/summary.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- $B > $O/pmc_l2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
python - $O <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
for d in ("pmc_fetch", "pmc_l2", "pmc_sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k1g_kernel" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{O}/{d}.txt", "w") as out:
        for k, cs in sorted(agg.items()):
            out.write(f"{k}: " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in cs.items()) + "\n")
    os.system(f"rm -rf {O}/{d}; cat {O}/{d}.txt")
PY
cd $R
{ run 0.05 0 1 --no-host-abi; run 0.05 3 1 --no-host-abi; run 0.05 4 1 --no-host-abi; run 0.15 0 1 --no-host-abi; run 0.15 3 1 --no-host-abi; run 0.15 4 1 --no-host-abi; } 2>&1 | tee -a $O/summary.txt
