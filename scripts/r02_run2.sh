#!/bin/bash
# round 2, GPU run 2: parity (N2 / K3 / CSC type vs the reference entry points), bench with host-ABI + CPU median,
# FETCH_SIZE calibration, PMC passes over the new kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_amazon.json 2> $O/bench_amazon.err; tail -6 $O/bench_amazon.err; cut -c1-300 $O/bench_amazon.json
cd /tmp && export TMPDIR=/tmp
C=$R/scripts/bin/calib_fetch
$C > $O/calib_plain.txt 2>&1; cat $O/calib_plain.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/calib$i -- $C > $O/calib$i.log 2>&1
done
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- $B > $O/pmc_l2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi --no-stats > $O/ktrace.log 2>&1
python - $O <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
for d in sorted(glob.glob(f"{O}/calib[0-9]")) + [f"{O}/pmc_fetch", f"{O}/pmc_write", f"{O}/pmc_l2", f"{O}/pmc_sq"]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "xrl::" in r["Kernel_Name"] or "calib" in d]
        with open(f"{d}.csv", "w", newline="") as out:
            w = csv.DictWriter(out, fieldnames=["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
            w.writeheader(); w.writerows(rows)
        if "calib" in d:
            agg = collections.defaultdict(list)
            for r in rows: agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for k, v in sorted(agg.items()): print(os.path.basename(d), k, ["%.4g" % x for x in v])
    os.system(f"rm -rf {d}")
for f in glob.glob(f"{O}/ktrace/**/*kernel_stats.csv", recursive=True): os.system(f"cp {f} {O}/kernel_stats.csv")
os.system(f"rm -rf {O}/ktrace")
PY
ls -la $O
