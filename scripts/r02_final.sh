#!/bin/bash
# Round-2 measurement on the GPU box (one gpurun call): parity tests, smoke, bench lines of every BASELINE config,
# PMC passes (fetch / write / L2 requests / VALU) and kernel-trace stats over the timed kernels of the Amazon-670K bench.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_final; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_amazon670k_n1.json 2> $O/bench_amazon670k_n1.err; tail -5 $O/bench_amazon670k_n1.err
for c in eurlex-4k wiki10-31k; do timeout 600 python bench.py --config $c > $O/bench_${c}_n1.json 2> $O/bench_${c}_n1.err; tail -4 $O/bench_${c}_n1.err; done
timeout 1500 python bench.py --config dense-768 --scale ${DENSE_SCALE:-0.25} --rows ${DENSE_ROWS:-250000} --steps 10 --warmup 2 --cpu-seconds 20 > $O/bench_dense768_n1.json 2> $O/bench_dense768_n1.err; tail -5 $O/bench_dense768_n1.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- $B > $O/pmc_l2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-abi --no-stats > $O/ktrace.log 2>&1
python - $O <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "xrl::" in r["Kernel_Name"]]
        with open(f"{O}/{d}.csv", "w", newline="") as out:
            w = csv.DictWriter(out, fieldnames=["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
            w.writeheader(); w.writerows(rows)
    os.system(f"rm -rf {O}/{d}")
for f in glob.glob(f"{O}/ktrace/**/*kernel_stats.csv", recursive=True): os.system(f"cp {f} {O}/kernel_stats.csv")
os.system(f"rm -rf {O}/ktrace")
PY
ls -la $O
