#!/bin/bash
# End-of-round measurement on the GPU box (one gpurun call): parity tests, smoke, PMC traffic, kernel stats, bench.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python $R/scripts/pmc_traffic.py $F $W $R/profiles/pmc_traffic.json && cp $R/profiles/pmc_traffic.json $O/
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/ktrace.log 2>&1
cd $R
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_amazon670k_n1.json 2> $O/bench_amazon670k_n1.err; python scripts/brief.py $O/bench_amazon670k_n1.json | cut -c1-600
for c in eurlex-4k wiki10-31k; do timeout 600 python bench.py --config $c --steps 10 --warmup 2 > $O/bench_${c}_n1.json 2> $O/bench_${c}_n1.err; python scripts/brief.py $O/bench_${c}_n1.json | head -1 | cut -c1-200; done
# keep the merged output small: per-launch PMC rows of the K1/K2 kernels only, kernel stats as they are
python - $O <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(f"{O}/{d}/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "xrl::" in r["Kernel_Name"]]
        if rows:
            with open(f"{O}/{d}.csv", "w", newline="") as out:
                w = csv.DictWriter(out, fieldnames=["Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
                w.writeheader(); w.writerows(rows)
    os.system(f"rm -rf {O}/{d}")
for f in glob.glob(f"{O}/ktrace/*/*kernel_stats.csv"): os.system(f"cp {f} {O}/kernel_stats.csv")
os.system(f"rm -rf {O}/ktrace")
PY
ls -la $O
