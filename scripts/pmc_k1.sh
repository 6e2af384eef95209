#!/bin/bash
# PMC passes over the K1 kernels of one bench step (each counter set in its own run, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/pmc_k1; mkdir -p $OUT
python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS > /dev/null 2>&1   # builds the model cache
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TAGRAM0_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/p$i.log 2>&1
done
python - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k1" not in k and "k2" not in k: continue
        k = k.split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        print(k, {c: "%.4g (n=%d)" % (v, cnt[(k, c)]) for c, v in agg[k].items()})
PY
