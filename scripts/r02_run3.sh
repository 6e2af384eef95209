#!/bin/bash
# round 2, GPU run 3: parity (all), benches of the other BASELINE configs, counter list
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for c in eurlex-4k wiki10-31k; do
  timeout 600 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; tail -5 $O/bench_$c.err; cut -c1-250 $O/bench_$c.json
  timeout 300 python bench.py --config $c --opt dense_layers=0 --no-cpu-baseline --no-host-abi > $O/bench_${c}_tile.json 2> $O/bench_${c}_tile.err; tail -2 $O/bench_${c}_tile.err
done
timeout 900 python bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 5 --warmup 1 --cpu-seconds 20 > $O/bench_dense768.json 2> $O/bench_dense768.err; tail -6 $O/bench_dense768.err; cut -c1-250 $O/bench_dense768.json
rocprofv3 -L 2>/dev/null | grep -iE "TCC_EA0|TCC_REQ|TCC_READ|TCP_TCC" | head -60 > $O/counters.txt; head -60 $O/counters.txt
