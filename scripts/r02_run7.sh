#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02g; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for c in eurlex-4k wiki10-31k amazon-670k; do
  timeout 600 python bench.py --config $c --steps 30 --no-cpu-baseline --no-host-abi > $O/bench_$c.json 2> $O/bench_$c.err; tail -1 $O/bench_$c.err; cut -c1-200 $O/bench_$c.json
done
timeout 900 python bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 10 --warmup 2 --no-cpu-baseline --no-host-abi > $O/bench_dense768.json 2> $O/bench_dense768.err; tail -1 $O/bench_dense768.err; cut -c1-200 $O/bench_dense768.json
