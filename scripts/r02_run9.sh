#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02k; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for c in wiki10-31k eurlex-4k; do
timeout 600 python bench.py --config $c --steps 30 --no-cpu-baseline --no-host-abi > $O/bench_$c.json 2> $O/bench_$c.err; tail -1 $O/bench_$c.err; cut -c1-200 $O/bench_$c.json
done
