#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02j; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "dense or fuzz or scaled or edge" 2>&1 | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 10 --warmup 2 --no-cpu-baseline --no-host-abi > $O/bench_dense768.json 2> $O/bench_dense768.err; tail -1 $O/bench_dense768.err; cut -c1-200 $O/bench_dense768.json
