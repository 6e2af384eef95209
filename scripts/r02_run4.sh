#!/bin/bash
# round 2, GPU run 4: parity with the tiled dense-query SGEMM (K1G), dense-768 bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 900 python bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_dense768_k1g.json 2> $O/bench_dense768_k1g.err; tail -4 $O/bench_dense768_k1g.err; cut -c1-250 $O/bench_dense768_k1g.json
timeout 900 python bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 5 --warmup 1 --no-cpu-baseline --no-host-abi --opt k1g_min_items=0 > $O/bench_dense768_k1q.json 2> $O/bench_dense768_k1q.err; tail -2 $O/bench_dense768_k1q.err
