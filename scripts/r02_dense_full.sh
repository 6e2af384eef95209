#!/bin/bash
# BASELINE.json configs[4] at FULL size: N = 1 M dense queries x 768, L = 3 M labels (tree [8, 128, 2048, 32768, 3 M]); after the
# GPU test suite on the final code.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_dense_full; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/pytest_gpu.log
timeout 840 python bench.py --config dense-768 --scale 1.0 --steps 5 --warmup 1 --host-steps 2 --cpu-seconds 10 > $O/bench_dense768_full.json 2> $O/bench_dense768_full.err
tail -6 $O/bench_dense768_full.err | cut -c1-600; tail -c 1500 $O/bench_dense768_full.json
