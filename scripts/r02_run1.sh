#!/bin/bash
# round 2, GPU run 1: parity of the new dense-format kernel K1Q + ballot K2, then A/B bench on Amazon-670K
set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02/pytest_gpu_run1.log
cat gpurun_out/r02/pytest_gpu_run1.log
for v in "" "--opt dense_layers=0" "--opt dense_layers=0 --opt k2_legacy=1" "--opt k2_legacy=1"; do
  tag=$(echo "$v" | tr -d ' -' | tr '=' '_'); tag=${tag:-default}
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline $v > gpurun_out/r02/bench_amazon_$tag.json 2> gpurun_out/r02/bench_amazon_$tag.err
  tail -3 gpurun_out/r02/bench_amazon_$tag.err; cut -c1-400 gpurun_out/r02/bench_amazon_$tag.json
done
