#!/bin/bash
# round 2, GPU run 5: fused multi-layer K1Q -- parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02f; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for v in "" "--opt k1q_fuse=0"; do
  tag=$(echo "$v" | tr -d ' -' | tr '=' '_'); tag=${tag:-default}
  timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-abi $v > $O/bench_amazon_$tag.json 2> $O/bench_amazon_$tag.err
  tail -2 $O/bench_amazon_$tag.err; cut -c1-220 $O/bench_amazon_$tag.json
done
for c in eurlex-4k wiki10-31k; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --no-host-abi > $O/bench_$c.json 2> $O/bench_$c.err; tail -1 $O/bench_$c.err; cut -c1-200 $O/bench_$c.json
done
