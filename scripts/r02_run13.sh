#!/bin/bash
# leaf K1 on tile-sorted items (sort_min_tiles) with the current kernels, Amazon-670K and Wiki10-31K
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_run13; rm -rf $O; mkdir -p $O
cd $R
for c in amazon-670k wiki10-31k; do for v in 0 1; do
  timeout 150 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-stats --opt sort_min_tiles=$v > $O/b_${c}_$v.json 2> $O/b_${c}_$v.err
  echo "$c sort_min_tiles=$v: $(grep per-launch $O/b_${c}_$v.err) $(python -c "import json; j=json.loads(open('$O/b_${c}_$v.json').read().splitlines()[-1]); print(j['ms_per_step'])")"
done; done 2>&1 | tee $O/summary.txt
