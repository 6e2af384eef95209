#!/bin/bash
# two-stream async predict test; K1Q fusion threshold on Amazon-670K (k1q_fuse = 3: levels 0-3 in one launch; 1: levels 0-1 fused,
# 2 and 3 on their own; 0: one launch per level)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_run10; rm -rf $O; mkdir -p $O
cd $R
free -g | head -2; df -h /tmp | tail -1; nproc
timeout 200 python -m pytest tests/test_gpu_device_inputs.py -m gpu -q -x -k "async or device_resident" 2>&1 | tail -3 | tee $O/pytest.log
for f in 3 1 0; do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-stats --opt k1q_fuse=$f > $O/b_fuse$f.json 2> $O/b_fuse$f.err
  echo "k1q_fuse=$f: $(grep per-launch $O/b_fuse$f.err) $(python -c "import json; print(json.loads(open('$O/b_fuse$f.json').read().splitlines()[-1])['ms_per_step'])")"
done 2>&1 | tee $O/summary.txt
