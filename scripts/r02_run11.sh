#!/bin/bash
# presence bits of the dense row format: parity (fuzz, goldens, full-size Eurlex / Wiki10), then Amazon-670K / Eurlex-4K / Wiki10-31K
# with k1q_pres on / off
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_run11; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.log
grep -q failed $O/pytest.log && exit 1
for c in amazon-670k eurlex-4k wiki10-31k; do for p in 1 0; do
  timeout 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-stats --opt k1q_pres=$p > $O/b_${c}_p$p.json 2> $O/b_${c}_p$p.err
  echo "$c k1q_pres=$p: $(grep per-launch $O/b_${c}_p$p.err) $(python -c "import json; print(json.loads(open('$O/b_${c}_p$p.json').read().splitlines()[-1])['ms_per_step'])")"
done; done 2>&1 | tee $O/summary.txt
