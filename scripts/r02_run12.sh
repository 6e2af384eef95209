#!/bin/bash
# K1Q: occupancy target of the fused kernel (amdgpu_waves_per_eu 7 / 8 vs the compiler's own 6) on Amazon-670K, Eurlex-4K
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_run12; mkdir -p $O
cd $R
for c in amazon-670k eurlex-4k; do for v in default wpe7 wpe8; do
  if [ $v = default ]; then unset PECOS_XRL_AMD_SO; else export PECOS_XRL_AMD_SO=$R/pecos_amd/lib/variants/libxrl_amd_k1q_$v.so; fi
  timeout 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-stats > $O/c_${c}_$v.json 2> $O/c_${c}_$v.err
  echo "$c $v: $(grep per-launch $O/c_${c}_$v.err) $(python -c "import json; j=json.loads(open('$O/c_${c}_$v.json').read().splitlines()[-1]); print(j['ms_per_step'])")"
done; done 2>&1 | tee $O/summary2.txt
