#!/bin/bash
# K1Q: weight rows in flight per wavefront (U for layers of <= 3 candidate registers / of 1) on Amazon-670K
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_run12; rm -rf $O; mkdir -p $O
cd $R
for v in default u6_16 u10_16 u12_16 u16_16 u8_24; do
  if [ $v = default ]; then unset PECOS_XRL_AMD_SO; else export PECOS_XRL_AMD_SO=$R/pecos_amd/lib/variants/libxrl_amd_k1q_$v.so; fi
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-stats > $O/b_$v.json 2> $O/b_$v.err
  echo "$v: $(grep per-launch $O/b_$v.err) $(python -c "import json; j=json.loads(open('$O/b_$v.json').read().splitlines()[-1]); print(j['ms_per_step'])")"
done 2>&1 | tee $O/summary.txt
