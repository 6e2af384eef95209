#!/bin/bash
# K1Q with the epilogue state of narrow layers stashed in LDS: fused kernel at 8 (64 VGPRs) vs 7 (68 VGPRs) wavefronts per SIMD
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_run14; rm -rf $O; mkdir -p $O
cd $R
timeout 150 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or golden or synthetic" 2>&1 | tail -3 | tee $O/pytest.log
for v in default w7; do
  if [ $v = default ]; then unset PECOS_XRL_AMD_SO; else export PECOS_XRL_AMD_SO=$R/pecos_amd/lib/variants/libxrl_amd_k1q_$v.so; fi
  timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-abi --no-stats > $O/b_$v.json 2> $O/b_$v.err
  echo "$v: $(grep per-launch $O/b_$v.err) $(python -c "import json; j=json.loads(open('$O/b_$v.json').read().splitlines()[-1]); print(j['ms_per_step'])")"
done 2>&1 | tee $O/summary.txt
