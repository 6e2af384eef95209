#!/bin/bash
# K1G tile-shape / unroll experiment (dense-768 at a tenth of the label count, N = 100k): correctness of every shape, then timings.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_k1g; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "dense or golden or fuzz or synthetic" 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
B="python bench.py --config dense-768 --scale 0.1 --rows 100000 --steps 10 --warmup 2 --no-cpu-baseline --no-host-abi"
for lib in default u1 u4 u16; do
  for v in 0 1 2 3; do
    if [ $lib = default ]; then unset PECOS_XRL_AMD_SO; else export PECOS_XRL_AMD_SO=$R/pecos_amd/lib/variants/libxrl_amd_k1g_$lib.so; fi
    timeout 300 $B --opt k1g_variant=$v > $O/b_${lib}_v$v.json 2> $O/b_${lib}_v$v.err
    echo "$lib v$v: $(grep per-launch $O/b_${lib}_v$v.err | sed 's/k0_prolongate\[[0-9]\]=[0-9.]*//g; s/k1_sort_items\[[0-9]\]=[0-9.]*//g; s/k2_topk\[[0-9]\]=[0-9.]*//g' | tr -s ' ') $(python -c "import json,sys; print(json.loads(open('$O/b_${lib}_v$v.json').read().splitlines()[-1])['ms_per_step'])" 2>/dev/null)"
  done
done 2>&1 | tee $O/summary.txt
