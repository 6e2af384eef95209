cd $GRAFT_REPO_ROOT
A="--config,dense-768,--scale,0.25,--rows,250000"
XRL_BENCH_TIMEOUT=900 XRL_PMC_TIMEOUT=400 bash scripts/gpu_round.sh r05i bench:$A,--steps,30 pmc:$A 2>&1 | grep -E "^k1|^k0|^k2|per step|value|bench|cpu reference|host ABI|Error|error" | cut -c1-500 | tail -30
# BASELINE.json configs[4] at its stated size: N = 1 M dense queries, L = 3 M labels (17 GB model)
XRL_BENCH_TIMEOUT=1500 bash scripts/gpu_round.sh r05j bench:--config,dense-768,--steps,10,--warmup,3,--cpu-seconds,8 2>&1 | grep -E "value|bench|cpu reference|host ABI|Error|error" | cut -c1-500 | tail -12
