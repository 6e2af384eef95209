cd $GRAFT_REPO_ROOT
XRL_BENCH_TIMEOUT=900 bash scripts/gpu_round.sh r05n bench:--steps,20 bench:--config,amazon-670k-hard,--steps,20 2>&1 | grep -E "value|== bench|extra|Error|error" | cut -c1-300 | tail -12
