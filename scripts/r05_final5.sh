cd $GRAFT_REPO_ROOT
A="--config,dense-768,--scale,0.25,--rows,250000"
XRL_BENCH_TIMEOUT=900 bash scripts/gpu_round.sh r05k bench:--steps,20 bench:--config,amazon-670k-hard,--steps,20 bench:--config,eurlex-4k,--steps,50 bench:--config,wiki10-31k,--steps,50 bench:$A,--steps,30 2>&1 | grep -E "value|== bench|extra|Error|error" | cut -c1-300 | tail -30
