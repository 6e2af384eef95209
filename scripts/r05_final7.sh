cd $GRAFT_REPO_ROOT
bash scripts/gpu_round.sh r05t tests smoke bench:--steps,20 2>&1 | grep -E "passed|failed|smoke|value|extra|Error|error" | cut -c1-400 | tail -12
