cd $GRAFT_REPO_ROOT
bash scripts/gpu_round.sh r05g pmc: bench:--config,amazon-670k-hard,--steps,20 pmc:--config,amazon-670k-hard 2>&1 | grep -E "^k1|^k0|^k2|per step|value|bench|cpu reference|host ABI|Error|error" | cut -c1-600 | tail -40
