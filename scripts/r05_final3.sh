cd $GRAFT_REPO_ROOT
bash scripts/gpu_round.sh r05h pmc: bench:--config,eurlex-4k,--steps,50 pmc:--config,eurlex-4k bench:--config,wiki10-31k,--steps,50 pmc:--config,wiki10-31k 2>&1 | grep -E "^k1|^k0|^k2|per step|value|bench|cpu reference|host ABI|Error|error" | cut -c1-500 | tail -50
