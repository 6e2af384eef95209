cd $GRAFT_REPO_ROOT
bash scripts/gpu_round.sh r05f tests smoke bench:--steps,20 2>&1 | tail -40
