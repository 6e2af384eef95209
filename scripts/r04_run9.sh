cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1
timeout 900 python -m pytest tests -m gpu -q -x -k "goldens or fuzz or scaled or guard or edge" 2>&1 | tail -5
for pr in 1 0; do
  for cfg in amazon-670k amazon-670k-hard; do
  echo "== XRL_PRESENCE=$pr $cfg"; XRL_PRESENCE=$pr python bench.py --config $cfg --steps 20 --no-cpu-baseline --no-host-abi --no-stats --parity-rows 4096 2>&1 | grep -E "per-launch|timed output" | cut -c1-260
  done
done
