cd $GRAFT_REPO_ROOT
for cfg in amazon-670k amazon-670k-hard wiki10-31k; do
  XRL_HOST_TIMING=0 timeout 400 python scripts/host_abi_probe.py --config $cfg --calls 40 --reuse-alloc --check 2>/dev/null | tail -1 | cut -c1-400
done
