cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1
XRL_HOST_TIMING=0 timeout 200 python scripts/host_abi_probe.py --calls 6 --reuse-alloc > /dev/null 2>&1   # warm the box
for t in 1 0; do for mb in 12 16 8; do
  echo "== taper=$t host_batch_mb=$mb"; XRL_HOST_TAPER=$t XRL_HOST_TIMING=0 timeout 200 python scripts/host_abi_probe.py --calls 10 --reuse-alloc --opt host_batch_mb=$mb 2>/dev/null | tail -1 | cut -c1-230
done; done
echo "== hard"; for t in 1 0; do XRL_HOST_TAPER=$t XRL_HOST_TIMING=0 timeout 300 python scripts/host_abi_probe.py --config amazon-670k-hard --calls 8 --reuse-alloc 2>/dev/null | tail -1 | cut -c1-230; done
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or multi_device or reference_binding or dense_input" 2>&1 | tail -3
