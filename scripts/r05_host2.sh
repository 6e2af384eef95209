cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q; mkdir -p $O
XRL_HOST_TIMING=0 timeout 200 python scripts/host_abi_probe.py --calls 6 --reuse-alloc > /dev/null 2>&1   # warm the box
for mb in 12 6 8 16 24; do
  echo "== host_batch_mb=$mb"; XRL_HOST_TIMING=0 timeout 200 python scripts/host_abi_probe.py --calls 10 --reuse-alloc --opt host_batch_mb=$mb 2>/dev/null | tail -1 | cut -c1-260
done
echo "== stage timing, default"; XRL_HOST_TIMING=1 timeout 200 python scripts/host_abi_probe.py --calls 5 --reuse-alloc 2>&1 | grep "xrl host" | tail -2 | cut -c1-400
