cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e
for v in "" _w8 _u16 _u4 _w8u4; do
  echo "== variant '$v'"; PECOS_XRL_AMD_SO=$PWD/pecos_amd/lib/libxrl_amd$v.so python bench.py --steps 20 --no-cpu-baseline --no-host-abi --no-stats --parity-rows 0 2> gpurun_out/r04e/var$v.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['per_kernel_ms_per_step'] if d.get('roofline') else None)"
  grep "per-launch" gpurun_out/r04e/var$v.err | cut -c1-200
done
bash scripts/gpu_round.sh r04e pmc: 2>&1 | grep -E "k1q|k1_kernel" | head -40
