#!/bin/bash
# Build kernel-tuning variants of libxrl_amd.so: scripts/build_variants.sh "P NB" "P NB" ...
# -> pecos_amd/lib/variants/libxrl_amd_P<P>_NB<NB>.so  (select one with PECOS_XRL_AMD_SO)
set -e
cd "$(dirname "$0")/../pecos_amd/csrc"
mkdir -p ../lib/variants build
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -pthread"
for v in "$@"; do
  set -- $v; P=$1; NB=$2; EXTRA="${@:3}"
  tag="P${P}_NB${NB}$(echo "$EXTRA" | tr -d ' =-' )"
  /opt/rocm/bin/hipcc $FLAGS -DXRL_K1_P=$P -DXRL_K1_NB=$NB $EXTRA -c xrl_kernels.hip -o build/k_$tag.o &
done
wait
for f in build/k_*.o; do
  tag=${f#build/k_}; tag=${tag%.o}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/variants/libxrl_amd_$tag.so build/xrl_io.o build/xrl_model.o build/xrl_predict.o build/xrl_select.o build/xrl_mmap.o build/xrl_abi.o $f -pthread
done
ls -la ../lib/variants
