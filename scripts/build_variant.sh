#!/bin/bash
# scripts/build_variant.sh <name> <extra compiler flags...>: builds pecos_amd/lib/libxrl_amd_<name>.so from a copy of pecos_amd/csrc (kernel-tuning
# variants for A/B runs on the GPU box: PECOS_XRL_AMD_SO selects the library)
set -e
R=$(cd $(dirname $0)/.. && pwd); N=$1; shift
D=/tmp/kv_$N; rm -rf $D; mkdir -p $D/pecos_amd/lib $D/include
cp -r $R/pecos_amd/csrc $D/pecos_amd/; cp $R/include/*.h $D/include/; rm -rf $D/pecos_amd/csrc/build $D/pecos_amd/csrc/build_asan
make -C $D/pecos_amd/csrc -j${JOBS:-8} CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fvisibility-inlines-hidden -Wall -Wno-unused-function -pthread $*" 2>&1 | grep -E "error|Error" || true
cp $D/pecos_amd/lib/libxrl_amd.so $R/pecos_amd/lib/libxrl_amd_$N.so; ls -la $R/pecos_amd/lib/libxrl_amd_$N.so
