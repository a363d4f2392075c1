#!/bin/bash
# developer A/B: a second copy of the library + kbench with gemm.hip compiled under extra flags (or from another source tree),
# into gligen_amd/build/var_NAME/ (git-ignored, travels to the GPU box):   tools/build_variant.sh NAME [SRC_TREE] [-D...]
set -e
name=$1; shift
tree=$PWD
if [ -n "$1" ] && [ -d "$1" ]; then tree=$1; shift; fi
out=gligen_amd/build/var_$name
mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I $tree/include"
if [ "$tree" = "$PWD" ]; then
  python -m gligen_amd.build > /dev/null
  hipcc $F "$@" -c gligen_amd/csrc/gemm.hip -o $out/gemm.hip.o
  objs="$out/gemm.hip.o $(ls gligen_amd/build/*.hip.o | grep -v /gemm.hip.o)"
else
  objs=""
  for s in gemm attention norm misc convnext engine capi; do
    [ -f $tree/gligen_amd/csrc/$s.hip ] || continue
    x=""; [ $s = attention ] && x="-mllvm -amdgpu-mfma-vgpr-form"
    hipcc $F $x "$@" -c $tree/gligen_amd/csrc/$s.hip -o $out/$s.hip.o &
    objs="$objs $out/$s.hip.o"
  done
  wait
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgligen_amd.so $objs
hipcc $F $tree/gligen_amd/csrc/kbench.hip -o $out/kbench -L $out -lgligen_amd '-Wl,-rpath,$ORIGIN'
ls -la $out/kbench $out/libgligen_amd.so
