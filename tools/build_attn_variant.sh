#!/bin/bash
# developer build: libgligen_amd.so + kbench with attention.hip under extra flags, into gligen_amd/build/var_NAME/
#   tools/build_attn_variant.sh NAME [-D...]
set -e
name=$1; shift
out=gligen_amd/build/var_$name
mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I include"
python -m gligen_amd.build > /dev/null
hipcc $F -mllvm -amdgpu-mfma-vgpr-form "$@" -c gligen_amd/csrc/attention.hip -o $out/attention.hip.o
objs="$out/attention.hip.o $(ls gligen_amd/build/*.hip.o | grep -v /attention.hip.o)"
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgligen_amd.so $objs
hipcc $F gligen_amd/csrc/kbench.hip -o $out/kbench -L $out -lgligen_amd '-Wl,-rpath,$ORIGIN'
ls -la $out/kbench $out/libgligen_amd.so
