#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
mkdir -p gpurun_out
K=gligen_amd/build/kbench
cat > /tmp/abl.shapes <<EOS
conv 8 64 64 320 0 320 1 0 7
gemm 32768 320 1280 0 10
gemm 32768 2560 320 1 10
gemm 32768 320 320 0 25
gemm 8192 5120 640 1 10
gemm 2048 10240 1280 1 10
conv 8 32 32 640 0 640 1 0 6
gemm 8192 640 640 0 25
gemm 2048 1280 1280 0 25
EOS
for f in 4,5,1 4,4,1 2,5,1; do
for d in 0 4 7 3; do
echo "== force $f dbg $d"
GL_GEMM_AUTOTUNE=0 KB_FORCE=$f GL_GEMM_DBG=$d timeout 100 $K /tmp/abl.shapes 20 | grep "^conv\|^gemm" | cut -c1-110
done
done > gpurun_out/abl3.txt 2>&1
