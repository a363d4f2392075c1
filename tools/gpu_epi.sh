#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
mkdir -p gpurun_out
K=gligen_amd/build/kbench
timeout 300 $K tools/unet_b8.shapes 10 - check > gpurun_out/kb_epi.txt 2>&1
grep "^gemm\|^conv\|TOTAL\|CHECK\|MISMATCH" gpurun_out/kb_epi.txt | cut -c1-120 > gpurun_out/kb_epi_s.txt
GL_GEMM_DBG=8 timeout 300 $K tools/unet_b8.shapes 10 > gpurun_out/kb_epi_old.txt 2>&1
paste <(awk '{printf "%-44s %8s %-12s\n", $1" "$2" "$3" "$4" "$5" "$6" "$7" "$8, $(NF-4), $NF}' gpurun_out/kb_epi_s.txt) <(grep "^gemm\|^conv\|TOTAL" gpurun_out/kb_epi_old.txt | awk '{printf "%8s %-12s\n", $(NF-4), $NF}')
timeout 300 $K tools/vae_b4.shapes 3 - check | tail -3
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -2
