#!/bin/bash
export GL_DEV_SWITCHES=1   # the library reads its developer switches (GL_GEMM_*, GL_ATTN_V2, ...) only with this set
# round 2, evidence run on the shipped tree: g1 (suite + bench lines of every BASELINE config) then g2 (kbench tables, in-situ,
# rocprofv3 kernel stats, PMC traffic per symbol / per problem)
bash tools/gpu_r2g1.sh
bash tools/gpu_r2g2.sh
