#!/bin/bash
# round 4 (g): training slice: block forward + backward vs the reference's autograd
export GL_DEV_SWITCHES=1
O=gpurun_out/r4g; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "fuser_block_backward" ) > $O/pytest_train.log 2>&1; tail -25 $O/pytest_train.log | cut -c1-400
