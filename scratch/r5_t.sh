#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -8
