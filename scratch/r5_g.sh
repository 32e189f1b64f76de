#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests/test_gpu_nerfacto.py tests/test_gpu_nerfacto_fp16.py tests/test_gpu_nerfacto_reference.py tests/test_gpu_nerfacto_encodings.py -x -q 2>&1 | tail -8 | tee gpurun_out/r5g/pytest.txt
