python -m pytest tests/test_gpu_nerfacto.py -q -x -k "fused_field" 2>&1 | tail -15
