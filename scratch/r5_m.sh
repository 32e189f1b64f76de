#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "--- new"; timeout 300 python -m pytest "tests/test_gpu_vs_reference_model.py::test_train_step_stats_and_derivatives_vs_reference" -x -q 2>&1 | tail -3
echo "--- old encoder"; HUGS_LIB_PATH=$PWD/scratch/libencold.so timeout 300 python -m pytest "tests/test_gpu_vs_reference_model.py::test_train_step_stats_and_derivatives_vs_reference" -x -q 2>&1 | tail -3
done
