# the A/B switches must all still produce correct steps: the train-step / backward / DP / graph tests under each
for env in "HUGS_TN_BATCH=0" "HUGS_TN_BATCH=2" "HUGS_STEP_GRAPH=1 HUGS_STEP_GRAPH_LANES=" "HUGS_STEP_GRAPH=0" "HUGS_HEAD_FOLD=0 HUGS_SIDE_LATE=1" "HUGS_TN_BATCH_SYNC=0 HUGS_DW_AFTER_PROP=0"; do
  echo "== $env"
  env $env python -m pytest tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_bench_config.py tests/test_gpu_determinism.py tests/test_gpu_data_parallel.py -q -x 2>&1 | tail -2
done
