timeout -k 10 600 python -u -m pytest tests/test_train_step.py tests/test_train_ddp.py tests/test_gpu_fuzz.py -q -x -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -8
