timeout -k 10 300 python -u -m pytest tests/test_gpu_model.py -q -x -s -k "cls_token_only or folded_and" --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -12
