python -m pytest tests/test_gpu_reference_style.py -m gpu -x -q --timeout 600 2>&1 | tail -12
