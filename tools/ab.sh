python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -12
