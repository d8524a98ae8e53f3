python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -2
python bench_configs.py --only c3 2>&1 | grep -vE "Warn|warn|amdgpu.ids|sparse_csr_tensor"
