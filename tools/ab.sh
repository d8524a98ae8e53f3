python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
