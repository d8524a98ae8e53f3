python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4b -o c4 -- python3 $GRAFT_REPO_ROOT/bench_configs.py --only c4 --no-cpu > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; head -7 gpurun_out/prof_c4b/c4_kernel_stats.csv | cut -c1-150
