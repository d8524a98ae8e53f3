#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_integration_stub.py tests/test_gpu_lattice.py::test_row_kernels_build_the_same_plans_as_the_tensor_op_builder tests/test_gpu_march.py::test_randomised_stencils_through_the_public_path "tests/test_gpu_round4.py::test_block_diag_and_split_on_device_bit_exact" -q -m gpu 2>&1 | grep -E "^E |Error|assert|took|FAILED|passed|failed" | head -60 > $O/tests.log
cat $O/tests.log
