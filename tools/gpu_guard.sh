#!/bin/bash
# kernel tests with every device tensor flush against the end of its own allocation (tests/conftest.py, T3D_GUARD):
# out-of-bounds accesses fault.  -v so the last line names the case that was running.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp T3D_GUARD=1 PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1
for f in ${@:-tests/test_gpu_dwconv.py tests/test_gpu_pwconv.py tests/test_gpu_misc_kernels.py tests/test_gpu_stem.py}; do
  echo "== $f"
  timeout 900 python -X faulthandler -m pytest $f -m gpu -v -p no:cacheprovider 2>&1 | grep -E "PASSED|FAILED|ERROR|fault|passed|failed|Error" | tail -4
done
