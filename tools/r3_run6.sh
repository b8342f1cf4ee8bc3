cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -3
timeout -k 10 400 python tools/leak_check.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout -k 10 400 python tools/soak.py 0 2>&1 | grep -v amdgpu.ids | tail -3
