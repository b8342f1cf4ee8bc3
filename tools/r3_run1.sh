cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t2.log 2>&1
tail -8 gpurun_out/r3_t2.log
BENCH_DEBUG=1 python tools/r3_order2.py 2>&1 | grep -v amdgpu.ids | tail -12
