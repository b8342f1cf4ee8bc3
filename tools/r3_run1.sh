cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_schedule.py -x -q -s > gpurun_out/r3_t5.log 2>&1
tail -25 gpurun_out/r3_t5.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "multi or cxx" > gpurun_out/r3_t6.log 2>&1
tail -5 gpurun_out/r3_t6.log
