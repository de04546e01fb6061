timeout 900 python -m pytest tests/test_ngp.py tests/test_abi.py -m gpu -q -x > gpurun_out/r4_gputest6.log 2>&1; tail -8 gpurun_out/r4_gputest6.log
timeout 600 python -m pytest tests/test_refine_driver.py -m gpu -q -x > gpurun_out/r4_gputest7.log 2>&1; tail -8 gpurun_out/r4_gputest7.log
