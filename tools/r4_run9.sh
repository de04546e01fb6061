timeout 1300 python -m pytest tests -m gpu -q -x > gpurun_out/r4_gputest8.log 2>&1; tail -6 gpurun_out/r4_gputest8.log
