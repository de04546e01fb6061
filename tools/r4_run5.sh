python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest3.log 2>&1; tail -4 gpurun_out/r4_gputest3.log
tools/pmc_profile.sh r4 > gpurun_out/r4_pmc.log 2>&1; tail -3 gpurun_out/r4_pmc.log
tools/pmc_profile.sh r4_lobe0 --steps 4 --warmup 1 --cpu-seconds 0 --no-roofline --no-extras --parity-pixels 0 --lobes 0 > gpurun_out/r4_pmc_lobe0.log 2>&1; tail -3 gpurun_out/r4_pmc_lobe0.log
