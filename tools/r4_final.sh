timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1300 python -m pytest tests -m gpu -q > gpurun_out/r4_gputest_final.log 2>&1; tail -3 gpurun_out/r4_gputest_final.log
timeout 900 tools/pmc_profile.sh r4 > gpurun_out/r4_pmc.log 2>&1; tail -2 gpurun_out/r4_pmc.log
timeout 600 tools/pmc_profile.sh r4_lobe0 --steps 4 --warmup 1 --cpu-seconds 0 --no-roofline --no-extras --parity-pixels 0 --lobes 0 > gpurun_out/r4_pmc_lobe0.log 2>&1; tail -2 gpurun_out/r4_pmc_lobe0.log
