timeout 400 python -m pytest tests/test_refine_driver.py -m gpu -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_refine -- python3 tools/bench_refine.py --batches 4 > gpurun_out/prof_refine.log 2>&1; tail -c 300 gpurun_out/prof_refine.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -- python3 tools/bench_pt_single.py --steps 20 > gpurun_out/prof_cfg5.log 2>&1; tail -c 300 gpurun_out/prof_cfg5.log
