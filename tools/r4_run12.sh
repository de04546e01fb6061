timeout 600 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "rc=$?"; tail -c 600 gpurun_out/r4_bench_default.json
timeout 600 python -m pytest "tests/test_hip_parity.py::test_full_size_1080p_determinism_and_kernel_agreement" -m gpu -q 2>&1 | tail -3
