timeout 300 python tools/bench_pt_single.py > gpurun_out/r4_cfg5_ngp.json 2>gpurun_out/r4_cfg5_ngp.err; tail -c 700 gpurun_out/r4_cfg5_ngp.json
timeout 300 python tools/bench_pt_single.py --material stub > gpurun_out/r4_cfg5_stub.json 2>/dev/null; tail -c 400 gpurun_out/r4_cfg5_stub.json
timeout 300 python tools/bench_pt_single.py --spp 128 --calls 1 > gpurun_out/r4_cfg5_ngp_1call.json 2>/dev/null; tail -c 400 gpurun_out/r4_cfg5_ngp_1call.json
timeout 400 python tools/bench_refine.py > gpurun_out/r4_refine_ngp.json 2>gpurun_out/r4_refine_ngp.err; tail -c 900 gpurun_out/r4_refine_ngp.json
timeout 400 python tools/bench_refine.py --batch-pixels 163840 --batches 4 > gpurun_out/r4_refine_ngp_big.json 2>/dev/null; tail -c 500 gpurun_out/r4_refine_ngp_big.json
