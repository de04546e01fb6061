for pm in 12 8 10 14 16; do AB_TAG=_pm$pm AB_ARGS="--parity-pixels 0 --debug-set phase_min=$pm" tools/ab_bench.sh gpurun_out/ab11 8 iris_amd/variants/v_final.so; done
