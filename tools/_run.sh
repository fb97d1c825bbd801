mkdir -p gpurun_out/r04am
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_corun_determinism_gpu.py tests/test_engine_gpu.py -q -x -k "xpool_sims or fused_xpool" > gpurun_out/r04am/pytest.txt 2>&1
tail -4 gpurun_out/r04am/pytest.txt
timeout 600 python tools/xpool_sims_bench.py 53000 4000 96 > gpurun_out/r04am/bench_full.txt 2>&1; cat gpurun_out/r04am/bench_full.txt
timeout 300 python tools/xpool_sims_stamps.py 96 > gpurun_out/r04am/stamps_96.txt 2>&1; cat gpurun_out/r04am/stamps_96.txt
timeout 300 python tools/xpool_sims_stamps.py > gpurun_out/r04am/stamps_ragged.txt 2>&1; cat gpurun_out/r04am/stamps_ragged.txt
