mkdir -p gpurun_out/r04ag
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "xpool_sims" > gpurun_out/r04ag/pytest.txt 2>&1
tail -3 gpurun_out/r04ag/pytest.txt
timeout 300 python tools/xpool_sims_stamps.py > gpurun_out/r04ag/stamps_ragged.txt 2>&1; cat gpurun_out/r04ag/stamps_ragged.txt
timeout 300 python tools/xpool_sims_stamps.py 96 > gpurun_out/r04ag/stamps_96.txt 2>&1; cat gpurun_out/r04ag/stamps_96.txt
timeout 600 python tools/xpool_sims_bench.py 53000 4000 96 > gpurun_out/r04ag/bench_full.txt 2>&1; cat gpurun_out/r04ag/bench_full.txt
