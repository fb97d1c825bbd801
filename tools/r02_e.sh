R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_ops_gpu.py -x -q -k "attention" 2>&1 | tail -3
python tools/attn_bwd_bench.py
