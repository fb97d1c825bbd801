R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "xpool_fused" 2>&1 | tail -2
timeout 300 python tools/xpool_only.py 53000 512 2>&1 | tail -1
timeout 300 python tools/xpool_only.py 53000 512 2>&1 | tail -1
python tools/xpool_stamps.py 96 | sed -n 2,3p; python tools/xpool_stamps.py 96 | sed -n 16,17p
