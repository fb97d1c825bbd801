R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "xpool_fused" 2>&1 | tail -3
python tools/xpool_stamps.py 96 | sed -n 1,4p; python tools/xpool_stamps.py 96 | sed -n 15,18p
for d in 0; do MADE_XPOOL_DBG=$d timeout 120 python tools/xpool_only.py 8192 256 32 full 2>&1 | tail -1; MADE_XPOOL_DBG=$d timeout 120 python tools/xpool_only.py 8192 256 96 full 2>&1 | tail -1; done
timeout 300 python tools/xpool_only.py 53000 512 2>&1 | tail -1
