# Runtime knobs of the HIP / ROCr layer against the eager training step (tools/train_graph_probe.py prints eager + captured-graph ms/step)
for kv in "BASE=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HIP_FORCE_DEV_KERNARG=0" "AMD_SERIALIZE_COPY=0" "HSA_ENABLE_SDMA=0"; do
  echo "== $kv"; env $kv timeout 100 python tools/train_graph_probe.py 2>&1 | tail -1
done
