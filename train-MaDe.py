#!/usr/bin/env python3
"""Training / validation entry point (same flags as the reference's train-MaDe.py; logic in mgsv_amd/driver.py).

    python train-MaDe.py --name run --do_train --do_eval --mml_fusion concat --detr_enc_layers 2 --audio_short_cut 0 \
        --max_v_frames 50 --synthetic_features 1 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train-MaDe.py ...      # one process per GPU
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from mgsv_amd.driver import main_train  # noqa: E402

if __name__ == "__main__":
    main_train()
