#!/usr/bin/env python3
"""Test-split evaluation entry point (same flags as the reference's test-MaDe.py; logic in mgsv_amd/driver.py).

    python test-MaDe.py --name eval --load_uni_model_path logs/.../best_R1.pth --mml_fusion concat --detr_enc_layers 2 --audio_short_cut 0 ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from mgsv_amd.driver import main_test  # noqa: E402

if __name__ == "__main__":
    main_test()
