#!/bin/bash
timeout 1700 python -m pytest tests/test_gpu_fanogan.py tests/test_gpu_zimmerer.py tests/test_gpu_caae_chen.py tests/test_gpu_gmvae_you.py tests/test_gpu_scale_parity.py tests/test_gpu_dp_rehearsal.py -q -s 2>&1 | grep -v "^$" | tail -60
