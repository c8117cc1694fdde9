#!/bin/bash
# the stream-order-sensitive part of the GPU suite under the package's new default GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
OUT=gpurun_out/r4_23; rm -rf $OUT; mkdir -p $OUT
timeout 230 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp_nccl.py tests/test_gpu_dp_rehearsal.py tests/test_gpu_trainers.py tests/test_gpu_scale_parity.py tests/test_gpu_cevae.py \
   "tests/test_gpu_knobs.py::test_any_order_edge_waits_for_the_slower_filter_gradient" "tests/test_gpu_knobs.py::test_bottleneck_sibling_exchange_is_bounded_and_reports" \
   tests/test_gpu_fanogan.py tests/test_gpu_gmvae.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 > $OUT/pytest_subset_q8.log
cat $OUT/pytest_subset_q8.log
python -c "import os, unsupervised_anomaly_detection_brain_mri_amd; print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))"
