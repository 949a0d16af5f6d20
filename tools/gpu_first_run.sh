#!/bin/bash
# First contact with the GPU: build check, parity tests with full output.
mkdir -p gpurun_out
python -c "import torch;print(torch.cuda.get_device_name(0))" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40
