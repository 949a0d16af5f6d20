#!/bin/bash
# GPU parity suite with hard caps (a hung kernel cannot be interrupted by pytest itself):
# every test file runs under its own `timeout`; the whole script stays well under 10 minutes.
mkdir -p gpurun_out
python -c "import torch;print(torch.cuda.get_device_name(0))" 2>&1 | tail -1
for f in tests/test_gpu_parity.py tests/test_huffmanfile.py; do
  timeout -k 5 240 python -m pytest $f -m gpu -x -q 2>&1 | tail -15
done
