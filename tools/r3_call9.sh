#!/bin/bash
# round 3, GPU call 9: UniDepthV1 on the DINOv2 ViT-L/14 backbone
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c9 && O=gpurun_out/r3c9
export PYTHONWARNINGS=ignore
timeout 900 python -m pytest tests/test_v1_gpu.py -x -q -m gpu -s -k "vitl14" 2>&1 | grep -v Warning | tail -30 > $O/vitl.txt
tail -30 $O/vitl.txt
