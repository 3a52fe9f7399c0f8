#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_ocr_det.py -m gpu -q -rA 2>&1 | grep -E "max abs err|err vs fp64|^seed|passed|failed|Error|assert " | head -60
