#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_io.py tests/test_gpu_lama.py -m gpu -x -q -k "resident or overlapping" 2>&1 | tail -30
