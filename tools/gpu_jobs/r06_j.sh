#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_train.py -q -x -k "expand_stride_0 or fallback_key" 2>&1 | tail -8
