#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_plugin.py -m gpu -q -x -k "all_matched" 2>&1 | grep -v Warn | tail -40 | cut -c1-220
