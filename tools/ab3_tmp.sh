#!/bin/bash
python -m pytest tests/test_gpu_engine.py -q -x -s -k "route_options" 2>&1 | grep -v "^$" | grep -B2 -A14 "Error\|\[xattn_waves\|\[exact_skip" | head -80
