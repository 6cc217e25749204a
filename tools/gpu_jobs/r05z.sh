#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -s -k "index_exact_route_other_seeds" 2>&1 | grep -E "index parity|passed|failed|Error|assert" | cut -c1-200
