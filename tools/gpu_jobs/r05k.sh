#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
