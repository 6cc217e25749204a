#!/bin/bash
HEAD=30 tools/prof_cmd.sh r05t/prof_exact python tools/run_engine.py --batch 16 --steps 20
