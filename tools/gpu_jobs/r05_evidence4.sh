#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r05_evidence.sh 2>&1 | tail -30
