#!/bin/bash
bash $GRAFT_REPO_ROOT/tools/pmc_dense.sh r05pmcdense 2>&1 | cut -c1-220
