#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "$1" 2>&1 | tail -25
