#!/bin/bash
cd $GRAFT_REPO_ROOT && timeout -k 10 300 python3 tools/dbg/two_streams.py 2>&1 | tail -6
