#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5_suite.txt
cat gpurun_out/r5_suite.txt
