#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_landmark_assign.py -q -x 2>&1 | tail -3
bash ab_tmp.sh 2>&1 | tail -4
