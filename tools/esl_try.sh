#!/bin/bash
cd $GRAFT_REPO_ROOT
export ONLY=single
echo "== default"; timeout 200 python tools/esl_groups.py 2>&1 | grep -E "single|group ==|rror"
echo "== XM_K2_FLAGS=1"; XM_K2_FLAGS=1 timeout 200 python tools/esl_groups.py 2>&1 | grep -E "single|group ==|rror"
echo "== XM_K2_DIRECT=1"; XM_K2_DIRECT=1 timeout 200 python tools/esl_groups.py 2>&1 | grep -E "single|group ==|rror"
