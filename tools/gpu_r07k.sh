#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07k "py:tools/fuzz_masks.py 1000 3000"
