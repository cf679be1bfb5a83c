#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c36; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
