#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c40; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bench_cli.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300
