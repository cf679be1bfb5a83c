#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c64; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_host_cpp.py -m gpu -x -q -k "offline" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
/tmp/jsg_offline_render_example 24 1875; /tmp/jsg_offline_render_example 44 1875
