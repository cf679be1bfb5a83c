#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c33; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -q -m "not gpu" -x > $O/pytest_cpu.log 2>&1; echo "cpu pytest rc=$?"; tail -2 $O/pytest_cpu.log
