#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c08; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
python tools/accuracy_report.py > $O/accuracy.jsonl 2> $O/accuracy.err; cat $O/accuracy.jsonl
JSG_4096_PLAN=2 tools/pmc_ab.sh r03_c5b c5 jadespectrogram_amd/libjsg.so "1 2 3" > $O/pmc_c5b.log 2>&1; tail -60 $O/pmc_c5b.log
