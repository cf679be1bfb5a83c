#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c11; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^c[235] |passed|failed|Error|assert" $O/pytest.log | cut -c1-900 | tail -30
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 3000 $O/bench_c2.json
