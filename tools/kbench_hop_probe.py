#!/usr/bin/env python3
"""DEV TOOL: large launches at hop = N (no overlap), N/2 and N/4: does re-reading the overlapped samples from L2 cost HBM rate?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import run
for hop in (1024, 512, 256):
    r = run(1024, hop, 65536, 1, 60)
    r["frames_per_us"] = round(r["frames"] / r["us"], 1)
    print(json.dumps(r), flush=True)
