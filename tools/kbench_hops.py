import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import run
for hop in (1024, 512, 256):
    r = run(1024, hop, 262144 * 512 // hop // 2, 1, 30)
    r["unique_in_plus_out_GBs"] = r["GBs"]
    r["L1_load_plus_store_GBs"] = round((4096 + 2052) * r["frames"] / r["us"] / 1e3, 1)
    print(json.dumps(r), flush=True)
