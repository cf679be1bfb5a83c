#!/usr/bin/env python3
"""DEV PROBE: the card's sustained packed-f32 instruction rate at its power cap (tools/probes/power_roof.hip), one JSON line per stream.

    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/power_roof.hip -o tools/variants/libpower_roof.so     (here, CPU only)
    gpurun -- 'python tools/power_roof.py > gpurun_out/power_roof.jsonl'

PR_SECONDS (default 3) per stream, PR_WAVES (default "2,3,8") waves per SIMD."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hwmon  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "v_pk_fma_f32 only", 1: "v_pk_add_f32 only", 2: "add/mul/fma mix of the FFT kernels",
         3: "that mix + one LDS instruction per six vector instructions",
         4: "... + one streamed 16-byte load per lane per 56 vector instructions (C3's bytes per instruction)"}


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "variants", "libpower_roof.so"))
    lib.power_roof_run.restype = ctypes.c_double
    lib.power_roof_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    cus = int(os.environ.get("PR_CUS", "256"))   # MI355X
    hw = hwmon.Hwmon(0)
    seconds = float(os.environ.get("PR_SECONDS", "3"))
    for waves in [int(x) for x in os.environ.get("PR_WAVES", "2,3,8").split(",")]:
        for mode in (0, 1, 2, 3, 4):
            with hwmon.Watch(hw, settle_s=1.0) as w:
                rate = lib.power_roof_run(mode, waves, seconds, cus)
            s = w.summary() if hw.ok else {}
            mhz = s.get("sclk_MHz_median")
            line = {"stream": NAMES[mode], "waves_per_simd": waves, "vector_instructions_per_s": rate, "cus": cus, **s}
            if mhz:
                line["issue_cycles_per_instruction_per_simd"] = mhz * 1e6 * cus * 4 / rate
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
