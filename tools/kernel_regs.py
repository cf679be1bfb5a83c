#!/usr/bin/env python3
"""Per-kernel resource usage of the built STFT units (VGPRs, SGPRs, spills, scratch, LDS) from the code objects' metadata, plus
instruction-mix counts from the disassembly.  CPU only (llvm-objcopy / clang-offload-bundler / llvm-readelf / llvm-objdump).

    python tools/kernel_regs.py [--isa PATTERN] [obj ...]      (default objects: jadespectrogram_amd/build/jsg_stft_{a,b}.o)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "scratch.o")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}"])
    return co


def short(name):
    out = subprocess.check_output(["c++filt", name]).decode().strip()
    m = re.search(r"stft_db_kernel<jsg::Cfg<([^>]*)>, (\d+), (\d+), (\d+), (\d+)>", out)
    if not m:
        return out[:60]
    v = [int(x.replace("(int)", "")) for x in m.group(1).split(", ")]
    tag = f"N{v[0]} {v[1]}x{v[2]}x{v[3]} L{v[4]} WPB{v[9]}" + (" PAIR" if len(v) > 15 and v[15] else "")
    return f"{tag:28s} MIX{m.group(2)} OUT{m.group(3)} STR{m.group(4)} XLOG{m.group(5)}"


def main():
    args = sys.argv[1:]
    isa_pat = None
    if args and args[0] == "--isa":
        isa_pat = args[1]
        args = args[2:]
    objs = args or [os.path.join(ROOT, "jadespectrogram_amd", "build", f) for f in ("jsg_stft_a.o", "jsg_stft_b.o")]
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            co = code_object(obj, tmp)
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co]).decode()
            mix = {}
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    mix[cur] = {}
                    continue
                if cur and line.strip() and not line.startswith("Disassembly"):
                    op = line.split("//")[0].split()
                    if op:
                        mix[cur][op[0]] = mix[cur].get(op[0], 0) + 1
            for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = g("name")
                if "stft_db_kernel" not in name:
                    continue
                mx = mix.get(name, {})
                valu = sum(c for o, c in mx.items() if o.startswith("v_"))
                pk = sum(c for o, c in mx.items() if o.startswith("v_pk_"))
                lds = sum(c for o, c in mx.items() if o.startswith("ds_"))
                vmem = sum(c for o, c in mx.items() if o.startswith(("global_", "buffer_", "scratch_")))
                print(f"{short(name)}  vgpr {g('vgpr_count'):>3} spill {g('vgpr_spill_count'):>2} sgpr {g('sgpr_count'):>3} sspill {g('sgpr_spill_count'):>2} "
                      f"scratch {g('private_segment_fixed_size'):>3} lds {g('group_segment_fixed_size'):>6} | static: valu {valu} (pk {pk}) ds {lds} vmem {vmem} "
                      f"nop {mx.get('s_nop', 0)} mov {mx.get('v_mov_b32_e32', 0) + mx.get('v_mov_b32_e64', 0)} swap {mx.get('v_permlane32_swap_b32_e32', 0)}")
                if isa_pat and re.search(isa_pat, short(name)):
                    print(dis.split(f"<{name}>:")[1].split("\n\n")[0][:2000000])


if __name__ == "__main__":
    main()
