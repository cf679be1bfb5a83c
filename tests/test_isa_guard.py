"""Build-time guard for the counted wait in stft_db_kernel (csrc/jsg_kernels.hip, "lane tables first").

The kernel issues its lane tables as LDS-DMA pieces (global_load_lds_dwordx4), then the F*P loads of its first frame, and
retires the tables with `s_waitcnt vmcnt(F*P)` + `s_barrier` while the frame is still in flight.  vmcnt counts
instructions in issue order, so the wait is right only if the compiler emits EXACTLY F*P vector-memory instructions
between the last table piece and the wait: were a frame load split in two, the barrier would release waves whose tables
have not landed (silently wrong twiddles); were two merged, the wait would also drain a frame load (slower, still right).
This test disassembles the device code object and checks the count for every instantiation.  CPU only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJS = [os.path.join(ROOT, "jadespectrogram_amd", "build", f) for f in ("jsg_stft_a.o", "jsg_stft_b.o")]   # the units that hold the STFT kernels

VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)")


def _disassemble(tmp_path):
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not found under {LLVM}")
    if not all(os.path.exists(o) for o in OBJS):
        from jadespectrogram_amd import _build
        _build.build_lib()
    text = ""
    for k, obj in enumerate(OBJS):
        fat, co = str(tmp_path / f"fat{k}.bin"), str(tmp_path / f"dev{k}.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, str(tmp_path / "scratch.o")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
        text += subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co]).decode() + "\n"
    return text


def _functions(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur and line.strip() and not line.startswith("Disassembly"):
            body.append(line.split("//")[0].strip())
    if cur:
        yield cur, body


def test_counted_wait_matches_the_emitted_frame_loads(tmp_path):
    text = _disassemble(tmp_path)
    checked = 0
    for name, body in _functions(text):
        if "stft_db_kernel" not in name and "stft_image_kernel" not in name:
            continue
        # Cfg<N, R1, R2, R3, L, S1, AX, AY, AZ, WPB, TLOC, WPS, FPW, EARLY1, TWF, PAIR, RTAB>
        m = re.search(r"3CfgI((?:Li\d+E)+)E", name)
        vals = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        n, lanes, tloc, fpw = vals[0], vals[4], vals[10], vals[12]
        pair = len(vals) > 15 and vals[15] == 1
        if tloc != 1:
            continue
        expect = (n // 2 // lanes) * fpw
        if pair:
            # the pair plan loads 2 P = 64 dwords (one per channel and value); the counter names 63 at most, so its wait also retires the
            # oldest frame load: 64 loads between the last table piece and vmcnt(63)
            loads, expect = 2 * expect, 63
        dma = [i for i, ins in enumerate(body) if ins.startswith("global_load_lds_dwordx4") or (ins.startswith("buffer_load_dwordx4") and " lds" in ins)]
        assert dma, f"{name}: no LDS-DMA table pieces found"
        bar = next(i for i, ins in enumerate(body) if ins.startswith("s_barrier"))
        assert dma[-1] < bar or all(d < bar for d in dma), f"{name}: table pieces after the barrier"
        last_dma = max(d for d in dma if d < bar)
        # the wait is the instruction in front of the barrier (one asm statement: nothing can be scheduled between them)
        w = body[bar - 1]
        mm = re.match(r"s_waitcnt vmcnt\((\d+)\)", w)
        assert mm, f"{name}: expected the counted wait in front of the first s_barrier, found '{w}'"
        assert int(mm.group(1)) == expect, f"{name}: waits for vmcnt({mm.group(1)}), the plan has {expect} frame loads"
        between = [ins for ins in body[last_dma + 1:bar - 1] if VMEM.match(ins)]
        want = loads if pair else expect
        assert len(between) == want, (f"{name}: {len(between)} vector-memory instructions between the last table piece and "
                                      f"s_waitcnt vmcnt({expect}): {between}")
        form = "global_load_dword " if pair else "global_load_dwordx2"
        assert all(ins.startswith(form) for ins in between), f"{name}: unexpected frame-load form {sorted(set(between))[:3]}"
        checked += 1
    assert checked >= 60, f"only {checked} instantiations checked"   # 7 plans x (6 + 2 strided) + the ARGB-out forms of 1024 and 4096 "B"


def test_runs_kernels_load_half_a_frame_inside_a_run(tmp_path):
    """STREAM == 2 ("runs", round 6): a wavefront transforms kRunLen consecutive columns of a row and keeps the overlapped half of the raw frame
    in registers -- which rounds load the whole frame (P loads: the prologue and the last round of a run) and which only the new hop (P/2) is a
    compile-time fact, so the count of frame loads in the code object is fixed: P (prologue) + [(RL - 1) * P/2 + P] (the run inside the loop)
    + (RL - 1) * P/2 (the peeled last run, whose last round prefetches nothing).  A conditional (run-time) choice between the two forms costs a
    register copy per value (measured: 8 v_mov per round) and would show here as a different count.  No scratch, no spill."""
    text = _disassemble(tmp_path)
    src = open(os.path.join(ROOT, "jadespectrogram_amd", "csrc", "jsg_stft_kernel.h")).read()
    rl = int(re.search(r"#define JSG_X_RUNLEN (\d+)", src).group(1))
    checked = 0
    for name, body in _functions(text):
        if "stft_db_kernel" not in name or not re.search(r"EELi\dELi\dELi2ELi\dEEEv", name):
            continue
        m = re.search(r"3CfgI((?:Li\d+E)+)E", name)
        vals = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        n, lanes = vals[0], vals[4]
        p = n // 2 // lanes
        loads = [ins for ins in body if ins.startswith("global_load_dwordx2")]
        want = p + ((rl - 1) * (p // 2) + p) + (rl - 1) * (p // 2)
        assert len(loads) == want, f"{name}: {len(loads)} frame loads, the run structure (run length {rl}) has {want}"
        assert not any(ins.startswith(("scratch_", "buffer_load", "buffer_store")) for ins in body), f"{name}: scratch / buffer accesses"
        checked += 1
    assert checked == 2, f"{checked} runs instantiations found (Cfg1024, one channel per column, the two logarithms)"


def test_no_fused_lds_pairs_in_the_stft_kernels(tmp_path):
    """ds_read2_b64 / ds_write2_b64 halve the LDS rate of 8-byte accesses on gfx950 and bank differently from what the exchange
    layouts were searched for (MI355X_MICROARCH.md, LDS): the backend pass that forms them is switched off for the kernel
    (JSG_NO_LDS_MERGE) and the one place where the IR vectorizer would (stage-3 reads of the AZ == 1 plans) is written so that
    it cannot.  A toolchain change that brings them back shows up here."""
    text = _disassemble(tmp_path)
    for name, body in _functions(text):
        if "stft_db_kernel" not in name:
            continue
        fused = [ins for ins in body if ins.startswith(("ds_read2", "ds_write2"))]
        assert not fused, f"{name}: {len(fused)} fused LDS pair instructions, e.g. {fused[0]}"


def test_early_first_layer_leaves_no_register_copies_in_front_of_the_frame_loads(tmp_path):
    """Cfg::EARLY1 (round 5): the first butterfly layer runs ahead of the next round's frame loads, so the loads land in the registers the raw
    samples leave.  Without it the compiler copied the raw lower inputs out of the way -- 32 v_mov per FFT round in Cfg4096B's loop.  The guard:
    in the 60 instructions in front of every burst of frame loads of an EARLY1 instantiation there are at most four VGPR-to-VGPR moves."""
    text = _disassemble(tmp_path)
    checked = 0
    for name, body in _functions(text):
        if "stft_db_kernel" not in name:
            continue
        m = re.search(r"3CfgI((?:Li\d+E)+)E", name)
        vals = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        if len(vals) < 14 or vals[13] != 1:      # Cfg<..., FPW, EARLY1, TWF, PAIR>
            continue
        loads = [i for i, ins in enumerate(body) if ins.startswith("global_load_dwordx2") or ins.startswith("global_load_dword ")]
        bursts = [i for k, i in enumerate(loads) if k == 0 or i - loads[k - 1] > 40]
        assert bursts, name
        for b in bursts:
            window = body[max(0, b - 60):b]
            copies = [ins for ins in window if re.match(r"v_mov_b32_e32 v\d+, v\d+$", ins) or re.match(r"v_mov_b64_e32 v\[\d+:\d+\], v\[\d+:\d+\]$", ins)]
            assert len(copies) <= 4, f"{name}: {len(copies)} register copies in front of the frame loads at instruction {b}"
        checked += 1
    assert checked >= 30   # Cfg2048, Cfg4096, Cfg4096B in all their output forms


def test_register_resident_twiddles_keep_five_waves_per_simd(tmp_path):
    """Cfg::RTAB (round 5): the one-channel 1024-point dB kernels hold two lane tables in registers.  Their LDS allows five waves per SIMD, i.e.
    96 VGPRs: the guard reads the register count of those instantiations from the code object's metadata (and that nothing spills)."""
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip(f"{tool} not found under {LLVM}")
    if not os.path.exists(OBJS[0]):
        from jadespectrogram_amd import _build
        _build.build_lib()
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", OBJS[0], str(tmp_path / "scratch.o")])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}"])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    checked = 0
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        m = re.search(r"stft_db_kernelINS_3CfgI((?:Li\d+E)+)EELi(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
        if not m:
            continue
        vals = [int(v) for v in re.findall(r"Li(\d+)E", m.group(1))]
        mixop, outk = int(m.group(2)), int(m.group(3))
        if not (len(vals) >= 17 and vals[16] == 1 and mixop == 3 and outk == 0 and int(m.group(4)) == 1):   # Cfg<..., PAIR, RTAB>, one channel per column, dB / power out, strided
            continue
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        assert vgpr <= 96 and spill == 0, f"{name}: {vgpr} VGPRs, {spill} spilled"
        checked += 1
    assert checked == 2   # strided dispatches: v_log / exact log (single launches keep the LDS reads: one frame per wave)
