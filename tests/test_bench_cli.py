"""bench.py's command-line contract and its N > 1 path (the multi-GPU bench had never executed anywhere).

CPU: `--dry-run` runs the whole N-rank plumbing (process group, barriers, max-over-ranks reduction, one JSON line from rank 0)
without touching a GPU.  GPU (one device): two ranks share the device over gloo and run the real kernels."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}: {r.stdout[-1000:]}"
    return json.loads(lines[0])


def _torchrun(nproc, extra):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), "bench.py", "--gpus", str(nproc)] + extra


def test_single_process_dry_run_prints_the_contract_fields():
    j = _run([sys.executable, "bench.py", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert REQUIRED <= set(j) and j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["dry_run"] is True
    assert j["metric"] == "STFT frames/sec (1024-pt, 50% hop)" and j["unit"] == "frames/s" and j["scaling"] == "weak"
    assert "4096 frames/launch" in j["config"]["workload"] and j["config"]["dispatches_per_step"] == 16 and j["config"]["batches_per_dispatch"] == 64
    assert "16 dispatches x 64 batches x 4096 frames" in j["config"]["step"] and j["config"]["hip_streams_per_gpu"] == 1
    assert j["value"] is None and j["vs_baseline"] is None and j["dtype"] == "f32"


def test_bench_does_not_touch_the_hardware_queue_variable():
    """Rounds 2-3 leaned on GPU_MAX_HW_QUEUES (set by bench.py and by a library constructor); the strided launch needs neither."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "GPU_MAX_HW_QUEUES\", \"" not in src and "setdefault(\"GPU_MAX" not in src and 'environ["GPU_MAX_HW_QUEUES"]' not in src
    eng = open(os.path.join(ROOT, "jadespectrogram_amd", "csrc", "jsg_engine.cpp")).read()
    assert "setenv(" not in eng.replace("setenv is not thread-safe", "")


def test_two_ranks_dry_run_over_gloo():
    j = _run(_torchrun(2, ["--dry-run", "--steps", "4", "--warmup", "1"]))
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["dry_run"] is True and j["ms_per_step"] > 0


def test_plain_gpus_2_starts_its_own_ranks_and_times_configs3():
    """`python bench.py --gpus N` with NO launcher around it -- the parent (which never imports torch or touches a GPU) starts the
    N ranks itself, rank 0's line comes through, and the N > 1 default workload is BASELINE configs[3]: 8 channels per GPU, one column per channel."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert REQUIRED <= set(j) and j["n_gpus"] == 2 and j["steps"] == 3 and j["dry_run"] is True and j["scaling"] == "weak"
    assert j["config"]["workload"].startswith("configs[3]") and "2 GPUs x 8 channels = 16 channels" in j["config"]["workload"]
    assert j["config"]["channels_per_gpu"] == 8 and j["config"]["channels_total"] == 16 and j["config"]["frames_per_batch"] == 8 * 4096
    assert j["metric"] == "STFT frames/sec (1024-pt, 50% hop)"


def test_the_parent_of_a_self_started_run_does_not_import_torch():
    """... and cannot have initialised the GPU: the spawn happens before bench.py's first `import torch`."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("spawn_ranks(args.gpus") < main.index("import torch")
    body = src[src.index("def spawn_ranks("):src.index("def main():")]
    assert "import torch" not in body and "os.exec" not in body and "subprocess.run" in body


def test_algorithmic_bytes_of_every_config():
    import bench
    assert bench.algorithmic_bytes_per_batch(bench.CONFIGS["c2"]) == 4096 * 4100
    assert bench.algorithmic_bytes_per_batch(bench.CONFIGS["c3"]) == 4096 * 20484
    assert bench.algorithmic_bytes_per_batch(bench.CONFIGS["c4"]) == 8 * 4096 * 4100          # "C4: as C2 per channel-frame"
    assert bench.algorithmic_bytes_per_batch(bench.CONFIGS["c5"]) == 1875 * 12292


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_and_run_the_kernels():
    j = _run(_torchrun(2, ["--config", "c2", "--steps", "2", "--warmup", "1", "--dispatches-per-step", "2", "--nbuf", "8", "--no-cpu-baseline"]),
             env={"JSG_BENCH_BACKEND": "gloo"})
    assert REQUIRED <= set(j) and j["n_gpus"] == 2 and j["value"] > 1e7            # north_star's >= 1e7 frames/s, whole job
    assert j["config"]["dispatches_per_step"] == 2 and j["config"]["batches_per_dispatch"] == 8 and j["roofline"]["avg_dispatch_us"] > 0
    assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 2 * 2 * 8 * 4096) < 1.0   # value = all ranks' frames / timed wall


@pytest.mark.gpu
def test_two_ranks_default_to_configs3_and_run_the_per_channel_kernels():
    """The N > 1 default (configs[3]): two ranks share the one GPU of the test box over gloo, 8 channels each, one column per channel."""
    j = _run(_torchrun(2, ["--steps", "2", "--warmup", "1", "--dispatches-per-step", "2", "--nbuf", "3", "--no-cpu-baseline"]),
             env={"JSG_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["config"]["workload"].startswith("configs[3]") and j["config"]["channels_total"] == 16
    assert "Cfg1024" in j["roofline"]["kernel"] and "one channel per column" in j["roofline"]["kernel"]
    assert j["roofline"]["algorithmic_bytes_per_dispatch"] == 3 * 8 * 4096 * 4100
    assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 2 * 2 * 3 * 8 * 4096) < 1.0
    assert abs(j["config"]["per_gpu_value"] * 2 - j["value"]) < 1e-3 * j["value"]


@pytest.mark.gpu
def test_one_rank_process_group_on_rccl_runs_the_n_gt_1_calls():
    """What the driver's 8-GPU run does through torch.distributed -- init_process_group("nccl", device_id=...), barriers, the max-over-ranks
    all_reduce of a CUDA tensor, destroy -- executed with ONE rank on the one GPU of the test box (RCCL had never run under bench.py)."""
    j = _run([sys.executable, "bench.py", "--config", "c4", "--steps", "2", "--warmup", "1", "--dispatches-per-step", "2", "--nbuf", "3", "--no-cpu-baseline",
              "--no-calibration", "--no-boundary", "--no-parity", "--no-power", "--no-single"], env={"JSG_BENCH_DIST_SINGLE": "1"})
    assert j["n_gpus"] == 1 and j["value"] > 1e7 and j["roofline"]["avg_dispatch_us"] > 0


@pytest.mark.gpu
def test_plain_gpus_2_self_started_runs_the_kernels_on_the_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it, real kernels: the parent starts two ranks that share the test box's one GPU (gloo for
    the barriers; on the driver's node the default is one rank per GPU over RCCL) and time configs[3]."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["JSG_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dispatches-per-step", "2", "--nbuf", "3", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["workload"].startswith("configs[3]") and j["value"] > 1e7
    assert "runs of 2 consecutive columns" in j["roofline"]["kernel"]


@pytest.mark.gpu
def test_c4_shard_line_on_one_gpu():
    j = _run([sys.executable, "bench.py", "--config", "c4", "--steps", "2", "--warmup", "1", "--dispatches-per-step", "2", "--no-cpu-baseline", "--no-calibration",
              "--no-boundary"])
    k = j["config"]["batches_per_dispatch"]
    assert j["unit"] == "frames/s" and j["value"] > 1e7 and k >= 8 and j["config"]["column_layout"].startswith("reference")
    assert j["roofline"]["algorithmic_bytes_per_dispatch"] == k * 8 * 4096 * 4100 and 0.05 < j["roofline"]["frac"] < 1.0
    assert j["parity"]["kernel"] == "Cfg1024" and j["parity"]["max_rel_power_err_bins_within_20dB_of_peak"] < 1e-5
    assert j["parity"]["strided_columns_differing_from_single_launches"] == 0
    assert j["parity"]["colour_index_flips_end_to_end"] <= max(4, j["parity"]["pixels_checked"] // 50000)
    assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 2 * k * 8 * 4096) < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,unit", [("c3", "frames/s"), ("c5", "columns/s")])
def test_other_configs_emit_a_line(cfg, unit):
    j = _run([sys.executable, "bench.py", "--config", cfg, "--steps", "2", "--warmup", "1", "--dispatches-per-step", "2", "--no-cpu-baseline", "--no-calibration"])
    k = j["config"]["batches_per_dispatch"]
    assert j["unit"] == unit and j["value"] > 0 and k >= 8
    assert j["parity"]["max_rel_power_err_bins_within_20dB_of_peak"] < 1e-5
    assert j["parity"]["colour_index_flips_end_to_end"] <= max(4, j["parity"]["pixels_checked"] // 500)
    if cfg == "c5":   # strided batches, one kernel launch for a whole rotation of images; the pixels are those of single launches
        assert "jsg_stft_image_launch_strided" in j["config"]["issue"]
        assert j["roofline"]["algorithmic_bytes_per_dispatch"] == k * 1875 * 12292
        assert j["parity"]["strided_pixels_differing_from_single_launches"] == 0
        assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 2 * k * 1875) < 1.0
    else:
        assert "jsg_stft_db_launch_strided" in j["config"]["issue"] and "Cfg2048B" in j["roofline"]["kernel"]
        assert j["roofline"]["algorithmic_bytes_per_dispatch"] == k * 4096 * 20484
        assert j["parity"]["strided_columns_differing_from_single_launches"] == 0
        assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 2 * k * 4096 * 8) < 1.0


@pytest.mark.gpu
def test_single_mode_one_dispatch_per_batch_still_runs():
    j = _run([sys.executable, "bench.py", "--mode", "single", "--steps", "2", "--warmup", "1", "--dispatches-per-step", "32", "--no-cpu-baseline",
              "--no-boundary", "--no-calibration", "--no-extra"])
    assert j["config"]["batches_per_dispatch"] == 1 and j["value"] > 1e7 and j["roofline"]["algorithmic_bytes_per_dispatch"] == 4096 * 4100


@pytest.mark.gpu
def test_default_config_line_carries_roofline_calibration_boundary_and_the_other_configs():
    """The driver's line (configs[1]) with few dispatches per step: the objects the contract asks for are there and consistent --
    roofline measured live on the timed region (fraction, per-dispatch duration, the whole-region figure beside it, traffic with its
    staleness flag, the one-batch-per-dispatch figure), the measured copy roof, parity of the kernel that is timed, the boundary block
    (producer latency under a reading consumer, PCIe-inclusive rate), and the c3 / c5 results from their child processes."""
    j = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--dispatches-per-step", "2", "--no-cpu-baseline"], timeout=1200)
    assert REQUIRED <= set(j) and j["n_gpus"] == 1 and j["value"] > 1e7 and j["scaling"] == "weak" and j["dtype"] == "f32"
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_dispatch"] == 64 * 4096 * 4100 and r["algorithmic_bytes_per_batch"] == 4096 * 4100
    assert abs(r["timed_region_frac"] / r["frac"] - 1.0) < 0.05          # one stream, back to back: the region IS its dispatches
    assert r["traffic_source"] is None or isinstance(r["traffic_source"]["matches_this_build"], bool)
    assert (r["frac_rocprof"] is not None) == bool(r["traffic_source"] and r["traffic_source"]["matches_this_build"])
    assert r["second_roof"]["bound"] == "valu_issue" and "Cfg1024" in r["kernel"]
    assert 0.2 < r["one_batch_per_dispatch"]["frac"] < r["frac"]
    # the headline is the reference's column layout, and every qualifier is a SCALAR of `roofline`
    assert r["column_layout"] == "reference" and j["config"]["column_layout"].startswith("reference") and r["frac_reference_layout"] == r["frac"]
    assert 0.2 < r["frac_tail_plane"] < 1.0 and r["other_column_layout"]["values_identical_to_the_timed_layout"] is True
    assert r["frac_one_batch_per_dispatch"] == r["one_batch_per_dispatch"]["frac"] and r["second_roof_frac"] == r["second_roof"]["frac"]
    for cfg in ("c3", "c4", "c5"):
        assert isinstance(r[f"extra_{cfg}_frac"], float) and r[f"extra_{cfg}_frac"] == j["extra"][cfg]["roofline"]["frac"]
        assert r[f"extra_{cfg}_value"] == j["extra"][cfg]["value"]
    cal = j["calibration"]
    assert 4000.0 < cal["peak_copy_GBps"] < 8000.0 and cal["sizes"]["8.4MB"]["GBps"] < cal["peak_copy_GBps"]
    assert j["config"]["GPU_MAX_HW_QUEUES"] is None and "jsg_stft_db_launch_strided" in j["config"]["issue"]
    assert j["parity"]["kernel"] == "Cfg1024" and j["parity"]["fused_image_pixels_differing_from_two_kernel_image"] == 0
    assert 0.0 <= j["parity"]["float32_cpu_fft_frac_bins_rel_power_err_gt_1e-5"] < 0.05      # what a single-precision CPU FFT does on the same frames
    assert j["parity"]["strided_columns_differing_from_single_launches"] == 0
    b = j["boundary"]
    assert b["process_block_latency"]["ring_bit_identical_to_undisturbed_batch_run"] is True and b["process_block_latency"]["p50_us"] < 100.0
    assert b["pcie_inclusive_rate"]["host_memory"]["pinned"]["frames_per_s"] > 1e6
    for cfg, kern in (("c3", "Cfg2048B"), ("c5", "Cfg4096B"), ("c4", "Cfg1024")):
        e = j["extra"][cfg]
        assert "error" not in e, e
        assert e["value"] > 1e6 and kern in e["kernel"] and 0.05 < e["roofline"]["frac"] < 1.0
