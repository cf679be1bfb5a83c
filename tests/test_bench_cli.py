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
    assert "4096 frames/launch" in j["config"]["workload"] and j["config"]["launches_per_step"] == 1024
    assert j["value"] is None and j["vs_baseline"] is None and j["dtype"] == "f32"


def test_two_ranks_dry_run_over_gloo():
    j = _run(_torchrun(2, ["--dry-run", "--steps", "4", "--warmup", "1"]))
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["dry_run"] is True and j["ms_per_step"] > 0


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_and_run_the_kernels():
    j = _run(_torchrun(2, ["--steps", "2", "--warmup", "1", "--launches-per-step", "64", "--no-cpu-baseline"]),
             env={"JSG_BENCH_BACKEND": "gloo"})
    assert REQUIRED <= set(j) and j["n_gpus"] == 2 and j["value"] > 1e7            # north_star's >= 1e7 frames/s, whole job
    assert j["config"]["launches_per_step"] == 64 and j["roofline"]["avg_launch_us"] > 0
    assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 2 * 64 * 4096) < 1.0   # value = all ranks' frames / timed wall


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,unit", [("c3", "frames/s"), ("c5", "columns/s")])
def test_other_configs_emit_a_line(cfg, unit):
    j = _run([sys.executable, "bench.py", "--config", cfg, "--steps", "2", "--warmup", "1", "--launches-per-step", "8", "--no-cpu-baseline"])
    assert j["unit"] == unit and j["value"] > 0 and j["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert j["parity"]["max_rel_power_err_bins_within_20dB_of_peak"] < 1e-5
    assert j["parity"]["colour_index_flips_end_to_end"] <= max(4, j["parity"]["pixels_checked"] // 500)
    if cfg == "c5":   # default: strided batches, one kernel launch for a whole rotation of images; the pixels are those of single launches
        k = j["config"]["images_per_launch"]
        assert k > 8 and "jsg_stft_image_launch_strided" in j["config"]["issue"] and j["config"]["launches_per_step"] == 8
        assert j["roofline"]["algorithmic_bytes_per_launch"] == k * 1875 * 12292
        assert j["parity"]["strided_batch_pixels_differing_from_single_launches"] == 0
        assert abs(j["ms_per_step"] * j["steps"] * 1e-3 * j["value"] - 2 * 8 * k * 1875) < 1.0


@pytest.mark.gpu
def test_c5_one_image_per_launch_mode_still_runs():
    j = _run([sys.executable, "bench.py", "--config", "c5", "--images-per-launch", "1", "--steps", "2", "--warmup", "1", "--launches-per-step", "9",
              "--no-cpu-baseline", "--no-boundary"])
    assert j["config"]["images_per_launch"] == 1 and j["config"]["hip_streams_per_gpu"] == 3 and j["value"] > 0
    assert j["roofline"]["algorithmic_bytes_per_launch"] == 1875 * 12292 and "strided_batch_pixels_differing_from_single_launches" not in j["parity"]


@pytest.mark.gpu
def test_default_config_line_carries_roofline_boundary_and_default_environment():
    """The driver's line (configs[1]) with few launches per step: the objects the contract asks for are there and consistent --
    roofline (fraction, source, traffic with its staleness flag, the event-timed figure beside it), parity of the kernel that is
    timed, the boundary block (producer latency under a reading consumer, PCIe-inclusive rate) and the same region in the
    default environment (no extra hardware queues)."""
    j = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--launches-per-step", "64", "--no-cpu-baseline"], timeout=900)
    assert REQUIRED <= set(j) and j["n_gpus"] == 1 and j["value"] > 1e7 and j["scaling"] == "weak" and j["dtype"] == "f32"
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["frac_event_timed"] > 0 and r["algorithmic_bytes_per_launch"] == 4096 * 4100
    assert r["traffic_source"] is None or isinstance(r["traffic_source"]["matches_this_build"], bool)
    assert (r["frac_rocprof"] is not None) == bool(r["traffic_source"] and r["traffic_source"]["matches_this_build"])
    assert r["second_roof"]["bound"] == "valu_issue"
    assert j["parity"]["kernel"] == "Cfg1024" and j["parity"]["fused_image_pixels_differing_from_two_kernel_image"] == 0
    b = j["boundary"]
    assert b["process_block_latency"]["ring_bit_identical_to_undisturbed_batch_run"] is True and b["process_block_latency"]["p50_us"] < 100.0
    assert b["pcie_inclusive_rate"]["host_memory"]["pinned"]["frames_per_s"] > 1e6
    d = j["config"]["same_region_default_environment"]
    assert d["value"] > 1e7 and "unset" in d["GPU_MAX_HW_QUEUES"]
    assert "jsg_stft_db_launch_batches" in j["config"]["issue"]
