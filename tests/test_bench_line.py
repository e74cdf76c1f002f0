"""The driver keeps 8 KB of bench.py's stdout tail: the ONE line it parses must fit with room to spare (round 4's grew to 20.8 KB and was
recorded as unparsed).  Builds the line from a canned full result object (profiles/r04_bench.json, the largest one any round produced)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline")


def canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))


def test_line_fits():
    full = canned()
    assert len(json.dumps(full)) > 8192          # the object that broke round 4
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    back = json.loads(line)
    for k in CONTRACT_KEYS:
        assert k in back, k
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    rf = back["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "mean_launch_ms", "selection"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["bound"] in ("hbm", "mfma")
    assert rf["selection"].startswith("largest time per step among the kernels of the critical")  # the rule stays fixed
    ro = back["roofline_overlapped"]  # the longest kernel of the matcher stream, printed beside it
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "mean_launch_ms", "selection"):
        assert k in ro, k
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and ro["kernel"] != rf["kernel"]
    cb = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert "workload" in back["config"] and "model" not in back["config"]


def test_every_kernel_class_has_its_rocprof_name():
    """From the line (class names of svgpu_profile_*) to profiles/rNN_kernel_stats.csv (kernel names) without reading the source."""
    for cls in bench.KERNEL_CLASSES:
        assert cls in bench.ROCPROF_KERNEL, cls
    src = open(os.path.join(ROOT, "stella_vslam_amd", "csrc", "orb_kernels.hip")).read() + open(os.path.join(ROOT, "stella_vslam_amd", "csrc", "match_kernels.hip")).read()
    for cls, name in bench.ROCPROF_KERNEL.items():
        assert ("void " + name.split("<")[0] + "(") in src, name


def test_line_survives_bloated_legs():
    """A leg that grows (or fails with a long message) may cost its own summary, never the contract keys."""
    full = canned()
    full["tracked_frame"]["error"] = "x" * 100000
    full["global_ba"]["error"] = "y" * 100000
    full["roofline"]["kernels"] = full["roofline"]["kernels"] * 50
    full["second_batch_point"] = {"frames_per_gpu_per_step": 256, "value": 1.0, "unit": "frames/s", "ms_per_step": 1.0}
    line = bench.compact_line(full)
    assert len(line) < 4096
    back = json.loads(line)
    for k in CONTRACT_KEYS:
        assert k in back, k


def test_emit_prints_the_line_last(tmp_path, capsys):
    full = canned()
    bench.emit(full, str(tmp_path / "bench_detail.json"))
    cap = capsys.readouterr()
    lines = cap.out.strip().split("\n")
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["value"] == full["value"]
    assert json.load(open(tmp_path / "bench_detail.json")) == full
    assert json.loads(cap.err.strip().split("\n")[-1]) == full
