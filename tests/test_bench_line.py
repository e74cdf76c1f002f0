"""bench.py's roofline object without a GPU: which kernel it names, what rides along, and the "valu" relabelling."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fe(B, times_ms):
    b = _bench()
    level_px = [640 * 480, 533 * 400, 444 * 333, 370 * 278, 309 * 231, 257 * 193, 214 * 161, 179 * 134]
    steps = 10
    return b, {"per_kernel": {k: (t * steps, steps) for k, t in times_ms.items()}, "steps": steps, "mfma_ops": 7.4e11 * steps,
               "alg": b.algorithmic_bytes(level_px, 2410.0, B), "level_px": level_px}


TIMES = {"k_resize": 1.55, "k_blur": 0.66, "k_fast": 1.19, "k_select": 0.05, "k_describe": 1.26, "k_bf_binsort": 0.17, "k_bf_topk": 1.09, "k_bf_replay": 2.22}


def test_roofline_object_names_the_longest_kernel_of_the_critical_stream(monkeypatch):
    b, fe = _fe(1024, TIMES)
    monkeypatch.setattr(b, "load_profile_json", lambda *a, **k: None)   # no PMC files for these sources
    kernels, dom = b.roofline_entries(fe, "nohash", 1024, 1)
    assert [k["kernel"] for k in kernels] == list(b.KERNEL_CLASSES)      # every kernel class is in the line
    assert dom["kernel"] == "k_resize" and dom["rocprof_kernel"] == "k_pyramid_lds" and dom["bound"] == "hbm"
    assert dom["overlapped_stream_longest"]["kernel"] == "k_bf_replay"   # longer by elapsed time, but in the extraction's shadow: reported beside it
    mf = dom["matrix_core_kernel"]
    assert mf["kernel"] == "k_bf_topk" and mf["bound"] == "mfma" and abs(mf["frac"] - 7.4e11 / 1.09e-3 / 1e12 / b.I8_MFMA_PEAK_TOPS) < 1e-4
    assert dom["pmc_profiles_match_sources"] is False and dom["valu_issue"] is None
    assert all(k["bound"] in ("hbm", "mfma") for k in kernels)          # nothing is called VALU-bound without the counters to show it
    # achieved = algorithmic bytes per launch / mean launch time
    ent = next(k for k in kernels if k["kernel"] == "k_fast")
    assert abs(ent["achieved"] - fe["alg"]["k_fast"] / 1.19e-3 / 1e9) < 0.01 and abs(ent["frac"] - ent["achieved"] / 8000.0) < 1e-4


def test_valu_bound_kernels_are_labelled_from_the_pmc_passes(monkeypatch):
    b, fe = _fe(1024, TIMES)
    valu = {"csrc_hash": "h", "batch": 1024, "simds": 1024, "cycles_per_valu_wave_inst": 4,
            "kernels": {"k_fast": {"valu_wave_insts": 669e6, "kernel_cycles": 2.77e6, "valu_issue_frac": 0.94},
                        "k_resize": {"valu_wave_insts": 295e6, "kernel_cycles": 1.46e6, "valu_issue_frac": 0.79},
                        "k_select": {"valu_wave_insts": 5e6, "kernel_cycles": 1e5, "valu_issue_frac": 0.24}}}
    monkeypatch.setattr(b, "load_profile_json", lambda pattern, *a, **k: valu if "valu" in pattern else None)
    kernels, dom = b.roofline_entries(fe, "h", 1024, 1)
    by = {k["kernel"]: k for k in kernels}
    assert by["k_fast"]["bound"] == "valu" and by["k_fast"]["frac"] == 0.94 and by["k_fast"]["hbm_context"]["unit"] == "GB/s"
    assert by["k_select"]["bound"] == "hbm"                               # an issue fraction below one half is context, not the bound
    assert by["k_describe"]["bound"] == "hbm" and by["k_describe"]["valu_issue"] is None
    # the top-level object keeps the contract's byte roofline for the kernel it names and carries the issue bound beside it
    assert dom["kernel"] == "k_resize" and dom["bound"] == "hbm" and dom["unit"] == "GB/s" and dom["valu_bound"]["frac"] == 0.79
    assert abs(dom["frac"] - by["k_resize"]["hbm_context"]["frac"]) < 1e-9


def test_small_batches_keep_the_same_rule(monkeypatch):
    b, fe = _fe(256, {"k_resize": 0.31, "k_blur": 0.21, "k_fast": 0.33, "k_select": 0.04, "k_describe": 0.331, "k_bf_binsort": 0.09, "k_bf_topk": 0.64, "k_bf_replay": 0.21})
    monkeypatch.setattr(b, "load_profile_json", lambda *a, **k: None)
    _, dom = b.roofline_entries(fe, "nohash", 256, 1)
    assert dom["kernel"] == "k_describe" and dom["overlapped_stream_longest"]["kernel"] == "k_bf_topk"
