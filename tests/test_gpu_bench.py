"""The driver's contract of bench.py on one GPU, at a scale that takes seconds: ONE JSON line on stdout with the metric of
BASELINE.json, the roofline of the dominant kernel from HIP events over the timed region, the step-level roofline, the CPU
baseline object, the three numbers of SURVEY.md 8(d) and the full-size parity verdict."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "0.02", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Gbases/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["aligned_bases_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["launches"] == 4
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["avg_launch_ms"] < d["ms_per_step"]
    assert 0 < d["step_roofline"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["port"] == "libgci_cpu" and cb["cores"] == os.cpu_count() and cb["value"] > 0
    assert cb["unit"] == "Gbases/s" and cb["sample"].startswith("100 %") and cb["equal_to_the_gpu_track"] is True
    assert cb["oracle_port_on_a_sample"]["value"] > 0
    assert d["parity_vs_oracle_full_size"] is True
    s8, cfg = d["survey_8d"], d["config"]
    # `value` is SURVEY 8(d)'s window with the inputs resident in HBM (record pages made, keys copied home inside every step); the
    # kernels-only figure of rounds 3 - 5 and the numbers (2) / (3) are FLAT SCALARS of config (the driver's record keeps those)
    assert "record pages" in cfg["value_is"] and cfg["n1_kernels_only_gbases_per_s"] >= d["value"] > 0
    assert s8["1_kernels_only_gbases_per_s"] == cfg["n1_kernels_only_gbases_per_s"] and s8["value_window_gbases_per_s"] == d["value"]
    for k in ("n1_kernels_only_ms_per_step", "n2_device_pipeline_gbases_per_s", "n2_device_pipeline_s", "n3_cli_genome_s", "n3_cli_genome_gbases_per_s",
              "n3_cli_first_pass_s", "n3_inflate_device_s", "n3_first_inflate_call_s", "n3_start_s", "n3_exit_s", "n3_bgzf_gb"):
        assert isinstance(cfg[k], (int, float)) and cfg[k] > 0, k
    assert cfg["n3_parity"] is True and "inflate" in cfg["cpu_baseline_excludes"] and "NO BGZF inflate" in d["cpu_baseline"]["sample"]
    assert d["step_roofline"]["algorithmic_bytes_per_step"]["record_pages"] > 0
    assert cfg["n3_start_s"] < 1.0                 # (a process that does not import torch)
    assert s8["2_device_pipeline_incl_h2d_d2h"]["seconds"] > 0 and s8["3_command_line_chr19_realistic_bam"]["seconds"] > 0
    g = s8["3_command_line_genome"]              # the command line as a process of its own on the two BGZF files of the workload
    assert g["parity"] is True and len(g["parity_vs_oracle_on_contigs"]) >= 3 and g["seconds"] > 0
    assert g["phases_device_s"]["bgzf_inflate + crc"] > 0 and "GCI.depth.gz" in g["outputs_bytes"] and g["deflate_ratio"] > 1.5
    sw = d["survey_window_step"]                  # the window SURVEY 8(d) defines: no text, keys on the host every step
    assert 0 < sw["ms_per_step"] and sw["gbases_per_s"] > 0 and sw["k_tile_build_avg_launch_ms"] > 0 and 0 < sw["hbm_frac"] < 1
    two = d["two_steps_in_flight"]
    assert two["status_ok"] and two["same_track_as_one_in_flight"]
