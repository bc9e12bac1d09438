"""bench.py prints ONE JSON line with the fields the driver and the judge read (task contract, section 4)."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_contract(tmp_path):
    env = dict(os.environ, FLX_BENCH_TRIS="30000", FLX_BVH_CACHE=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3", "--num-tasks", "262144"],
                         env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "Mrays/s" and j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 3 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"].startswith("synthetic") and "night.hdr" in j["data"]
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # the committed PMC capture belongs to the default configuration (and to the kernel sources it was taken on): here it must be
    # refused, and then `achieved` / `frac` are null -- never another quantity under the same name (frac_own carries the upper bound)
    assert r["traffic"] is None and r["achieved"] is None and r["frac"] is None and r["frac_source"] is None
    assert r["traffic_capture_stale_keys"] and "num_tasks" in r["traffic_capture_stale_keys"]
    assert 0.0 < r["frac_own"] < 1.5
    # the other roofs: null without a capture as well, but the blocks are there and the three priced kernels are timed inside the timed region
    assert r["valu"] is None and set(r["other_kernels"]) == {"logic", "shadow"}
    for k, o in r["other_kernels"].items():
        assert o["launch_ms"] > 0 and o["timed"] == "timed region" and o["hbm"]["frac"] is None and o["valu"] is None, (k, o)
    assert "shadow_split" in r["capture_key"] and r["capture_key"]["library"] == "libfluctus_hip.so" and "gfx950" in r["capture_key"]["build_flags"]
    # protocol: settle >= 2 * maxBounces + 2 whatever --warmup says, five windows, headline = the median one
    assert j["settle_iterations"] >= 2 * j["config"]["max_bounces"] + 2 and j["settle_iterations"] >= j["warmup"]
    w = j["windows"]
    assert w["count"] == 5 and len(w["Mrays_s"]) == 5 and sorted(w["Mrays_s"])[2] == pytest.approx(j["value"], rel=1e-9)
    assert "extend" in j["kernel_ms_avg_source"]["timed_region"]    # the roofline kernel is event-timed inside the timed region
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    # whole-job throughput = rays of the timed steps / time
    rays = j["rays"]["extension"] + j["rays"]["shadow"]
    assert abs(j["value"] - rays / (j["ms_per_step"] * 1e-3 * j["steps"]) / 1e6) <= 1e-6 * j["value"]
