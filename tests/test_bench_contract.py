"""bench.py's output contract, checked on the committed line of the last measured run (profiles/r02_bench_line.json)
and on the argument parser (no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r02_bench_line.json")) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].replace("x", "×") == base["metric"] or d["metric"] == base["metric"].replace("×", "x")
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    for key in ("cpu_baseline", "cpu_baseline_all_cores"):
        c = d[key]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["cpu"]
    assert d["cpu_baseline"]["cores"] == 4 and d["cpu_baseline_all_cores"]["cores"] <= 16
    # value = faces the step produced / step time
    faces = d["config"]["faces_per_step_rank0"] * d["n_gpus"]
    assert abs(d["value"] - faces / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_cli_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no RANK in the environment must launch two ranks through torch.distributed.run
    (the driver's own launch line) and relay rank 0's line.  --dry-run: no GPU here, so the ranks meet on gloo and go through
    the barrier / max-over-ranks protocol around empty steps; the line says so (value null, dry_run true)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                                   # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["scaling"] == "weak"
    assert d["units_all_ranks"] == 2 * 3 * 128                           # both ranks' units were summed
    assert "x2" in d["config"]["parallelism"]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
