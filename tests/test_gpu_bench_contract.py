"""bench.py prints exactly one JSON line on stdout with the fields the measurement contract names"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]       # nothing but the result on stdout
    return json.loads(lines[0])


def test_bench_line_small_pipe(gpu):
    j = _run(["--nx", "64", "--ny", "66", "--nz", "66", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["unit"] == "MLUPS" and j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["dtype"] == "f64"
    assert j["higher_is_better"] is True and j["vs_baseline"] is None and j["scaling"] == "weak" and j["data"] == "synthetic"
    assert abs(j["value"] - 64 * 66 * 66 / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches"] == 20 and rf["launches_per_step"] == 1.0
    # units of one launch = the nodes it visits; the whole-box figure stays next to it, labelled as the convention it is
    assert abs(rf["achieved"] * 1e9 - j["active_node_fraction"] * 64 * 66 * 66 * rf["bytes_per_node"] / (rf["collide_ms_per_step"] * 1e-3)) < 1e-6 * rf["achieved"] * 1e9
    assert abs(rf["frac_whole_box_convention"] * j["active_node_fraction"] - rf["frac"]) < 1e-9
    assert j["roofline_alone"]["launches"] == 10 and j["roofline_alone"]["frac"] > rf["frac"] * 0.8   # the kernel with the GPU to itself, measured after the timed region
    assert rf["copy_GBps_this_gpu"] > 1000 and rf["traffic"] is None and rf["frac_real_traffic"] is None   # the PMC figure belongs to the 256^3 headline workload only
    # the line says what the 353 B/node convention hides: fluid-node-only rate and the share of the box the kernel visits
    assert 0.5 < j["fluid_node_fraction"] < j["active_node_fraction"] < 1.0
    assert abs(j["mlups_fluid_nodes"] - j["value"] * j["fluid_node_fraction"]) < 1e-6 * j["value"]
    assert abs(rf["frac_active_nodes"] - rf["frac"]) < 1e-9 and len(rf["kernel_build"]) == 16
    assert "checked every step" in j["config"]["workload"]
    cb = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    # every iteration is in exactly one of the two collide classes, and the schedule of section 4a of DESIGN.md was used
    k = j["kernel_ms"]
    assert k["collide_stream_alone"]["launches"] + k["collide_stream_beside"]["launches"] == 20
    assert k["collide_stream_beside"]["launches"] > 0 and k["ibm_interpolate"]["launches"] == 4


@pytest.mark.parametrize("launcher", ["self", "torchrun", "self-auto"])
def test_bench_two_ranks_share_the_gpu(gpu, launcher):
    """the N > 1 path of bench.py, rehearsed with two ranks on the one GPU of the test box: started plainly
    (`python bench.py --gpus 2`, which spawns its ranks) and the way the driver starts it (torch.distributed.run, one
    process per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).  RCCL refuses two ranks on one device,
    so the data plane is the library's TCP staging here; everything else -- native slab schedule, streams, timing,
    reduction of the result -- is the same code"""
    port = str(29700 + os.getpid() % 200)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nx", "64", "--ny", "66", "--nz", "66", "--steps", "20", "--warmup", "5"]
    if launcher != "self-auto":
        tail += ["--transport", "tcp"]     # "self-auto": no transport named -> RCCL is tried, refuses the shared device, and the library says so and stages through the host
    if launcher.startswith("self"):
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HEMOCELL_TRANSPORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["lattice"] == [128, 66, 66]
    assert j["value"] > 0 and j["config"]["cells"] > 0 and "cpu_baseline" not in j      # the CPU baseline is an N = 1 item
    assert abs(j["value"] - 128 * 66 * 66 / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]
    assert "TCP" in j["config"]["parallelism"] and j["slab_schedule"]["host_ms_per_step"] > 0
    if launcher == "self-auto":
        assert "RCCL point-to-point is not usable here" in r.stderr
    assert j["slab_schedule"]["records_sent_rank0"] > 0            # cells do sit at the slab faces in this packing


def test_bench_config3_strong_scaling_four_slabs_equal_one(gpu):
    """BASELINE config 3 in the shape BASELINE states it: the 512 x 256 x 256 pipe, RBC + PLT, cut into x-slabs (`--config c3`,
    strong scaling).  Four ranks share the one GPU of the test box (128 planes each, host-staged data plane; 8 x 64 planes is
    the same code with twice the ranks, which the box's limit of 6 processes per GPU does not allow here): cell count, owned
    vertices, fluid nodes, mass and the velocity statistics after 15 iterations are those of the one-slab run"""
    one = _run(["--config", "c3", "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--copy-reps", "2"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HEMOCELL_TRANSPORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "c3", "--transport", "tcp", "--steps", "10", "--warmup", "5",
                        "--copy-reps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    four = json.loads(lines[0])
    assert one["scaling"] == four["scaling"] == "strong" and one["n_gpus"] == 1 and four["n_gpus"] == 4
    assert one["config"]["lattice"] == four["config"]["lattice"] == [512, 256, 256]
    assert one["config"]["cells"] == four["config"]["cells"] > 4000            # ~3870 RBC + ~280 PLT
    assert one["config"]["vertices"] == four["config"]["vertices"]             # every vertex owned by exactly one slab
    assert "pltSimpleModel platelets per RBC: 0.07" in four["config"]["workload"]
    a, b = one["diagnostics"], four["diagnostics"]
    assert a["fluid_nodes"] == b["fluid_nodes"] and a["owned_vertices"] == b["owned_vertices"]
    assert a["all_nodes"] == b["all_nodes"] == 512 * 256 * 256
    assert abs(a["mass_minus_nodes"] - b["mass_minus_nodes"]) <= 1e-12 * a["all_nodes"]      # total mass to 1e-12 relative (and conserved: the sum stays ~0)
    assert abs(a["mass_minus_nodes"]) <= 1e-9 * a["all_nodes"]
    for key in ("fluid_speed_max", "fluid_speed_mean", "vertex_speed_max", "vertex_speed_mean", "rho_bar_min", "rho_bar_max"):
        assert abs(a[key] - b[key]) <= 1e-9 * abs(a[key]) + 1e-18, (key, a[key], b[key])
    assert four["slab_schedule"]["records_sent_rank0"] > 0


def test_bench_64_plane_slabs_five_ranks_equal_one(gpu):
    """the slab thickness BASELINE config 3 gives each of 8 GPUs -- 64 planes of 256 x 256 -- with real neighbours on both faces:
    five ranks of 64 planes (the box allows five rank processes beside this one) against the one-slab run of the same 320 x 256 x
    256 pipe with RBC and PLT: same cells, owned vertices, mass and velocity statistics after 15 iterations"""
    common = ["--ny", "256", "--nz", "256", "--plt-ratio", "0.07", "--steps", "10", "--warmup", "5", "--copy-reps", "2"]
    one = _run(["--nx", "320", "--no-cpu-baseline"] + common)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HEMOCELL_TRANSPORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--nx", "64", "--transport", "tcp"] + common, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    five = json.loads(lines[0])
    assert five["n_gpus"] == 5 and one["config"]["lattice"] == five["config"]["lattice"] == [320, 256, 256]
    assert one["config"]["cells"] == five["config"]["cells"] > 2000 and one["config"]["vertices"] == five["config"]["vertices"]
    a, b = one["diagnostics"], five["diagnostics"]
    assert a["fluid_nodes"] == b["fluid_nodes"] and a["owned_vertices"] == b["owned_vertices"] and a["all_nodes"] == b["all_nodes"]
    assert abs(a["mass_minus_nodes"] - b["mass_minus_nodes"]) <= 1e-12 * a["all_nodes"]
    for key in ("fluid_speed_max", "fluid_speed_mean", "vertex_speed_max", "vertex_speed_mean", "rho_bar_min", "rho_bar_max"):
        assert abs(a[key] - b[key]) <= 1e-9 * abs(a[key]) + 1e-18, (key, a[key], b[key])
    assert five["slab_schedule"]["records_sent_rank0"] > 100          # at 64 planes most cells of a slab sit within the envelope of a face
