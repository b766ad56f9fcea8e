"""CPU suite: the N>1 path of bench.py (one process per GPU, barrier, max-over-ranks timing,
whole-job aggregate, rank 0 prints one JSON line) under gloo with world_size 2.  The path has no
data-path collective -- replicas only -- so the harness self-test stand-in (sleeps, no transform)
exercises everything that differs from N=1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, extra, port, gpus=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(gpus or nproc)] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1")
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300, cwd=ROOT)


def test_world_size_2_gloo():
    r = _launch(2, ["--steps", "3", "--warmup", "1", "--log2n", "20", "--selftest-sleep-ms", "20", "--inverse-steps", "1"], 29577)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["higher_is_better"] is True and j["vs_baseline"] is None and j["unit"] == "MB/s"
    # slowest rank sleeps 2 x 20 ms per step: the MAX over ranks sets the time, the value is the aggregate of both
    assert 40.0 <= j["ms_per_step"] < 80.0
    expect = 2 * (1 << 20) / 1e6 / (j["ms_per_step"] / 1e3)
    assert abs(j["value"] - expect) / expect < 0.01
    assert j["roundtrip_exact"] is True and "cpu_baseline" not in j
    # the config-5 leg (files in tmpfs -> batch entry point -> host buffers): 4 items per rank, the slowest rank sleeps 2 x 20 ms per item;
    # aggregate = all ranks' bytes over the max wall time
    c5 = j["config5_host_path"]
    assert c5["items_per_gpu"] == 4 and 160.0 <= c5["wall_ms"] < 400.0
    expect5 = 2 * 4 * c5["bytes_each"] / 1e6 / (c5["wall_ms"] / 1e3)
    assert abs(c5["aggregate_MBps"] - expect5) / expect5 < 0.02


def test_self_launch_without_outer_torchrun():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver may run it): bench.py starts its two ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2n", "20",
                        "--selftest-sleep-ms", "15", "--inverse-steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and 30.0 <= j["ms_per_step"] < 70.0


def test_gpus_must_match_world_size():
    r = _launch(2, ["--steps", "1", "--selftest-sleep-ms", "5"], 29579, gpus=3)
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stderr


def test_single_process_selftest():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--log2n", "16",
                        "--selftest-sleep-ms", "5", "--inverse-steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and 5.0 <= j["ms_per_step"] < 30.0
