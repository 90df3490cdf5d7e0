"""`python bench.py --gpus N` must start its N ranks by itself (one process per GPU), and behave the same under an external
launcher (torchrun: WORLD_SIZE in the environment).  On this box there is no GPU, so the decode is left out
(`--launcher-selftest`, gloo): what runs is bench.py's own launcher, rank plan (`_rank_share`), barrier, MAX / SUM reductions,
per-rank gather and imbalance figure -- the code the N-GPU bench line goes through -- with made-up step times of (1 + rank) ms.
No product decode path is involved and nothing here touches the oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out            # exactly ONE JSON line (rank 0's)
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_by_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--steps", "4", "--frames", "300"],
                       env=_clean_env(), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["selftest"] is True and j["value"] is None
    cfg = j["config"]
    assert cfg["process_group"]["world_size"] == 2
    assert "self-spawned" in cfg["launcher"]
    assert [p["range"] for p in cfg["per_rank"]] == [[0, 300], [300, 600]]          # contiguous ranges of ONE index, weak scaling
    assert cfg["samples_per_step"] == 2 * 300 * 2 * 4096                             # SUM over ranks
    assert j["ms_per_step"] == 2.0                                                   # MAX over ranks of (1 + rank) ms
    assert cfg["shard"]["imbalance"] == 0.0 and j["scaling"] == "weak"


def test_gpus_1_and_world_size_1_agree():
    a = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launcher-selftest", "--steps", "4", "--frames", "300"],
                       env=_clean_env(), capture_output=True, timeout=600)
    env = _clean_env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    b = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launcher-selftest", "--steps", "4", "--frames", "300"],
                       env=env, capture_output=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0
    ja, jb = _line(a.stdout), _line(b.stdout)
    assert ja["n_gpus"] == jb["n_gpus"] == 1
    assert ja["config"]["samples_per_step"] == jb["config"]["samples_per_step"] and ja["ms_per_step"] == jb["ms_per_step"]


def test_external_launcher_environment_is_respected():
    """Under torchrun the ranks exist already: bench.py must not spawn again (each of the two processes below is one rank)."""
    from test_shard_gloo import _free_port
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = _clean_env()
        env.update({"WORLD_SIZE": "2", "RANK": str(rank), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--steps", "4", "--workload", "config5",
                                       "--total-frames", "3000", "--unique", "48"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1].decode()[-2000:]
    j = _line(outs[0][0])
    assert not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")]     # rank 1 prints nothing
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "external" in j["config"]["launcher"]
    (a0, a1), (b0, b1) = [p["range"] for p in j["config"]["per_rank"]]
    assert a0 == 0 and a1 == b0 and b1 == 3000                                      # strong scaling: the 3000 frames are cut in two
    assert j["config"]["shard"]["imbalance"] < 0.02                                  # ... by algorithmic bytes


def test_devices_option_maps_ranks_to_devices():
    """`--gpus 2 --devices 0,0` (round 4: two ranks on ONE GPU, the rehearsal of the N-rank line where no second GPU exists): the option
    travels to the self-spawned ranks and maps local rank r to devices[r]; without it rank r runs on device r."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--devices", "0,0", "--launcher-selftest", "--steps", "4", "--frames", "300"],
                       env=_clean_env(), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    cfg = _line(r.stdout)["config"]
    assert cfg["devices"] == "0,0" and cfg["device_of_rank"] == [0, 0]
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--steps", "4", "--frames", "300"],
                       env=_clean_env(), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    cfg = _line(r.stdout)["config"]
    assert cfg["devices"] is None and cfg["device_of_rank"] == [0, 1]
    sys.path.insert(0, ROOT)
    import bench
    assert [bench._device_of("3,1,2", r) for r in range(4)] == [3, 1, 2, 3] and bench._device_of("", 5) == 5
    assert bench._launch_sizes(20, 12) == [12, 8] and bench._launch_sizes(24, 12) == [12, 12] and bench._launch_sizes(5, 1) == [1] * 5


def test_a_process_group_of_one_rank_reduces_like_none():
    """`bench.py --process-group` (round 5): the MAX / SUM / all_gather reductions through a process group of ONE rank give what
    the single-process path computes without a group (gloo here; on the GPU box the same flag runs them through RCCL)."""
    import torch.distributed as dist
    from test_shard_gloo import _free_port
    sys.path.insert(0, ROOT)
    import bench
    from claxon_amd import shard
    plain = shard.reduce_job(None, 0.25, 1234, 0)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert shard.reduce_job(dist, 0.25, 1234, 0) == plain == (0.25, 1234, 0)
        assert bench._gather_floats(dist, [1.5, 2.0], 1) == bench._gather_floats(None, [1.5, 2.0], 1) == [[1.5, 2.0]]
    finally:
        dist.destroy_process_group()
