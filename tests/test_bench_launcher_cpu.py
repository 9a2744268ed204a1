"""CPU: `python bench.py --gpus N` with no launcher around it becomes one (bench.launch_ranks): N rank processes with the
torch.distributed.run environment, rank 0's JSON line relayed, failures propagated.  The children here are stand-ins (no GPU in
this container); the real children are rehearsed on the GPU box with two gloo ranks on one device (tests/test_dp_gpu.py)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys
keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
rec = {k: os.environ.get(k) for k in keys}
rec["argv"] = sys.argv[1:]
open(os.path.join(os.environ["OUT_DIR"], "rank%s.json" % rec["RANK"]), "w").write(json.dumps(rec))
if rec["RANK"] == "0":
    print(json.dumps({"metric": "stand-in", "n_gpus": int(rec["WORLD_SIZE"])}), flush=True)
else:
    print("chatter from a non-zero rank", flush=True)   # must not reach the parent's stdout
'''


def test_launcher_starts_n_ranks_and_relays_rank0(tmp_path, capfd):
    import bench
    env = dict(os.environ, OUT_DIR=str(tmp_path))
    env.pop("MASTER_PORT", None)
    rc = bench.launch_ranks(3, [], child=[sys.executable, "-c", CHILD, "--steps", "5"], env=env)
    assert rc == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "stand-in", "n_gpus": 3}
    recs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and recs[0]["MASTER_PORT"].isdigit()
    assert {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}
    assert recs[0]["argv"] == ["--steps", "5"]


def test_launcher_propagates_a_dead_rank_and_stops_the_others(tmp_path):
    import time
    import bench
    child = ("import os, sys, time\n"
             "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
             "time.sleep(120)\n")
    t0 = time.time()
    rc = bench.launch_ranks(2, [], child=[sys.executable, "-c", child])
    assert rc == 7 and time.time() - t0 < 60, "the surviving rank was left waiting"


def test_launcher_time_limit_terminates_stuck_ranks_and_fails(tmp_path, capfd):
    """a rank that never finishes (stuck in a collective / a capture: nothing inside the rank can catch that) must not hold the job"""
    import time
    import bench
    child = ("import os, time\n"
             "open(os.path.join(os.environ['OUT_DIR'], 'pid%s' % os.environ['RANK']), 'w').write(str(os.getpid()))\n"
             "time.sleep(300)\n")
    t0 = time.time()
    rc = bench.launch_ranks(2, [], child=[sys.executable, "-c", child], env=dict(os.environ, OUT_DIR=str(tmp_path)), time_limit=3.0)
    took = time.time() - t0
    assert rc == 124 and 2.5 < took < 30, (rc, took)
    assert "time limit" in capfd.readouterr().err
    for r in range(2):  # the children are gone (terminated by PID)
        pid = int((tmp_path / f"pid{r}").read_text())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_graph_queue_setting_is_clamped_to_the_hardware_queue_count():
    """DEBUG_HIP_FORCE_GRAPH_QUEUES above GPU_MAX_HW_QUEUES (default 4) is what segfaulted hipGraphLaunch in round 3: never let it through"""
    import hidvae_amd
    f = hidvae_amd._graph_queues
    assert f({}) == ("3", [])
    assert f({"HIDVAE_GRAPH_QUEUES": "0"}) == (None, [])
    assert f({"HIDVAE_GRAPH_QUEUES": "4"}) == ("4", [])
    q, notes = f({"HIDVAE_GRAPH_QUEUES": "12"})
    assert q == "4" and len(notes) == 1 and "clamped" in notes[0]
    q, notes = f({"DEBUG_HIP_FORCE_GRAPH_QUEUES": "6"})
    assert q == "4" and "clamped" in notes[0]
    assert f({"DEBUG_HIP_FORCE_GRAPH_QUEUES": "8", "GPU_MAX_HW_QUEUES": "8"}) == ("8", [])
    q, notes = f({"HIDVAE_GRAPH_QUEUES": "0", "DEBUG_HIP_FORCE_GRAPH_QUEUES": "5"})   # an explicit runtime setting is clamped too
    assert q == "4" and notes
    q, notes = f({"HIDVAE_GRAPH_QUEUES": "many"})
    assert q == "3" and notes
    assert f({"DEBUG_HIP_FORCE_GRAPH_QUEUES": "2", "HIDVAE_GRAPH_QUEUES": "4"}) == ("2", [])   # the runtime's own variable wins
    assert hidvae_amd.GRAPH_QUEUES == os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")


def test_main_becomes_the_launcher_only_without_a_world_in_the_environment(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, **kw: calls.append((n, list(argv))) or 0)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"])]
    # too few devices for one-rank-per-device RCCL: refuse before spawning anything
    calls.clear()
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2 and calls == []
