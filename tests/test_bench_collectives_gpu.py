"""The N > 1 code path of bench.py on the one GPU a test box has (VERDICT r3 item 6 i).

`bench.py --force-collectives` runs everything an N-rank launch runs -- RCCL process group, replay gather in flight during the
next step, learner-side ingest of the gathered blocks inside the timed region, barrier + synchronize on both sides, max over
ranks -- with WORLD_SIZE = 1.  Each bench is a CHILD process started before it has touched the GPU (a process that has
initialised HIP must neither fork-and-use nor exec: the children are plain `subprocess` starts of a fresh interpreter).
The collectives' run must carry `learner_ingest` and land within a few per cent of the plain single-rank run: what the
distributed plumbing costs when there is nobody to talk to.  No multi-GPU claim is made from this."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(extra_args, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env["SGRL_BENCH_NO_CHILD"] = "1"            # no exact-f32 grand-child: one process per measurement
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "10", "--warmup", "3", "--regions", "3", "--preroll", "60",
           "--no-cpu-baseline"] + extra_args
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_forced_collectives_run_matches_the_plain_run():
    plain = _bench([], {})
    coll = _bench(["--force-collectives"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                            "MASTER_PORT": str(_free_port()), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert plain["n_gpus"] == coll["n_gpus"] == 1
    assert "learner_ingest" in coll and coll["learner_ingest"]["in_timed_region"] is True
    assert coll["learner_ingest"]["blocks_per_step_on_rank0"] == 1
    assert "torch.distributed.gather" in coll["config"]["replay_gather"]
    assert "learner_ingest" not in plain
    # the gather, the ingest and the barriers of a one-rank world cost a few per cent of a 4.8 ms step (measured: 1-3 %)
    ratio = coll["value"] / plain["value"]
    print("forced-collectives / plain env-steps per second: %.3f (%.0f vs %.0f)" % (ratio, coll["value"], plain["value"]))
    assert 0.93 < ratio < 1.05, (coll["value"], plain["value"])
    # and the ingest itself is what the run without it says it is: below 5 % of the step
    with_ing, without = coll["ms_per_step"], coll["learner_ingest"]["ms_per_step_without_ingest"]
    assert with_ing < without * 1.08, (with_ing, without)
