"""The RCCL code path on the one GPU this box has: a communicator of one rank.  Every solve must leave the bits the
communicator-free process leaves (a fold over one rank adds 0.0 + v)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run(kind):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_worker.py"), kind], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_one_rank_communicator_changes_no_bit():
    plain, rccl = run("plain"), run("rccl")
    assert plain.keys() == rccl.keys() and len(plain) == 12
    for key in plain:
        assert plain[key] == rccl[key], key
        assert plain[key]["status"] == 0
    # and the device-driven loops equal the host-scalar loops in both processes
    for key in plain:
        if key.startswith("0:"):
            assert plain[key] == plain["1:" + key[2:]], key
            assert rccl[key] == rccl["1:" + key[2:]], key
