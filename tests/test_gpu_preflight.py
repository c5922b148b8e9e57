"""tools/multi_gpu_preflight.py under `-m gpu`: on a box with >= 2 visible devices the whole preflight must
come back ok (peer access, raw RCCL with N ranks, 200 value-checked direct all-reduces across devices, strict + weak updates
through the fused direct exchange and RCCL inside a replayed hipGraph against the step-wise path); on a one-GPU box (gpurun)
it must say so and leave -- never hang.  Reference analogue: xt/framework/trainer.py:86-92 (the dead host-side exchange)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_gpu_preflight.py")] + list(extra), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    line = [x for x in proc.stdout.decode().splitlines() if x.startswith("{")][-1]
    return proc.returncode, json.loads(line)


def test_preflight_on_a_one_gpu_box_reports_that_it_needs_two():
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU box: the real preflight runs instead")
    rc, res = _run()
    assert rc == 2 and res["ok"] is False and "needs >= 2" in res["skipped"]


def test_preflight_passes_on_every_visible_gpu():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (gpurun boxes have one)")
    rc, res = _run()
    assert rc == 0 and res["ok"], json.dumps(res)[:3000]
    assert res["world"] == torch.cuda.device_count()
    assert all(all(row) for row in res["peer_access"])
    assert res["checks"]["direct_exchange_200_value_checked"]["all_reduces"] == 200
