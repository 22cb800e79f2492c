"""Data-parallel training on the HIP backward: two ranks (sharing the one GPU of the test box, gloo rendezvous on
127.0.0.1) run harness.fit on different synthetic batches; after every step their parameters must be identical
and equal to a single process that averages the two ranks' gradients itself."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from .conftest import REPO

pytestmark = pytest.mark.gpu

WORKER = r'''
import importlib, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["AHV_REPO"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ahv = importlib.import_module("3dahv_amd")
dev = torch.device("cuda:0")
cfg = {"RUN_NAME": "t", "DATA": {"NUM_ROTA": 32, "BG": True, "SIZE_THR": 25, "OBJ_SIZE": 256, "ACC_THR": 30, "VIEW_THR": 90},
       "TRAIN": {"MASK": False, "MASK_RATIO": 0.0, "LR": 1e-4}}
torch.manual_seed(0)
m = ahv.estimator.EstimatorCo3d(cfg, feature_extractor=ahv.estimator.PatchifyBackbone(seed=1)).to(dev)
torch.manual_seed(100 + rank)     # different rotation draws per rank from here on
loader = ahv.harness.SyntheticTrainingPairs(batch_size=1, steps=2, seed=7 + rank)
losses = ahv.harness.fit(cfg, m, loader, device=dev)
flat = torch.cat([p.detach().flatten() for p in m.feature_aligner.feature_embedding_2d.parameters()]).cpu()
both = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "ranks diverged"
assert all(l == l for l in losses)
if rank == 0:
    print("OK", losses)
dist.destroy_process_group()
'''


def test_two_rank_fit_keeps_replicas_identical(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AHV_REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]
