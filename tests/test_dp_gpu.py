"""Two real ranks (two processes on cuda:0, gloo collectives on GPU tensors) through Model.training_step: the data-parallel
backbone (nets._backbone.DataParallel: rank-0 broadcast, in-place staged all-reduce of the flat gradient arena) and the
class-sharded head with its collectives, all on the HIP kernels.  RCCL itself needs one GPU per rank: its call sequence is
rehearsed in a 1-rank group by `bench.py --dist-path`."""
import os
import sys
import tempfile
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, ws, path, rate, ret):
    for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=ws)
    from model.FR_PartialFC import Model
    conf = types.SimpleNamespace(network="ResNet18", emd_size=512, img_size=112, local_rank=0, world_size=ws,
                                 sample_rate=rate, mixed_precision=True, loss_s=30.0, loss_m=0.35, n_classes=400,
                                 optimizer="SGD", lr=0.05, wd=5e-4, mom=0.9, loss="PartialFC", lr_scheduler=None,
                                 frhip_dtype="fp32", ckpt_path=None)
    torch.manual_seed(100 + rank)                      # different initial weights per rank: the wrapper must broadcast rank 0's
    model = Model(conf, None, "train")
    g = torch.Generator().manual_seed(7 + rank)        # different data per rank
    img = torch.randn((4, 3, 112, 112), generator=g).clamp_(-1, 1).cuda()
    ids = torch.randint(0, 400, (4,), generator=g).cuda()
    losses = []
    for _ in range(2):
        losses.append(float(model.training_step((img, ids.clone()))["loss"]))
    enc = model.encoder.module
    sums = np.array([float(p.detach().double().sum()) for p in enc.parameters()])
    absmax = float(max(p.grad.abs().max() for p in enc.parameters()))
    # through a file, not a multiprocessing.Manager: a Manager is a FORKED child of the pytest process, i.e. one more process
    # holding the GPU open (a box admits 6), and it outlives a failing test for as long as pytest keeps the traceback
    np.savez(os.path.join(ret, "rank%d.npz" % rank), losses=np.array(losses), sums=sums, absmax=absmax,
             wrapped=type(model.encoder).__name__, head_rows=int(model.loss.num_local))
    dist.destroy_process_group()


@pytest.mark.parametrize("rate", [1.0, 0.5])
def test_two_ranks_stay_in_step(rate):
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(2, os.path.join(td, "pg"), rate, td), nprocs=2, join=True)
        a, b = (dict(np.load(os.path.join(td, "rank%d.npz" % r))) for r in range(2))
        assert str(a["wrapped"]) == "DataParallel" and int(a["head_rows"]) == int(b["head_rows"]) == 200
        # the global margin-softmax loss is the same number on every rank, and it moves
        np.testing.assert_allclose(a["losses"], b["losses"], rtol=1e-6)
        assert a["losses"][1] != a["losses"][0] and np.isfinite(a["losses"]).all()
        # same initial weights (broadcast) + averaged gradients every step => identical backbones after two steps,
        # although the ranks saw different images
        np.testing.assert_allclose(a["sums"], b["sums"], rtol=0, atol=0)
        assert float(a["absmax"]) > 0
