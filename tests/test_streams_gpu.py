"""The weight-gradient side stream must really run beside the main stream: HIP multiplexes streams onto a few hardware queues
(GPU_MAX_HW_QUEUES), and with RCCL's streams alive a new stream can share the main stream's queue -- the step then loses its
overlap (measured 31.3 vs 26.1 ms).  nets._backbone.side_stream probes candidates with timed spin kernels."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")) if p not in sys.path]

pytestmark = pytest.mark.gpu


def test_spin_kernel_keeps_a_queue_busy_for_the_requested_time():
    from frhip._abi import check, lib
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check(lib().frhip_spin(100, s.cuda_stream), "frhip_spin")
    torch.cuda.synchronize()
    e0.record(s)
    check(lib().frhip_spin(100000, s.cuda_stream), "frhip_spin")          # 1 ms of the 100-MHz wall clock
    e1.record(s)
    torch.cuda.synchronize()
    assert 0.9 < e0.elapsed_time(e1) < 1.5
    assert lib().frhip_spin(-1, s.cuda_stream) != 0                       # refused, no launch


def test_probe_tells_shared_queues_from_concurrent_ones():
    from nets import _backbone as bb
    main = torch.cuda.current_stream()
    assert not bb._runs_beside(main, main)                                # one stream = one queue: serialised by construction
    side = bb.side_stream(torch.device("cuda", 0))
    assert side != main and bb._runs_beside(main, side)
    assert bb.side_stream(torch.device("cuda", 0)) is side                # chosen once per device
