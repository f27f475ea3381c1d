"""Inference throughput of the encoder in eval mode (reference model/FR_PartialFC.py:205-211), folded BatchNorm against separate passes.
usage (GPU box): python tools/bench_eval.py [network] [batch]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
import nets._backbone as bb
import nets.resnet as R
name = sys.argv[1] if len(sys.argv) > 1 else "ResNet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
conf = types.SimpleNamespace(network=name, emd_size=512, frhip_dtype="bf16")
net = getattr(R, name)(conf).cuda().eval()
x = torch.randn(B, 3, 112, 112).clamp_(-1, 1).cuda()
for fold in (False, True, False, True):
    bb._EVAL_FOLD = fold
    with torch.no_grad():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            net(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s eval B=%d  folded BatchNorm %d:  %.3f ms / batch  %.0f img/s" % (name, B, fold, ms, B / ms * 1e3), flush=True)
