"""Where does the bf16 step drift from the fp32-validation step?  (GPU box; diagnostic, prints a table)

Runs one ResNet50 training step of the drop-in Model twice on the same inputs -- frhip_dtype fp32 and bf16 -- and compares the
tensors that cross the stage boundaries of the forward and of the backward pass (relative l2 error and cosine)."""
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
import torch.distributed as dist

from oracle import recipe, resnet_ref

B, C = int(os.environ.get("B", "16")), 1000
NET = os.environ.get("NET", "ResNet50")


def conf(dtype):
    return types.SimpleNamespace(network=NET, emd_size=512, img_size=112, local_rank=0, world_size=1, sample_rate=1.0,
                                 mixed_precision=True, loss_s=30.0, loss_m=0.35, n_classes=C, optimizer="SGD", lr=0.1, wd=5e-4,
                                 mom=0.9, loss="PartialFC", lr_scheduler=None, frhip_dtype=dtype, ckpt_path=None)


def run(dtype, sd, W, img, ids):
    import nets._backbone as bb
    from model.FR_PartialFC import Model, normalize
    rec = {}
    orig_fwd, orig_bwd, orig_tail_b, orig_tail_f = bb.basic_block_forward, bb.basic_block_backward, bb.tail_backward, bb.tail_forward
    cnt = {"f": 0, "b": 0}

    def fwd(blk, xin, *a, **k):
        out, s = orig_fwd(blk, xin, *a, **k)
        rec["fwd.block%02d" % cnt["f"]] = out.float().cpu()
        cnt["f"] += 1
        return out, s

    def bwd(blk, s, dout, *a, **k):
        rec["bwd.into_block%02d" % (cnt["f"] - 1 - cnt["b"])] = dout.float().cpu()
        cnt["b"] += 1
        return orig_bwd(blk, s, dout, *a, **k)

    def tail_b(net, sv, d_emb, bc):
        rec["bwd.d_encoder_out"] = d_emb.float().cpu()
        return orig_tail_b(net, sv, d_emb, bc)

    def tail_f(net, cur, *a, **k):
        emb = orig_tail_f(net, cur, *a, **k)
        rec["fwd.encoder_out"] = emb.float().cpu()
        return emb

    import nets.resnet as R
    for mod in (bb, R):
        if hasattr(mod, "basic_block_forward"):
            mod.basic_block_forward, mod.basic_block_backward = fwd, bwd
        if hasattr(mod, "tail_backward"):
            mod.tail_backward, mod.tail_forward = tail_b, tail_f
    try:
        model = Model(conf(dtype), None, "train")
        model.encoder.load_state_dict(sd, strict=True)
        with torch.no_grad():
            model.loss.weight_activated.data.copy_(W.cuda())
        model.opt.zero_grad()
        model.encoder.train()
        f = normalize(model.forward(img.cuda()))
        rec["fwd.normalized"] = f.detach().float().cpu()
        loss = model.loss(f, ids.cuda(), model.opt)
        loss.backward()
        rec["loss"] = loss.detach().float().cpu().view(1)
        for k, p in model.encoder.named_parameters():
            if k.endswith("conv1.weight") or k.endswith("conv2.weight") or k in ("fc.weight",):
                rec["grad." + k] = p.grad.detach().float().cpu()
    finally:
        for mod in (bb, R):
            if hasattr(mod, "basic_block_forward"):
                mod.basic_block_forward, mod.basic_block_backward = orig_fwd, orig_bwd
            if hasattr(mod, "tail_backward"):
                mod.tail_backward, mod.tail_forward = orig_tail_b, orig_tail_f
    return rec


def main():
    dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
    torch.cuda.set_device(0)
    sd = recipe.fill_state(resnet_ref.resnet_spec(resnet_ref.BLOCKS[NET]), 9101)
    W = recipe.normal(9102, (C, 512), 0.01)
    img, ids = recipe.images(9103, B), recipe.labels(9104, B, C)
    a = run("fp32", sd, W, img, ids)
    b = run("bf16", sd, W, img, ids)
    for k in a:
        x, y = a[k].double().flatten(), b[k].double().flatten()
        rel = float((x - y).norm() / (x.norm() + 1e-300))
        cos = float((x @ y) / (x.norm() * y.norm() + 1e-300))
        print("%-34s rel %.4f  cos %.5f  |fp32| %.3e" % (k, rel, cos, float(x.norm())), flush=True)


if __name__ == "__main__":
    main()
