"""MI355X-native drop-in for the reference train-step orchestrator `model/FR_PartialFC.py`.

Kept from /root/reference/model/FR_PartialFC.py: `Model(conf, logger, stage)` with `.encoder`, `.loss`, `.opt`,
`.sch`, `.forward(x)` (:154-156), `.training_step(batch) -> {'loss': np.ndarray}` (:162-193),
`.configure_optimizers()` (:434-474, ONE optimizer over param groups [encoder, head], head group last),
`.training_epoch_end`, `.validation_step/.test_step(+_epoch_end)` (:196-373) and `.cross_test_step/_epoch_end` (:379-427).  The step composition is the reference's:
    opt.zero_grad -> encoder.train() -> feat = normalize(encoder(img)) -> loss = head(feat, id, opt)
    -> loss.backward() -> clip_grad_norm_(encoder, 5) -> opt.step()
but encoder, normalize and head are the libfrhip kernels (nets.resnet / nets.PartialFC of this package).
`conf.mixed_precision` selects bf16 MFMA compute (the reference's fp16 autocast + GradScaler has no role with
bf16: no loss scaling is needed, so the GradScaler branch collapses into the plain one).
"""
import importlib

import numpy as np
import torch
from torch import nn
from torch.nn.parallel import DistributedDataParallel as DDP

import nets._backbone as _bb


class _NormalizeFn(torch.autograd.Function):
    """F.normalize(x) rows (model/FR_PartialFC.py:171) on the HIP kernels."""

    @staticmethod
    def forward(ctx, x):
        from frhip import ops
        xh, nrm = ops.l2norm_rows(x.contiguous().float(), torch.float32)
        ctx.save_for_backward(xh, nrm)
        return xh

    @staticmethod
    def backward(ctx, g):
        from frhip import ops
        xh, nrm = ctx.saved_tensors
        return ops.l2norm_bwd(g.contiguous().float(), xh, nrm)


def normalize(x):
    return _NormalizeFn.apply(x)


class Model(nn.Module):
    def __init__(self, conf, logger=None, stage="train"):
        super().__init__()
        self.conf = conf
        self.logger_ = logger
        self.epoch = 0
        self.lr = conf.lr
        self.sync_loss = True          # False: training_step returns the device tensor (no D2H sync per step)
        self._graph = None             # set by capture_training_step()
        if "ResNet" in conf.network:
            self.encoder = importlib.import_module("nets.resnet").Encoder(conf=conf)
        elif "AlterNet" in conf.network:
            self.encoder = importlib.import_module("nets.AlterNet_SwinV2_FAN").Encoder(conf=conf)
        elif "Swin" in conf.network:
            self.encoder = importlib.import_module("nets.SwinV2").Encoder(conf=conf)
        else:
            raise NotImplementedError("frhip: backbone %r is not built yet (SURVEY.md section 8f)" % conf.network)
        if self.encoder is None:
            raise ValueError("unknown network %r" % conf.network)
        self.encoder = self.encoder.to(conf.local_rank)
        ckpt = getattr(conf, "ckpt_path", None)
        if ckpt is not None:
            pre = torch.load(ckpt, map_location="cpu")["model_state_dict"]
            self.encoder.load_state_dict({k[7:]: v for k, v in pre.items()}, strict=True)   # strip 'module.'
        if stage == "train":
            if getattr(conf, "world_size", 1) > 1 or getattr(conf, "force_ddp", False):
                # where the reference has DistributedDataParallel(encoder): same surface ('module.' keys, rank-0 broadcast at
                # construction), but the backbone averages its flat gradient arena in place with RCCL during backward
                # (nets._backbone.DataParallel) instead of copying every gradient through DDP's buckets
                if hasattr(self.encoder, "_backward_impl"):
                    self.encoder = importlib.import_module("nets._backbone").DataParallel(self.encoder)
                else:
                    self.encoder = DDP(self.encoder, broadcast_buffers=False, device_ids=[conf.local_rank])
            head_mod = importlib.import_module("nets.%s" % getattr(conf, "loss", "PartialFC"))
            if conf.optimizer == "SGD":
                self.loss = head_mod.PartialFC(conf=conf, num_classes=conf.n_classes)
            elif conf.optimizer == "AdamW":
                self.loss = head_mod.PartialFCAdamW(conf=conf, num_classes=conf.n_classes)
            else:
                raise ValueError(conf.optimizer)
            self.loss.train().to(conf.local_rank)
            self.opt, self.sch = self.configure_optimizers()
            # the side stream of the backward pass is chosen by a timed probe (nets._backbone.side_stream: HIP shares a few hardware
            # queues among all streams; GPU_MAX_HW_QUEUES, default 4, raises their number and must be set before the runtime
            # initialises): probe here, at construction, instead of inside the first backward pass
            if next(self.encoder.parameters()).is_cuda:
                with torch.cuda.device(next(self.encoder.parameters()).device):
                    importlib.import_module("nets._backbone").side_stream(next(self.encoder.parameters()).device)

    def forward(self, x):
        return self.encoder(x)

    def _step(self, img, id_):
        """the reference's step body (model/FR_PartialFC.py:167-188, non-GradScaler branch)"""
        self.opt.zero_grad()
        self.encoder.train()
        if hasattr(self.loss, "prepare"):          # label all-gather + the one sync sampling needs, before the GPU gets busy
            self.loss.prepare(id_, self.opt)
        feat = normalize(self.forward(img))
        self.loss.train()
        loss = self.loss(feat, id_, self.opt)
        if hasattr(self.opt, "step_group_early") and hasattr(self.loss, "arm_early_update"):
            # the clip below covers the ENCODER only (reference :181) and step() follows this one backward(): the head's parameter
            # group may be updated as soon as its gradient exists, beside the backbone's backward pass
            self.loss.arm_early_update(self.opt)
        loss.backward()
        # an early head update the backbone's backward pass did not get to launch (the hook fired behind it): launch it now, so that
        # nothing parked survives into the next step
        _bb.run_deferred_side()
        if hasattr(self.opt, "last_grad_norm"):          # frhip.optim.SGD: the clip rides inside the fused update
            self.opt.step(clip=(self.encoder.parameters(), 5))
        else:
            torch.nn.utils.clip_grad_norm_(self.encoder.parameters(), 5)
            self.opt.step()
        return loss.detach()

    def training_step(self, batch):
        img, id_ = batch
        img, id_ = img.to(self.conf.local_rank), id_.to(self.conf.local_rank)
        if self._graph is not None:
            self._g_img.copy_(img, non_blocking=True)
            self._g_id.copy_(id_.view(self._g_id.shape), non_blocking=True)
            self._graph.replay()
            loss = self._g_loss
        else:
            loss = self._step(img, id_)
        if self.sync_loss:
            return {"loss": loss.cpu().detach().numpy()}
        return {"loss": loss}

    def capture_training_step(self, batch, warmup=3):
        """Record one whole optimisation step (some 1.1 K kernel launches) into a HIP graph and replay it from then
        on: the step becomes ONE host call, so the MI355X is never waiting for Python.  Valid while the batch shape
        stays fixed and the head does no host-side sampling (sample_rate == 1; PartialFC's sampled variant draws
        from the CPU generator every step, nets/PartialFC.py:110, which a graph cannot replay)."""
        if getattr(self.conf, "sample_rate", 1.0) < 1 or getattr(self.conf, "world_size", 1) > 1:
            raise RuntimeError("graph capture needs sample_rate == 1 and world_size == 1")
        img, id_ = batch
        self._g_img = img.to(self.conf.local_rank).clone()
        self._g_id = id_.to(self.conf.local_rank).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step(self._g_img, self._g_id.clone())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            self._g_loss = self._step(self._g_img, self._g_id.clone())
        self._graph = graph

    # ---- evaluation (reference :196-270, :345-373): pairs [b,2,c,h,w] -> eval-mode embeddings -> verification metrics
    def _shared_eval_step(self, batch, dataset_name, prefix):
        pair, label = batch
        pair, label = pair.to(self.conf.local_rank), label.to(self.conf.local_rank)
        pair = pair.reshape(-1, *pair.shape[2:])                     # 'b p c h w -> (b p) c h w'
        self.encoder.eval()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        with torch.no_grad():
            embedding = normalize(self.forward(pair))
        end.record()
        torch.cuda.synchronize()
        return {f"{dataset_name}_embedding_1": embedding[0::2].cpu().numpy(),
                f"{dataset_name}_embedding_2": embedding[1::2].cpu().numpy(),
                f"{dataset_name}_infer_time": start.elapsed_time(end),
                f"{dataset_name}_label_list": label.cpu().numpy(), "dataset_name": dataset_name}

    def validation_step(self, batch, dataset_idx):
        return self._shared_eval_step(batch, self.conf.val_dataset[dataset_idx], "val")

    def test_step(self, batch, dataset_idx):
        return self._shared_eval_step(batch, self.conf.test_dataset[dataset_idx], "test")

    def _eval_epoch_end(self, outputs):
        ev = importlib.import_module("utils.eval")
        name = outputs[0]["dataset_name"]
        labels = np.concatenate([o[f"{name}_label_list"] for o in outputs]).reshape(-1)
        e1 = np.concatenate([o[f"{name}_embedding_1"] for o in outputs])
        e2 = np.concatenate([o[f"{name}_embedding_2"] for o in outputs])
        hg, hi, scores = ev.pair_score(e1, e2, labels)
        roc, eer_th = ev.performance_roc(hg, hi, min_level=getattr(self.conf, "min_level", 3),
                                         max_level=getattr(self.conf, "max_level", 9))
        acc = ev.performance_acc(scores, labels, eer_th)
        return {"dataset_name": name, "acc": acc, "roc": roc, "eer_th": eer_th,
                "infer_time": float(np.mean([o[f"{name}_infer_time"] for o in outputs]))}

    validation_epoch_end = _eval_epoch_end
    test_epoch_end = _eval_epoch_end

    # ---- cross-matching test (reference :379-427): one embedding per image, every pair scored
    def cross_test_step(self, batch, dataset_idx):
        import time
        dataset_name = self.conf.cross_test_dataset[dataset_idx]
        img, label = batch
        img, label = img.to(self.conf.local_rank), label.to(self.conf.local_rank)
        start = time.time()
        self.encoder.eval()
        with torch.no_grad():
            embedding = normalize(self.forward(img))
        torch.cuda.synchronize()
        infer_time = time.time() - start
        return {f"{dataset_name}_embedding": embedding.cpu(), f"{dataset_name}_infer_time": infer_time,
                f"{dataset_name}_label_list": label.cpu(), "dataset_name": dataset_name}

    def cross_test_epoch_end(self, outputs):
        ev = importlib.import_module("utils.eval")
        name = outputs[0]["dataset_name"]
        labels = np.concatenate([np.asarray(o[f"{name}_label_list"]).reshape(-1) for o in outputs])
        embeds = np.concatenate([np.asarray(o[f"{name}_embedding"]) for o in outputs])
        hg, hi, scores, pair_labels = ev.cross_score(embeds, labels)
        roc, eer_th = ev.performance_roc(hg, hi, min_level=getattr(self.conf, "min_level", 3),
                                         max_level=getattr(self.conf, "max_level", 9))
        acc = ev.performance_acc(scores, pair_labels, eer_th)
        return {"dataset_name": name, "acc": acc, "roc": roc, "eer_th": eer_th,
                "infer_time": float(np.mean([o[f"{name}_infer_time"] for o in outputs]))}

    def training_epoch_end(self, outputs, t=None):
        self.sch.step() if self.sch is not None else None
        self.epoch += 1
        losses = [float(np.asarray(o["loss"].cpu() if torch.is_tensor(o["loss"]) else o["loss"])) for o in outputs]
        return {"lr": self.opt.param_groups[0]["lr"], "train_loss": float(np.mean(losses)) if losses else None,
                "val_acc": None}

    def configure_optimizers(self):
        c = self.conf
        groups = [{"params": self.encoder.parameters()}, {"params": self.loss.parameters()}]
        if c.optimizer == "AdamW":
            opt = importlib.import_module("frhip.optim").AdamW(groups, lr=self.lr, weight_decay=c.wd, eps=c.eps, betas=c.betas)
        elif c.optimizer == "SGD":
            # torch.optim.SGD subclass whose step() runs as three libfrhip kernels (same state / param_groups)
            opt = importlib.import_module("frhip.optim").SGD(groups, lr=self.lr, momentum=c.mom, weight_decay=c.wd)
        sch = None
        name = getattr(c, "lr_scheduler", None)
        if name == "CosineAnnealingWarmupRestarts":
            sch = importlib.import_module("utils.scheduler").CosineAnnealingWarmupRestarts(
                opt, first_cycle_steps=c.num_epoch, warmup_steps=c.warmup_steps, min_lr=c.min_lr, max_lr=self.lr)
        elif name == "MultiStep":
            sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=c.lr_decay_epoch, gamma=c.lr_decay_ratio)
        elif name == "StepLR":
            sch = torch.optim.lr_scheduler.StepLR(opt, step_size=c.lr_decay_epoch_size, gamma=c.lr_decay_ratio)
        return opt, sch
