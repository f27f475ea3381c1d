"""MI355X-native drop-in for the reference backbone module `nets/resnet.py`.

Same surface as /root/reference/nets/resnet.py: `ResNet18/34/50/100/200(conf)`, `Encoder(conf)` (:253-316);
each returns an nn.Module whose `forward(x: float32[B,3,H,W]) -> float32[B, conf.emd_size]` and whose
state_dict has exactly the reference's keys/shapes (conv1.weight, bn1.*, layerS.B.{conv1,bn1,conv2,bn2,
downsample.0,downsample.1}.*, bn2.*, fc.*, bn3.*), so reference checkpoints load with strict=True.

What is different underneath (nothing is delegated to torch's conv/BN kernels):
  * activations are NHWC in the compute dtype (bf16, or fp32 "validation mode"); conv weights are
    channels_last Parameters, i.e. already [K][R][S][C] in memory;
  * every convolution is the hand-written MFMA implicit GEMM of libfrhip (forward, data-gradient and
    weight-gradient), BN batch statistics come out of the conv epilogue, BN-apply/ReLU/residual are fused
    element-wise passes, the stem is im2col + GEMM + fused BN-ReLU-MaxPool;
  * the whole backbone is ONE autograd node: forward keeps the activations it needs, backward runs the
    hand-written gradient kernels and returns all parameter gradients.
There is no CPU / eager fallback: without libfrhip.so or without a GPU tensor, forward raises.
"""
import torch
import torch.nn as nn

from ._backbone import (BackwardCtx, BasicBlock, DEFER_EARLY_BLOCKS, Fp8Ctx, Saved, _BN, _Conv, _Linear, basic_block_backward,
                        basic_block_forward, compute_dtype, encoder_call, prepare_conv_weights, stem_backward,
                        stem_forward, stem_reduction_operands,
                        tail_backward, tail_forward, use_fp8)

_BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 4), 50: (3, 4, 14, 4), 100: (3, 13, 30, 4), 200: (3, 43, 50, 4)}


# ------------------------------------------------------------------------------------------------- network
class ResNet(nn.Module):
    def __init__(self, block, layers, conf):
        super().__init__()
        self.emd_size = conf.emd_size
        self.dtype = compute_dtype(conf)
        self.fp8 = use_fp8(conf)                   # forward GEMMs with >= 128 input channels on the fp8 MFMA path (csrc/igemm_fp8.hip)
        self.inplanes = 64
        self.conv1 = _Conv(3, 64, 3, 1)
        self.bn1 = _BN(64)
        self.layer1 = self.stack_layers(block, 64, layers[0])
        self.layer2 = self.stack_layers(block, 128, layers[1], stride=2)
        self.layer3 = self.stack_layers(block, 256, layers[2], stride=2)
        self.layer4 = self.stack_layers(block, conf.emd_size, layers[3], stride=2)
        self.bn2 = _BN(block.expansion * conf.emd_size)
        self.fc = _Linear(block.expansion * conf.emd_size * 7 * 7, conf.emd_size)
        self.bn3 = _BN(conf.emd_size)
        # same initialisation rule as the reference (nets/resnet.py:201-209)
        for m in self.modules():
            if isinstance(m, (_Conv, _Linear)):
                nn.init.xavier_normal_(m.weight)

    def stack_layers(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_Conv(self.inplanes, planes * block.expansion, 1, stride),
                                       _BN(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _blocks(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                yield blk

    def forward(self, x):
        return encoder_call(self, x)

    # ---- the autograd node's two halves (host orchestration only; every op is a libfrhip kernel)
    def _forward_impl(self, x, training, save):
        sv = Saved() if save else None
        cur = stem_forward(self, x, training, sv)
        saved_blocks = []
        blocks = list(self._blocks())
        convs = [c for b in blocks for c in ((b.conv1, b.conv2) + ((b.downsample[0],) if b.downsample is not None else ()))]
        wprep = prepare_conv_weights(convs, self.dtype)          # every conv operand of the step in one launch
        q8 = None
        if self.fp8 and self.dtype == torch.bfloat16:
            q8 = Fp8Ctx([(c, c.physical()) for c in convs if Fp8Ctx.eligible(c.cin)])
        for blk in blocks:
            cur, s = basic_block_forward(blk, cur, self.dtype, training, save, wprep, q8)
            saved_blocks.append(s)
        emb = tail_forward(self, cur, training, sv)
        if save:
            sv.blocks = saved_blocks
        return emb, sv

    def _backward_impl(self, sv, d_emb, params):
        bc = BackwardCtx(params, d_emb.device, allreduce=getattr(self, "_frhip_allreduce", False))
        dout = tail_backward(self, sv, d_emb, bc)
        blocks = list(self._blocks())
        part = None
        for i in range(len(blocks) - 1, -1, -1):
            # the gradient leaving block i enters bn2 of block i-1: its reduction rides in block i's last kernel
            # ... and the gradient leaving block 0 enters the stem's pool / ReLU / BN (stem_reduction_operands)
            nxt = (sv.blocks[i - 1].y2, sv.blocks[i - 1].st2) if i > 0 else stem_reduction_operands(self, sv)
            res = basic_block_backward(blocks[i], sv.blocks[i], dout, self.dtype, bc, part2=part, next_bn=nxt)
            dout, part = res if nxt is not None else (res, None)
            bc.reduce_down_to(blocks[i].conv1.weight)      # data parallel: block i and everything behind it is final
            if len(blocks) - 1 - i == DEFER_EARLY_BLOCKS:
                bc.run_deferred()                          # the head's early parameter update: beside MFMA-bound blocks, not beside the tail
        stem_backward(self, sv, dout, bc, part)
        return bc.join()


# ------------------------------------------------------------------------------------------------- constructors
def ResNet18(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[18]), conf, **kwargs)


def ResNet34(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[34]), conf, **kwargs)


def ResNet50(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[50]), conf, **kwargs)


def ResNet100(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[100]), conf, **kwargs)


def ResNet200(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[200]), conf, **kwargs)


def Encoder(conf):
    """Name dispatch of the reference (nets/resnet.py:308-316) -- plus 'ResNet18', which the reference's
    dispatcher forgets although its constructor exists."""
    table = {"ResNet200": ResNet200, "ResNet100": ResNet100, "ResNet50": ResNet50, "ResNet34": ResNet34,
             "ResNet18": ResNet18}
    if conf.network in table:
        return table[conf.network](conf)
    return None
