"""ResNet encoder of Monodepth2, declared without torchvision (absent from this image).

Reference: MD2/networks/resnet_encoder.py:62-98 wraps ``torchvision.models.resnet{18,34,50,...}`` and
returns the five feature maps after ``(x - 0.45) / 0.225``.  The module/parameter names below follow
torchvision's ResNet (conv1, bn1, layer1..4.{i}.conv{j}/bn{j}/downsample.{0,1}, fc) so reference
checkpoints (``encoder.pth``) load unchanged.  Convolutions stay on PyTorch-ROCm/MIOpen (north_star).

In eval() mode on the GPU -- every attack step, torchattacks/attack.py:165-182 -- the element-wise chains between
the convolutions (BatchNorm with running statistics, identity add, ReLU, the stem's max-pool) run as the fused K9
kernels of libdmh_hip (``ops.bn_act``, ``ops.stem_bn_relu_pool``); train() mode and CPU tensors take the module
path below, which is the reference's.  ``ResNet.fuse_eval_bn = False`` switches the fused path off.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def _conv_bn_act(conv, x, aff, residual=None):
    """relu(bn_eval(conv(x)) (+ residual)): one fused K10 launch for 3x3 stride-1 convolutions that fill the chip,
    conv + the K9 bn_act pass otherwise."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None):
        return ops.conv3x3_bn_act(x, conv.weight, aff[0], aff[1], residual, True, 1)
    return ops.bn_act(conv(x), aff[0], aff[1], residual)


def _plain3x3(conv):
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None)


def _train_fused(bn, x):
    """Train-mode BatchNorm on a CUDA tensor with the standard configuration (affine, running statistics, momentum)."""
    return (ops.WINO_ENABLED and x.is_cuda and x.dtype == torch.float32 and bn.training and bn.momentum is not None
            and bn.weight is not None and bn.track_running_stats and torch.is_grad_enabled())


def _conv(conv, x):
    """conv(x); 3x3 stride-1 pad-1 convolutions of CUDA tensors go through ops.conv3x3 (Winograd-MFMA kernel where the
    shape fills the chip, MIOpen otherwise) -- same parameters, same result within fp32 rounding."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1):
        return ops.conv3x3(x, conv.weight, conv.bias, 1)
    return conv(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def _is_down_pair(self):
        """The block opens a resolution level with the standard pair: conv1 3x3 stride 2 + a 1x1 stride-2 shortcut."""
        d = self.downsample
        if d is None:
            return False
        c1, cd = self.conv1, d[0]
        return (c1.kernel_size == (3, 3) and c1.stride == (2, 2) and c1.padding == (1, 1) and c1.dilation == (1, 1)
                and c1.groups == 1 and c1.bias is None and cd.kernel_size == (1, 1) and cd.stride == (2, 2)
                and cd.padding == (0, 0) and cd.groups == 1 and cd.bias is None)

    def _down_pair(self, x):
        """(conv1(x), downsample[0](x)) as one K15 launch where the block opens a resolution level (3x3 stride 2 + 1x1
        stride 2 on the same input: layer2.0 / layer3.0 / layer4.0); None for every other block."""
        if self._is_down_pair() and ops.down_convs_ok(x, self.conv1.weight, self.downsample[0].weight):
            return ops.down_convs(x, self.conv1.weight, self.downsample[0].weight)
        return None

    def forward(self, x):
        if _train_fused(self.bn1, x):       # train(): batch statistics by the K9 kernels, one pass for BN + add + ReLU
            pair = self._down_pair(x)
            if pair is not None:
                idt = ops.bn_act_train(self.downsample[1], pair[1], relu=False)
                out = ops.bn_act_train(self.bn1, pair[0])
            else:
                idt = x if self.downsample is None else ops.bn_act_train(self.downsample[1], self.downsample[0](x), relu=False)
                out = ops.bn_act_train(self.bn1, _conv(self.conv1, x))
            return ops.bn_act_train(self.bn2, _conv(self.conv2, out), residual=idt)
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(_conv(self.conv1, x)))
        out = self.bn2(_conv(self.conv2, out))
        return self.relu(out + idt)

    def forward_fused(self, x, aff, want_skip=False):
        """``want_skip``: x is also a pyramid feature (the previous layer's output).  Returns (y, x') then, where x' is the
        tensor the decoder should take in place of x: an alias whose gradient enters the down-sampling node (added in
        K15's epilogue instead of by autograd's accumulation pass), or x itself on the other paths."""
        if want_skip:
            if (self._is_down_pair() and _plain3x3(self.conv2) and x.requires_grad
                    and ops.down_block_eval_ok(x, self.conv1.weight, self.downsample[0].weight, self.conv2.weight)):
                return ops.down_block_eval(x, self.conv1.weight, *aff[self.bn1], self.downsample[0].weight,
                                           *aff[self.downsample[1]], self.conv2.weight, *aff[self.bn2], return_skip=True)
            return self.forward_fused(x, aff), x
        if (self.downsample is None and all(_plain3x3(c) for c in (self.conv1, self.conv2))
                and ops.basic_block_eval_ok(x, self.conv1.weight, self.conv2.weight)):
            # inside an attack: the whole block as one autograd node (masks and the identity add in K10's epilogues)
            return ops.basic_block_eval(x, self.conv1.weight, *aff[self.bn1], self.conv2.weight, *aff[self.bn2])
        if (self._is_down_pair() and _plain3x3(self.conv2)
                and ops.down_block_eval_ok(x, self.conv1.weight, self.downsample[0].weight, self.conv2.weight)):
            # inside an attack: the whole down-sampling block as one node (BatchNorms and ReLU in the K15 / K10 epilogues)
            return ops.down_block_eval(x, self.conv1.weight, *aff[self.bn1], self.downsample[0].weight,
                                       *aff[self.downsample[1]], self.conv2.weight, *aff[self.bn2])
        pair = self._down_pair(x)
        if pair is not None:
            idt = ops.bn_act(pair[1], *aff[self.downsample[1]], relu=False)
            out = ops.bn_act(pair[0], *aff[self.bn1])
        else:
            idt = x if self.downsample is None else ops.bn_act(self.downsample[0](x), *aff[self.downsample[1]], relu=False)
            out = _conv_bn_act(self.conv1, x, aff[self.bn1])
        return _conv_bn_act(self.conv2, out, aff[self.bn2], residual=idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(_conv(self.conv2, out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)

    def forward_fused(self, x, aff):
        idt = x if self.downsample is None else ops.bn_act(self.downsample[0](x), *aff[self.downsample[1]], relu=False)
        out = ops.bn_act(self.conv1(x), *aff[self.bn1])
        out = _conv_bn_act(self.conv2, out, aff[self.bn2])
        return ops.bn_act(self.conv3(out), *aff[self.bn3], residual=idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_input_images=1, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)  # unused by the encoder; kept for checkpoint keys
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

        self.fuse_eval_bn = True
        self._bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        self._eps = None

    def eval_affine(self):
        """{bn: (scale, shift)} of every BatchNorm2d from its running statistics, recomputed on each forward in one
        batched pass over all layers (9 small launches instead of 4 per layer); inside ops.frozen_weights() (an
        attack: parameters and statistics are constants) once per scope."""
        return ops.frozen_memo(("eval_affine", id(self)), self._eval_affine)

    def _eval_affine(self):
        with torch.no_grad():
            w, b, mu, var = (torch.cat([getattr(bn, n).detach().float() for bn in self._bns])
                             for n in ("weight", "bias", "running_mean", "running_var"))
            sizes = [bn.num_features for bn in self._bns]
            if self._eps is None or self._eps.device != w.device:
                self._eps = torch.cat([torch.full((n,), bn.eps, dtype=torch.float32) for bn, n in zip(self._bns, sizes)]
                                      ).to(w.device)
            scale = w * torch.rsqrt(var + self._eps)
            shift = b - mu * scale
            return {bn: (sc, sh) for bn, sc, sh in zip(self._bns, scale.split(sizes), shift.split(sizes))}

    def bn_params_constant(self):
        """The fused eval path folds every BatchNorm into a detached (scale, shift): only valid when no gradient can be
        owed to a BatchNorm weight / bias -- inside ops.frozen_weights() (an attack), under torch.no_grad(), or when no
        affine parameter requires grad.  Fine-tuning with frozen statistics (eval() + grad on) must take the module
        path, where nn.BatchNorm2d propagates to weight and bias as the reference's does."""
        if ops.weights_frozen() or not torch.is_grad_enabled():
            return True
        return not any(p is not None and p.requires_grad for bn in self._bns for p in (bn.weight, bn.bias))

    def fused_eval_ok(self, x):
        return (self.fuse_eval_bn and x.is_cuda and x.dtype == torch.float32 and not any(bn.training for bn in self._bns)
                and self.bn_params_constant())

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)


_CONFIGS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
            101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class ResnetEncoder(nn.Module):
    """Five-level feature pyramid [64,64,128,256,512] (x4 above ResNet-34)."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        if num_layers not in _CONFIGS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained:
            # MD2/networks/resnet_encoder.py:52-56 downloads ImageNet weights; there is no network here.
            print("ResnetEncoder: ImageNet weights are not available offline -- using random initialisation")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        block, layers = _CONFIGS[num_layers]
        self.encoder = ResNet(block, layers, num_input_images)
        self.roi_backward = True    # forward(x, roi=...) may run the head's backward on the attack's windows (ops.encoder_head_eval)
        self.roi_incremental = os.environ.get("DMH_ROI_INCREMENTAL", "1") != "0"     # ... and, given the clean frames, its forward
        self.roi_incremental_layer2 = os.environ.get("DMH_ROI_LAYER2", "1") != "0"   # ... and layer2's forward and backward
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def _k10_convs(self):
        """[(conv, bn)] of the 3x3 stride-1 convolutions of the residual blocks: the filters K10 may be asked for."""
        if getattr(self, "_k10_list", None) is None:
            out = []
            for layer in (self.encoder.layer1, self.encoder.layer2, self.encoder.layer3, self.encoder.layer4):
                for blk in layer:
                    for conv, bn in ((getattr(blk, "conv1", None), getattr(blk, "bn1", None)),
                                     (getattr(blk, "conv2", None), getattr(blk, "bn2", None))):
                        if conv is not None and conv.kernel_size == (3, 3) and _plain3x3(conv):
                            out.append((conv, bn))
            self._k10_list = out
        return self._k10_list

    def _prefetch_filters(self, aff=None):
        """All K10 filters of this encoder in one launch (ops.wino_prefetch): with ``aff`` (an attack: inside
        ops.frozen_weights(), once per scope) the forms with the eval-mode BatchNorm scale folded in, otherwise (the train pass,
        every forward) the plain forms; forward and -- when a backward can follow -- backward-data."""
        dirs = (False, True) if torch.is_grad_enabled() else (False,)
        if aff is not None:
            ops.frozen_memo(("prefetch", id(self)), lambda: ops.wino_prefetch(
                [(c.weight, bw, aff[bn][0]) for c, bn in self._k10_convs() for bw in dirs]) or True)
        else:
            ops.wino_prefetch([(c.weight, bw, None) for c, _ in self._k10_convs() for bw in dirs], fresh=True)

    def _stem(self, input_image):
        """conv1((input_image - 0.45) / 0.225): one K14 launch for the standard first layer on the GPU, the reference's
        two steps otherwise."""
        c = self.encoder.conv1
        if (ops.WINO_ENABLED and input_image.is_cuda and input_image.dtype == torch.float32 and input_image.dim() == 4
                and input_image.shape[1] == 3 and tuple(c.weight.shape) == (64, 3, 7, 7) and c.stride == (2, 2)
                and c.padding == (3, 3) and c.dilation == (1, 1) and c.groups == 1 and c.bias is None
                and input_image.shape[2] % 2 == 0 and input_image.shape[3] % 2 == 0):
            return ops.stem_conv_norm(input_image, c.weight, 0.45, 0.225)
        return self._conv1((input_image - 0.45) / 0.225)

    def forward(self, input_image, roi=None, clean=None):
        """``roi`` = (roi.RoiPlan, its device table): the caller (an object attack, DepthModelWrapper.masked_sq_mean) reads
        d / d input_image inside the plan's image window only, so the head's backward may run on the plan's windows.
        ``clean`` [B,3,H,W]: the same frames without the pasted object -- input_image equals it outside the plan's boxes --
        so the head's FORWARD may run on windows too, on top of the clean frames' cached feature (ops.encoder_head_incremental;
        the cache lives as long as the enclosing ops.frozen_weights() scope, i.e. one attack)."""
        e = self.encoder
        if e.fused_eval_ok(input_image):
            self.features = self._forward_fused_eval(input_image, roi, clean)
            return self.features
        if input_image.is_cuda and input_image.dtype == torch.float32 and e.bn1.training:
            self._prefetch_filters()
        z = self._stem(input_image)
        if _train_fused(e.bn1, z) and z.shape[2] % 2 == 0 and z.shape[3] % 2 == 0:
            f0, p = ops.stem_bn_relu_pool_train(e.bn1, z)
        else:
            f0 = e.relu(e.bn1(z))
            p = e.maxpool(f0)
        f1 = e.layer1(p)
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        self.features = [f0, f1, f2, f3, f4]
        return self.features

    def _conv1(self, x):
        c = self.encoder.conv1
        if (x.is_cuda and x.dtype == torch.float32 and c.kernel_size == (7, 7) and c.stride == (2, 2)
                and c.padding == (3, 3) and c.dilation == (1, 1) and c.groups == 1 and c.bias is None):
            return ops.stem_conv(x, c.weight)
        return c(x)

    def _head_blocks(self, aff):
        """layer1 as the argument list of ops.encoder_head_eval, or None when it is not two plain stride-1 BasicBlocks."""
        l1 = self.encoder.layer1
        if len(l1) != 2 or not all(isinstance(b, BasicBlock) and b.downsample is None and _plain3x3(b.conv1)
                                   and _plain3x3(b.conv2) for b in l1):
            return None
        return [(b.conv1.weight, aff[b.bn1], b.conv2.weight, aff[b.bn2]) for b in l1]

    def _layer2_blocks(self, aff):
        """layer2 as the ``layer2`` argument of ops.encoder_head_incremental, or None when it is not the ResNet-18 pair (a
        down-sampling BasicBlock with the standard 3x3/2 + 1x1/2 entry, then a plain stride-1 BasicBlock)."""
        l2 = self.encoder.layer2
        if len(l2) != 2 or not all(isinstance(b, BasicBlock) for b in l2):
            return None
        a, b = l2
        if not (a._is_down_pair() and _plain3x3(a.conv2) and b.downsample is None and _plain3x3(b.conv1) and _plain3x3(b.conv2)):
            return None
        return ((a.conv1.weight, aff[a.bn1], a.downsample[0].weight, aff[a.downsample[1]], a.conv2.weight, aff[a.bn2]),
                (b.conv1.weight, aff[b.bn1], b.conv2.weight, aff[b.bn2]))

    def _forward_fused_eval(self, input_image, roi=None, clean=None):
        e = self.encoder
        aff = e.eval_affine()
        if ops.weights_frozen():
            self._prefetch_filters(aff)
        layers = (e.layer1, e.layer2, e.layer3, e.layer4)
        feats = None
        if roi is not None:
            blocks = self._head_blocks(aff)
            c = e.conv1
            if (blocks is not None and c.stride == (2, 2) and c.padding == (3, 3) and c.dilation == (1, 1) and c.groups == 1
                    and c.bias is None and ops.encoder_head_ok(input_image, c.weight, [(b[0], b[2]) for b in blocks])):
                if (clean is not None and self.roi_incremental and ops.weights_frozen() and roi[0].head_incremental_ok
                        and clean.shape == input_image.shape and clean.is_cuda and clean.dtype == torch.float32):
                    # forward AND backward of the head on one window per scene, on top of the clean frames' feature 1
                    l2 = self._layer2_blocks(aff) if self.roi_incremental_layer2 else None
                    if l2 is not None and not (roi[0].layer2_incremental_ok and ops.layer2_incremental_ok(input_image, l2)):
                        l2 = None
                    cache = ops.frozen_memo(("clean_head", id(self), clean.data_ptr(), clean._version, l2 is not None),
                                            lambda: ops.clean_head(clean, c.weight, aff[e.bn1], blocks, l2))
                    outs = ops.encoder_head_incremental(input_image, roi[0], roi[1], cache, c.weight, aff[e.bn1], blocks, l2)
                    if l2 is not None:      # conv1 ... layer2 on windows: the whole-frame layers start at layer3
                        feats, layers = [outs[0], outs[1], outs[2]], layers[2:]
                    f0, y = outs[0], outs[-1]
                else:
                    # the attack's encoder head: one node, backward on the plan's windows (K19)
                    f0, y = ops.encoder_head_eval(input_image, roi[0], roi[1], c.weight, aff[e.bn1], blocks)
                if feats is None:
                    feats, layers = [f0, y], layers[1:]
        if feats is None:
            z = self._stem(input_image)
            if z.shape[2] % 2 == 0 and z.shape[3] % 2 == 0:
                f0, y = ops.stem_bn_relu_pool(z, *aff[e.bn1])
            else:
                f0 = ops.bn_act(z, *aff[e.bn1])
                y = e.maxpool(f0)
            feats = [f0]
        first = len(feats) - 1          # index (0: layer1, 1: layer2, ...) of the first layer still to run
        for li, layer in enumerate(layers, start=first):
            for bi, blk in enumerate(layer):
                if li > 0 and bi == 0 and hasattr(blk, "_is_down_pair"):
                    # y is the previous layer's output = a pyramid feature with two consumers (this block and the decoder)
                    y, feats[-1] = blk.forward_fused(y, aff, want_skip=True)
                else:
                    y = blk.forward_fused(y, aff)
            feats.append(y)
        return feats
