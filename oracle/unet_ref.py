"""CPU restatement of the depth network the attacks differentiate through: Monodepth2's ResNet-18 encoder + depth decoder.

Reference (under /root/reference/DepthNetworks/monodepth2):
  ResnetEncoder.forward      networks/resnet_encoder.py:85-98   ((x - 0.45) / 0.225 -> conv1 / bn1 / relu -> maxpool + layer1 ...)
  DepthDecoder               networks/depth_decoder.py:17-65
  ConvBlock / Conv3x3 / upsample   layers.py:106-136,200-204    (ReflectionPad2d(1) + Conv2d(3), ELU, nearest x2)
  DepthModelWrapper          depth_model.py:10-20               (encoder -> decoder -> outputs[("disp", 0)])

The encoder body is ``torchvision.models.resnet18`` (torchvision==0.8.2, requirements.txt:93): third-party and absent from
this image, so its published algorithm is restated here -- BasicBlock = conv3x3(stride) / bn / relu / conv3x3 / bn, the 1x1
stride-2 conv + bn shortcut where a layer changes resolution, identity add, relu; ResNet = conv7x7/2 / bn / relu /
maxpool(3,2,1) / four layers of two blocks / (avgpool / fc: declared for the state dict, never run by the encoder).
Module and parameter names follow torchvision's, construction order follows the reference's (so a seed reproduces the
reference decoder's initial weights), and every state dict of the product loads with strict=True.

Plain ``torch.nn`` modules: ``.double()`` gives the float64 anchor of the gradient tests.  Test infrastructure only (see
oracle/__init__.py): nothing in the product imports this file.
Pinned by: tests/golden/unet_decoder.npz -- the reference's own networks.DepthDecoder run by oracle/make_goldens.py (same
seed, same features).  The torchvision part has no reference fixture ("parity unpinned" for BasicBlock's op order; it is
checked against the product's independent module path on the CPU, tests/test_oracle_golden.py).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------- torchvision 0.8.2 resnet18
class BasicBlockRef(nn.Module):
    """torchvision/models/resnet.py (0.8.2) BasicBlock, expansion 1."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet18Ref(nn.Module):
    """torchvision.models.resnet18(pretrained=False): layers [2, 2, 2, 2] of BasicBlock."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2, 1)
        self.layer2 = self._make_layer(128, 2, 2)
        self.layer3 = self._make_layer(256, 2, 2)
        self.layer4 = self._make_layer(512, 2, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlockRef(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlockRef(self.inplanes, planes))
        return nn.Sequential(*layers)


class ResnetEncoderRef(nn.Module):
    """networks/resnet_encoder.py:62-98 with num_layers = 18, one input image."""

    def __init__(self):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet18Ref()

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        x = e.bn1(e.conv1(x))
        feats = [e.relu(x)]
        feats.append(e.layer1(e.maxpool(feats[-1])))
        feats.append(e.layer2(feats[-1]))
        feats.append(e.layer3(feats[-1]))
        feats.append(e.layer4(feats[-1]))
        return feats


# ------------------------------------------------------------------------------------------- layers.py / depth_decoder.py
class Conv3x3Ref(nn.Module):
    """layers.py:121-136."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlockRef(nn.Module):
    """layers.py:106-118."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3Ref(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=False)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoderRef(nn.Module):
    """networks/depth_decoder.py:17-65 (use_skips, nearest upsampling); same construction order, so the same seed gives the
    reference decoder's initial weights, and the same ``decoder.{i}`` state-dict keys."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1):
        super().__init__()
        self.scales = list(scales)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlockRef(num_ch_in, self.num_ch_dec[i])
            num_ch_in = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if i > 0 else 0)
            self.convs[("upconv", i, 1)] = ConvBlockRef(num_ch_in, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3Ref(self.num_ch_dec[s], num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features):
        outputs = {}
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            x = [F.interpolate(x, scale_factor=2, mode="nearest")]      # layers.py:200-204
            if i > 0:
                x += [input_features[i - 1]]
            x = self.convs[("upconv", i, 1)](torch.cat(x, 1))
            if i in self.scales:
                outputs[("disp", i)] = torch.sigmoid(self.convs[("dispconv", i)](x))
        return outputs


class UNetRef(nn.Module):
    """depth_model.py:10-20 DepthModelWrapper: images -> disparity at scale 0."""

    def __init__(self):
        super().__init__()
        self.encoder = ResnetEncoderRef()
        self.decoder = DepthDecoderRef(self.encoder.num_ch_enc)

    def forward(self, input_image):
        return self.decoder(self.encoder(input_image))[("disp", 0)]

    @classmethod
    def twin_of(cls, model, dtype=torch.float32):
        """A CPU copy (``dtype``) of a product DepthModelWrapper: same parameters and BatchNorm statistics, same mode."""
        twin = cls()
        twin.encoder.load_state_dict({k: v.detach().cpu() for k, v in model.encoder.state_dict().items()}, strict=True)
        twin.decoder.load_state_dict({k: v.detach().cpu() for k, v in model.decoder.state_dict().items()}, strict=True)
        twin.train(model.training)
        return twin.to(dtype)


def randomize_batchnorm(model, seed):
    """Non-trivial BatchNorm statistics and affine parameters (a fresh network has mean 0 / var 1 / weight 1 / bias 0, under
    which eval-mode BatchNorm is the identity and a scale / shift bug would pass)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            c = m.num_features
            with torch.no_grad():
                m.running_mean.copy_((torch.rand(c, generator=g) - 0.5) * 0.2)
                m.running_var.copy_(0.5 + torch.rand(c, generator=g))
                m.weight.copy_(0.8 + 0.4 * torch.rand(c, generator=g))
                m.bias.copy_((torch.rand(c, generator=g) - 0.5) * 0.2)


def set_relu_margins(model, images=None, k=16.0, seed=0):
    """Give every ReLU of the encoder a margin: no pre-activation within ~k - 8 standard deviations of zero (convolution
    outputs at object edges reach 8 sigma on KITTI-like frames: k = 8 leaves margins of 1e-5, k = 16 of 0.2).

    Why: the gradient of the attack cost is discontinuous in the ReLU masks, and an fp32 implementation flips the mask of a
    unit whose pre-activation is ~1e-5 of its scale.  ONE such flip changes the image gradient by 1e-5 ... 1e-3 of its norm
    (measured on this network: the fp32 oracle is 5.6e-5 from its float64 self, 5.5e-7 with its own masks forced into the
    float64 run), so a strict float64 gate is impossible for ANY fp32 code on natural weights.  With weight 1 and bias +-k per
    channel on every BatchNorm that feeds a ReLU (a channel is robustly on or robustly off; the residual branches keep one
    type per channel inside a layer, the shortcut BatchNorms are scaled to 0.1) no unit is near the kink, every fp32
    implementation has the float64 masks, and the gate can be 1e-5.  Masks still matter (half the channels are dead).

    ``model``: anything with torchvision's names under ``model.encoder.encoder`` (UNetRef or the product's DepthModelWrapper).
    ``images``: if given, the running statistics are set to those of this batch (eval-mode calibration: one train-mode forward
    with momentum 1), so that eval-mode BatchNorm normalises as train mode does.  Returns the model."""
    g = torch.Generator().manual_seed(seed)
    net = model.encoder.encoder

    def types(c):
        return (torch.rand(c, generator=g) < 0.5).float() * 2 - 1

    def set_bn(bn, bias, weight=1.0):
        with torch.no_grad():
            bn.weight.fill_(weight)
            bn.bias.copy_(bias.to(bn.bias))

    t_prev = types(64)
    set_bn(net.bn1, k * t_prev)
    for name in ("layer1", "layer2", "layer3", "layer4"):
        layer = getattr(net, name)
        c = layer[0].bn2.num_features
        t_out = t_prev if layer[0].downsample is None else types(c)   # layer1's identity is the stem's output: keep its types
        for block in layer:
            set_bn(block.bn1, k * types(c))
            set_bn(block.bn2, k * t_out)
            if block.downsample is not None:
                set_bn(block.downsample[1], torch.zeros(c), 0.1)
        t_prev = t_out
    if images is not None:
        bns = [m for m in model.encoder.modules() if isinstance(m, nn.BatchNorm2d)]
        was_training = model.training
        for m in bns:
            m.momentum = 1.0
        model.train()
        with torch.no_grad():
            model.encoder(images)
        for m in bns:
            m.momentum = 0.1
        model.train(was_training)
    return model


def min_relu_margin(model, images):
    """min |pre-activation| over every ReLU of the encoder on ``images`` (module path of UNetRef), per ReLU as a dict."""
    out = {}
    names = {m: n for n, m in model.named_modules()}
    handles = []

    def hook(m, inp, _):
        key = names[m]
        out[key + "#%d" % sum(k.startswith(key + "#") for k in out)] = float(inp[0].detach().abs().min())
    for m in model.encoder.modules():
        if isinstance(m, nn.ReLU):
            handles.append(m.register_forward_hook(hook))
    with torch.no_grad():
        model.encoder(images)
    for h in handles:
        h.remove()
    return out
