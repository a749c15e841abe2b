"""U-Net depth decoder of Monodepth2 (MD2/networks/depth_decoder.py:17-65): reflection-padded 3x3
convs + ELU, nearest x2 upsampling, skip connections, one sigmoid disparity head per scale.
``self.decoder`` is the ModuleList in the reference's insertion order, so ``depth.pth`` keys match."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from ..layers import ConvBlock, Conv3x3, upsample


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = 'nearest'
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if (use_skips and i > 0) else 0)
            self.convs[("upconv", i, 1)] = ConvBlock(cin, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features, only_scales=None):
        """``only_scales``: compute only these disparity heads (the stages themselves always run).  DepthModelWrapper
        reads ("disp", 0) alone (depth_model.py:19), so inside an attack the other three heads -- which the reference
        computes and throws away 21 times per training step -- are skipped; the returned entries are unchanged."""
        if input_features[-1].is_cuda and self.use_skips and self.upsample_mode == 'nearest':
            return self._forward_fused(input_features, only_scales)
        return self._forward_reference(input_features)

    def _prefetch_filters(self):
        """The K10 filters of the stages in one launch (ops.wino_prefetch): once per ops.frozen_weights() scope inside an
        attack, once per forward otherwise; forward and -- when a backward can follow -- backward-data forms."""
        from .. import ops
        dirs = (False, True) if torch.is_grad_enabled() else (False,)
        jobs = [(blk.conv.conv.weight, bw, None) for key, blk in self.convs.items() if key[0] == "upconv" for bw in dirs]
        if ops.weights_frozen():
            ops.frozen_memo(("prefetch", id(self)), lambda: ops.wino_prefetch(jobs) or True)
        else:
            ops.wino_prefetch(jobs)

    def _forward_fused(self, input_features, only_scales=None):
        """Same arithmetic as the reference forward, with the element-wise passes between the convolutions fused
        into the HIP glue kernels (ops.up_cat_pad / ops.elu_pad) and the convolutions run un-padded on pre-padded
        tensors (ops.conv3x3: the Winograd-MFMA kernel where the shape fills the chip, MIOpen otherwise).  Identical
        parameters / state_dict."""
        from .. import ops

        def conv(block, t):
            c = block.conv
            return ops.conv3x3(t, c.weight, c.bias, 0)

        self.outputs = {}
        self._prefetch_filters()
        p = ops.elu_pad(input_features[-1], apply_elu=False)
        for i in range(4, -1, -1):
            y = conv(self.convs[("upconv", i, 0)].conv, p)
            p = ops.up_cat_pad(y, input_features[i - 1] if i > 0 else None)
            z = conv(self.convs[("upconv", i, 1)].conv, p)
            p = ops.elu_pad(z)                      # feeds both the next stage and this scale's disparity head
            if i in self.scales and (only_scales is None or i in only_scales):
                self.outputs[("disp", i)] = self.sigmoid(conv(self.convs[("dispconv", i)], p))
        return self.outputs

    def roi_static_ok(self, plan):
        """What can be said about ``roi_ok`` before any feature exists (asked by DepthModelWrapper BEFORE the encoder is allowed
        to run its windowed head: a decoder that cannot take the plan must be handed whole-frame features)."""
        return bool(self.use_skips and self.upsample_mode == 'nearest' and 0 in self.scales and self.num_output_channels == 1
                    and plan.depth in (2, 3, 4))

    def roi_ok(self, input_features, plan):
        """The attack's windowed cost (ops.roi_tail_cost) applies to these features under ``plan`` (roi.RoiPlan): the fused
        CUDA path, scale 0 among the heads, the reference's channel plan, the frame sizes of a five-level pyramid (feature 0
        may be its compact "hz" window: plan.f0_compact)."""
        from .. import ops
        depth = plan.depth
        if not (input_features[-1].is_cuda and len(input_features) == 5 and self.roi_static_ok(plan)):
            return False
        h1, w1 = plan.H >> 2, plan.W >> 2
        if any(tuple(f.shape[2:]) != (h1 >> (k - 1), w1 >> (k - 1)) for k, f in enumerate(input_features) if k >= 1):
            return False
        f0 = input_features[0]
        if tuple(f0.shape[2:]) != (plan.size["hz"] if plan.f0_compact else (2 * h1, 2 * w1)):
            return False
        top = input_features[depth]     # stands in for upconv(depth,0)'s output: same size as feature `depth`
        return ops.roi_tail_ok(top.new_empty((1, int(self.num_ch_dec[depth])) + tuple(top.shape[2:])),
                               tuple(input_features[:depth]), self._tail_convs(depth), depth)

    def _tail_convs(self, depth):
        c = self.convs
        out = []
        for i in range(depth, -1, -1):
            if i < depth:
                out.append(c[("upconv", i, 0)].conv.conv)
            out.append(c[("upconv", i, 1)].conv.conv)
        return out + [c[("dispconv", 0)].conv]

    def masked_sq_mean(self, input_features, mask, plan, tab, negate=False):
        """mean((disp_0 * mask)^2) -- the attack's cost (phy_obj_atk.py:92-94) -- with the decoder below stage
        (plan.depth, 0) evaluated on the windows of ``plan`` only (roi.RoiPlan around the pasted object, where the mask
        lives): the stages above run on the whole maps, upconv(depth,1) ... dispconv(0) inside the windows.  Exact, not an
        approximation: the mask is zero outside the plan's boxes, so nothing outside the windows' receptive field reaches
        the cost.  ``negate``: returns minus the mean (the attacks' `cost = -loss`), signed inside the cost kernel."""
        from .. import ops

        def conv(block, t):
            c = block.conv
            return ops.conv3x3(t, c.weight, c.bias, 0)

        depth = plan.depth
        self._prefetch_filters()
        p = ops.elu_pad(input_features[-1], apply_elu=False)
        for i in range(4, depth, -1):
            y = conv(self.convs[("upconv", i, 0)].conv, p)
            p = ops.up_cat_pad(y, input_features[i - 1])
            p = ops.elu_pad(conv(self.convs[("upconv", i, 1)].conv, p))
        y_top = conv(self.convs[("upconv", depth, 0)].conv, p)
        return ops.roi_tail_cost(y_top, input_features[:depth], mask, plan, tab, self._tail_convs(depth), negate=negate)

    def _forward_reference(self, input_features):
        self.outputs = {}
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = upsample(self.convs[("upconv", i, 0)](x))
            if self.use_skips and i > 0:
                x = torch.cat([x, input_features[i - 1]], 1)
            x = self.convs[("upconv", i, 1)](x)
            if i in self.scales:
                self.outputs[("disp", i)] = self.sigmoid(self.convs[("dispconv", i)](x))
        return self.outputs
