"""The callable the attacks differentiate through (reference ``depth_model.py:10-20,89-161``)."""
import os

import torch
import torch.nn

from . import networks


class DepthModelWrapper(torch.nn.Module):
    """encoder -> decoder -> outputs[("disp", 0)]."""

    def __init__(self, encoder, decoder) -> None:
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, input_image):
        feats = self.encoder(input_image)
        if feats[-1].is_cuda and hasattr(self.decoder, "_forward_fused"):
            return self.decoder(feats, only_scales=(0,))[("disp", 0)]     # the heads nobody reads are not computed
        return self.decoder(feats)[("disp", 0)]


    def masked_sq_mean(self, input_image, mask, plan=None, tab=None, clean=None, negate=False):
        """mean((disp_0(input_image) * mask)^2): the cost every object attack maximises (phy_obj_atk.py:92-94,
        phy_obj_atk_l0.py:125-127).  With a window plan (roi.RoiPlan + its device table) around the pasted object, and a
        decoder that supports it, the decoder below its first stage and the encoder head's backward run on the windows only;
        with ``clean`` (the same frames without the object: input_image equals it outside the plan's boxes) the encoder
        head's forward does too.  ``negate``: returns -mean(.), what the L_inf attack differentiates (phy_obj_atk.py:95)."""
        from . import ops
        # the decoder's static capability is asked BEFORE the encoder may hand back window-shaped features
        roi_dec = plan is not None and hasattr(self.decoder, "roi_ok") and self.decoder.roi_static_ok(plan)
        if plan is not None:        # what THIS call's encoder does with the plan (a plan may be used for more than one call)
            plan.head_windowed = plan.f0_compact = False
            tab = plan.device_table(input_image.device, tab)    # the plan's own table; one of another plan raises
        if roi_dec and getattr(self.encoder, "roi_backward", False):
            feats = self.encoder(input_image, roi=(plan, tab), clean=clean)
        else:
            feats = self.encoder(input_image)
        if roi_dec and self.decoder.roi_ok(feats, plan):
            return self.decoder.masked_sq_mean(feats, mask, plan, tab, negate=negate)
        if plan is not None and (plan.head_windowed or plan.f0_compact):
            # the decoder turned the plan down on the features' sizes after all: the whole-frame path serves the call
            plan.head_windowed = plan.f0_compact = False
            feats = self.encoder(input_image)
        if feats[-1].is_cuda and hasattr(self.decoder, "_forward_fused"):
            disp = self.decoder(feats, only_scales=(0,))[("disp", 0)]
        else:
            disp = self.decoder(feats)[("disp", 0)]
        cost = ops.masked_sq_mean(disp, mask)
        return -cost if negate else cost


def import_depth_model(scene_size, model_type='monodepth2', pre_model_path=None):
    """Build the ResNet-18 depth model of ``model_type`` and load ``encoder.pth`` / ``depth.pth`` (depth_model.py:89-161, which
    filters the encoder dict by key the same way).  'monodepth2' (``mono+stereo_1024x320``) and 'depthhints'
    (``DH_MS_320_1024``, BASELINE config 4) are the same architecture -- DepthHints' networks/ is Monodepth2's -- and differ in
    the weights folder only; 'manydepth' (cost-volume encoder) is outside the hot path.  The reference reads
    ``DepthNetworks/<fork>/models/<name>``; here the folder is ``pre_model_path`` or ``$DMH_MODELS_DIR/<name>``; without
    weights on disk the model is randomly initialised (no network in this environment).

    scene_size: (width, height)."""
    names = {'monodepth2': 'mono+stereo_1024x320', 'depthhints': 'DH_MS_320_1024'}
    if tuple(scene_size) not in ((1024, 320),):
        raise RuntimeError("scene size undefined!")
    if model_type == 'manydepth':
        raise RuntimeError("the manydepth depth model (cost-volume encoder) is outside the hot-path scope")
    if model_type not in names:
        raise RuntimeError("depth model unfound")
    model_path = pre_model_path
    if model_path is None and os.environ.get("DMH_MODELS_DIR"):
        cand = os.path.join(os.environ["DMH_MODELS_DIR"], names[model_type])
        model_path = cand if os.path.isdir(cand) else None
    encoder = networks.ResnetEncoder(18, False)
    decoder = networks.DepthDecoder(num_ch_enc=encoder.num_ch_enc, scales=range(4))
    if model_path is not None:
        enc = torch.load(os.path.join(model_path, "encoder.pth"), map_location="cpu")
        encoder.load_state_dict({k: v for k, v in enc.items() if k in encoder.state_dict()})
        decoder.load_state_dict(torch.load(os.path.join(model_path, "depth.pth"), map_location="cpu"))
    model = DepthModelWrapper(encoder, decoder)
    model.model_type, model.model_name = model_type, names[model_type]
    return model
