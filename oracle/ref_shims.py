"""Import harness for the REAL reference models (test infrastructure, container-only).

/root/reference's hot-path modules are pure Python but import packages that are
not installed here (gym, habitat, habitat_baselines, torchvision, yacs).  This
file injects minimal stand-in modules so that
`robo_vln_baselines.models.seq2seq_highlevel_cma.Seq2Seq_HighLevel_CMA` and
`...seq2seq_lowlevel.Seq2Seq_LowLevel` import and run UNMODIFIED from where they
lie (recipe: SURVEY.md Appendix A).  Nothing from /root/reference is copied.

Stand-ins written here from the public architecture definitions (SURVEY.md
Appendix C), i.e. third-party arithmetic that is "parity-unpinned" against the
original packages:
  * torchvision.models.resnet50      (v1.5 bottleneck, BN)
  * habitat_baselines.rl.ddppo.policy.resnet.resnet50 + resnet_policy.ResNetEncoder (GroupNorm)
  * habitat_baselines.rl.models.simple_cnn.SimpleCNN helpers, common.utils.Flatten
Real code used as-is: transformers.BertModel (random-init BertConfig instead of
the un-downloadable 'bert-base-uncased' weights) and the reference's in-tree
`models/decoder/state_encoder.py`, which stands in for habitat's RNNStateEncoder
(it is a verbatim in-tree copy of it).

This module only runs where /root/reference exists; the GPU box never imports it.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("HCM_REFERENCE_ROOT", "/root/reference")


class AttrDict(dict):
    """yacs/habitat Config stand-in: attribute access, no-op defrost/freeze."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def defrost(self):
        pass

    def freeze(self):
        pass


# ------------------------------------------------------------------ gym
class _Space:
    pass


class _Box(_Space):
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class _DictSpace(_Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)


# ------------------------------------------------------------------ torchvision.models.resnet50
class _TVBottleneck(nn.Module):
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
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _TVResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)
        # global average (newer-torchvision behaviour) so 128^2/256^2 inputs work in flat mode
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [_TVBottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_TVBottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        x = torch.flatten(x, 1)
        return self.fc(x)


def _tv_resnet50(pretrained=False, **kw):
    return _TVResNet50()


# ------------------------------------------------------------------ habitat DDPPO GroupNorm resnet
def _gn_branch(inplanes, planes, ngroups, stride):
    return nn.Sequential(
        nn.Conv2d(inplanes, planes, 1, bias=False), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
        nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
        nn.Conv2d(planes, planes * 4, 1, bias=False), nn.GroupNorm(ngroups, planes * 4),
    )


class _GNBottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = _gn_branch(inplanes, planes, ngroups, stride)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.convs(x)
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _GNResNet(nn.Module):
    def __init__(self, in_channels, base_planes, ngroups, layers=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, base_planes, 7, 2, 3, bias=False),
                                   nn.GroupNorm(ngroups, base_planes), nn.ReLU(True))
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.inplanes = base_planes
        self.layer1 = self._make(ngroups, base_planes, layers[0], 1)
        self.layer2 = self._make(ngroups, base_planes * 2, layers[1], 2)
        self.layer3 = self._make(ngroups, base_planes * 4, layers[2], 2)
        self.layer4 = self._make(ngroups, base_planes * 8, layers[3], 2)
        self.final_channels = self.inplanes
        self.final_spatial_compress = 1.0 / 32

    def _make(self, ngroups, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.GroupNorm(ngroups, planes * 4))
        layers = [_GNBottleneck(self.inplanes, planes, ngroups, stride, ds)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_GNBottleneck(self.inplanes, planes, ngroups))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def _gn_resnet50(in_channels, base_planes, ngroups):
    return _GNResNet(in_channels, base_planes, ngroups)


class _ResNetEncoder(nn.Module):
    def __init__(self, observation_space, baseplanes=32, ngroups=32, spatial_size=128, make_backbone=None,
                 normalize_visual_inputs=False):
        super().__init__()
        assert not normalize_visual_inputs
        sp = observation_space.spaces["depth"]
        self._n_input_depth = sp.shape[2]
        spatial_size = sp.shape[0] // 2
        self.backbone = make_backbone(self._n_input_depth, baseplanes, ngroups)
        final_spatial = int(spatial_size * self.backbone.final_spatial_compress)
        ncomp = int(round(2048 / (final_spatial ** 2)))
        self.compression = nn.Sequential(nn.Conv2d(self.backbone.final_channels, ncomp, 3, padding=1, bias=False),
                                         nn.GroupNorm(1, ncomp), nn.ReLU(True))
        self.output_shape = (ncomp, final_spatial, final_spatial)

    @property
    def is_blind(self):
        return False

    def forward(self, observations):
        x = observations["depth"].permute(0, 3, 1, 2)
        x = F.avg_pool2d(x, 2)
        return self.compression(self.backbone(x))


# ------------------------------------------------------------------ habitat SimpleCNN / Flatten
class _Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class _SimpleCNN(nn.Module):
    def _conv_output_dim(self, dimension, padding, dilation, kernel_size, stride):
        out = []
        for i in range(len(dimension)):
            out.append(int(np.floor((dimension[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) / stride[i] + 1)))
        return tuple(out)

    def layer_init(self):
        for layer in self.cnn:
            if isinstance(layer, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(layer.weight, nn.init.calculate_gain("relu"))
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, val=0)

    @property
    def is_blind(self):
        return self._n_input_rgb + self._n_input_depth == 0


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_INSTALLED = False


def install():
    """Inject the stand-ins and register the reference package path.  Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    import transformers  # noqa: F401  must be imported BEFORE a fake torchvision exists
    from transformers import BertConfig, BertModel

    _mod("gym", Space=_Space, spaces=_mod("gym.spaces", Box=_Box, Dict=_DictSpace, Space=_Space))
    import logging
    _mod("habitat", Config=AttrDict, logger=logging.getLogger("habitat"))

    # the reference's own in-tree copy of RNNStateEncoder, loaded by path
    spec = importlib.util.spec_from_file_location(
        "_ref_state_encoder", os.path.join(REF_ROOT, "robo_vln_baselines/models/decoder/state_encoder.py"))
    se = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(se)

    _mod("habitat_baselines")
    _mod("habitat_baselines.common")
    _mod("habitat_baselines.common.utils", Flatten=_Flatten)
    _mod("habitat_baselines.rl")
    _mod("habitat_baselines.rl.models")
    _mod("habitat_baselines.rl.models.rnn_state_encoder", RNNStateEncoder=se.RNNStateEncoder)
    _mod("habitat_baselines.rl.models.simple_cnn", SimpleCNN=_SimpleCNN)
    _mod("habitat_baselines.rl.ddppo")
    resnet_mod = _mod("habitat_baselines.rl.ddppo.policy.resnet", resnet50=_gn_resnet50)
    _mod("habitat_baselines.rl.ddppo.policy", resnet=resnet_mod)
    _mod("habitat_baselines.rl.ddppo.policy.resnet_policy", ResNetEncoder=_ResNetEncoder)
    _mod("habitat_baselines.rl.ppo")
    _mod("habitat_baselines.rl.ppo.policy", Net=nn.Module)
    tvm = _mod("torchvision.models", resnet50=_tv_resnet50)
    _mod("torchvision", models=tvm)

    pkg = types.ModuleType("robo_vln_baselines")
    pkg.__path__ = [os.path.join(REF_ROOT, "robo_vln_baselines")]
    sys.modules["robo_vln_baselines"] = pkg

    # no network: random-init bert-base architecture; layer count may be overridden per build_models()
    def _from_pretrained(cls, *a, **k):
        return cls(BertConfig(num_hidden_layers=_BERT_LAYERS[0]))
    BertModel.from_pretrained = classmethod(_from_pretrained)
    # Visual_Ling_Attn does `.to(input.get_device())`; -1 on CPU
    torch.Tensor.get_device = lambda self: self.device
    _INSTALLED = True


_BERT_LAYERS = [12]


def model_config(cfg):
    """HCMConfig -> the attr-dict the reference model constructors read."""
    return AttrDict(
        TORCH_GPU_ID=0, ablate_instruction=bool(getattr(cfg, "ablate_instruction", False)), ablate_depth=bool(getattr(cfg, "ablate_depth", False)),
        ablate_rgb=bool(getattr(cfg, "ablate_rgb", False)),
        TRANSFORMER_INSTRUCTION_ENCODER=AttrDict(d_in=768, d_model=256),
        DEPTH_ENCODER=AttrDict(cnn_type=cfg.depth_encoder, output_size=cfg.depth_out, backbone="resnet50",
                               ddppo_checkpoint="NONE"),
        RGB_ENCODER=AttrDict(cnn_type=cfg.rgb_encoder, output_size=cfg.rgb_out, resnet_output_size=256),
        VISUAL_LING_ATTN=AttrDict(N=cfg.vla_layers, vis_in_features=cfg.vis_in, ins_in_features=cfg.ins_in,
                                  d_model=cfg.d_model, h=cfg.vla_heads, d_ff=cfg.d_ff, dropout=0.25),
        IMAGE_CROSS_MODAL_ENCODER=AttrDict(d_model=cfg.cm_d_model),
        SEQ2SEQ=AttrDict(use_prev_action=False),
        STATE_ENCODER=AttrDict(hidden_size=cfg.hidden, rnn_type=cfg.rnn_type),
        PROGRESS_MONITOR=AttrDict(use=False, alpha=1.0),
    )


def obs_space(cfg):
    # declare the RGB *space* 224x224 (ctor NameError otherwise); tensors may differ
    return _DictSpace({
        "rgb": _Box(0, 255, (224, 224, 3), np.uint8),
        "depth": _Box(0.0, 1.0, ((*cfg.depth_shape, 1) if hasattr(cfg, "depth_shape") else (cfg.depth_hw, cfg.depth_hw, 1)), np.float32),
    })


def build_models(cfg, hi_sd=None, lo_sd=None, want_hi=True, want_lo=True):
    """Construct the reference hi/lo modules (hierarchical_trainer.py:315-328) and load numpy state_dicts
    with strict=True -- which also validates robo-vln_amd/synth.py's key/shape specs."""
    install()
    _BERT_LAYERS[0] = cfg.bert_layers
    mc = model_config(cfg)
    space = obs_space(cfg)
    if cfg.rgb_encoder != "TorchVisionResNet50":
        space.spaces["rgb"] = _Box(0, 255, (*cfg.rgb_shape, 3), np.uint8)
    hi = lo = None
    if want_hi:
        from robo_vln_baselines.models.seq2seq_highlevel_cma import Seq2Seq_HighLevel_CMA
        hi = Seq2Seq_HighLevel_CMA(space, cfg.num_actions, mc, 1).eval()
        if hi_sd is not None:
            hi.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in hi_sd.items()}, strict=True)
    if want_lo:
        from robo_vln_baselines.models.seq2seq_lowlevel import Seq2Seq_LowLevel
        lo = Seq2Seq_LowLevel(space, cfg.lo_actions, cfg.num_sub_tasks, mc, 1).eval()
        if lo_sd is not None:
            lo.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in lo_sd.items()}, strict=True)
    return hi, lo


def ref_masks(mask_b):
    """(B,) -> the (B,2,1) shape that makes the reference's batched single_forward broadcast correctly
    (input-only workaround)."""
    m = torch.as_tensor(mask_b, dtype=torch.float32).view(-1, 1, 1)
    return m.expand(-1, 2, 1).contiguous()


def cma_model_config(cfg):
    """CMAConfig -> the attr-dict CMANet's constructor reads (cma.py:28-186)."""
    return AttrDict(
        TORCH_GPU_ID=0, ablate_instruction=bool(getattr(cfg, "ablate_instruction", False)), ablate_depth=bool(getattr(cfg, "ablate_depth", False)),
        ablate_rgb=bool(getattr(cfg, "ablate_rgb", False)),
        INSTRUCTION_ENCODER=AttrDict(vocab_size=cfg.vocab_size, embedding_size=cfg.embedding_size, hidden_size=cfg.instr_hidden,
                                     rnn_type=cfg.instr_rnn, bidirectional=cfg.bidirectional, final_state_only=True,
                                     use_pretrained_embeddings=False, fine_tune_embeddings=False),
        DEPTH_ENCODER=AttrDict(cnn_type="VlnResnetDepthEncoder", output_size=cfg.depth_out, backbone="resnet50",
                               ddppo_checkpoint="NONE"),
        RGB_ENCODER=AttrDict(cnn_type="TorchVisionResNet50", output_size=cfg.rgb_out, resnet_output_size=256),
        CMA=AttrDict(use=True, use_prev_action=False, rcm_state_encoder=False),
        STATE_ENCODER=AttrDict(hidden_size=cfg.hidden, rnn_type=cfg.rnn_type),
        PROGRESS_MONITOR=AttrDict(use=False, alpha=1.0),
    )


def build_cma(cfg, sd=None):
    """Construct the reference CMANet (robo_vln_trainer.py:326-331) and load a numpy state_dict with strict=True."""
    install()
    from robo_vln_baselines.models.cma import CMANet
    net = CMANet(obs_space(cfg), cfg.num_actions, cma_model_config(cfg)).eval()
    if sd is not None:
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return net
