"""Readers for the reference's checkpoint formats: the trainer's own file, and the DDPPO point-nav checkpoint the
depth trunk is initialised from (`load_ddppo_depth_weights`, models/encoders/resnet_encoders.py:38-52).

`RoboDaggerTrainer.save_checkpoint` writes `torch.save({"high_level_state_dict", "low_level_state_dict", "config"})`
(robo_vln_baselines/hierarchical_trainer.py:349-363); `_setup_actor_critic_agent` loads the two state_dicts with
`load_state_dict` (:343-345).  The pickled `config` is a yacs `CfgNode`; yacs is not needed here: every class outside a
small allow-list (tensor rebuild helpers, OrderedDict, builtin containers) un-pickles as an inert dict stand-in, so a
checkpoint cannot execute code through the un-pickler.
"""
import pickle

import torch

# buffers that some `transformers` versions register inside BertEmbeddings and older checkpoints therefore carry;
# they are not parameters of the model and are ignored (everything else stays strict)
IGNORED_SUFFIXES = ("embeddings.position_ids", "embeddings.token_type_ids")


class _Stub(dict):
    """Inert stand-in for every class a checkpoint's pickle names outside the allow-list below (e.g. yacs.config.CfgNode,
    habitat's Config): constructed without running any code of the named class, keeps whatever state the pickle sets."""

    def __init__(self, *a, **k):
        dict.__init__(self)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)

    def __reduce_ex__(self, protocol):
        return (dict, (dict(self),))


# What a `torch.save`d state_dict needs and nothing else.  Un-pickling resolves ONLY these globals; any other (module, name) --
# importable or not -- becomes an inert _Stub subclass, so loading a checkpoint from an untrusted source cannot run code through
# `find_class` (the hole of a plain pickle / weights_only=False load).
_ALLOWED = {
    ("collections", "OrderedDict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.serialization", "_get_layout"), ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "bytes"), ("builtins", "complex"),
}


def _allowed(module, name):
    if (module, name) in _ALLOWED:
        return True
    # legacy typed storage classes (torch.FloatStorage, ...) and dtype singletons (torch.float32, ...)
    if module == "torch" and (name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)):
        return True
    return False


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if _allowed(module, name):
            return super().find_class(module, name)
        return type(name, (_Stub,), {"__module__": module})


class _TolerantPickle:
    Unpickler = _TolerantUnpickler
    __name__ = "tolerant_pickle"

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def load_checkpoint(path_or_file, map_location="cpu"):
    """-> (high_level_state_dict, low_level_state_dict, config) from a reference checkpoint file."""
    ckpt = torch.load(path_or_file, map_location=map_location, pickle_module=_TolerantPickle, weights_only=False)
    for k in ("high_level_state_dict", "low_level_state_dict"):
        if k not in ckpt:
            raise KeyError(f"checkpoint has no '{k}' (keys: {list(ckpt)})")
    clean = []
    for sd in (ckpt["high_level_state_dict"], ckpt["low_level_state_dict"]):
        clean.append({k: v for k, v in sd.items() if not k.endswith(IGNORED_SUFFIXES)})
    return clean[0], clean[1], ckpt.get("config")


def load_ddppo_depth_weights(path_or_file, map_location="cpu"):
    """The remap `VlnResnetDepthEncoder.__init__` applies to a DDPPO point-nav checkpoint (models/encoders/resnet_encoders.py:38-52):
    for every key of `ckpt["state_dict"]`, drop the first two dotted components (`actor_critic.net.`), keep only what then starts with
    `visual_encoder.`, and strip that component too.  -> {name relative to `depth_encoder.visual_encoder`: tensor}; the file is read through
    the same restricted un-pickler as the trainer's checkpoints (its pickled config classes are not importable here either)."""
    ckpt = torch.load(path_or_file, map_location=map_location, pickle_module=_TolerantPickle, weights_only=False)
    if "state_dict" not in ckpt:
        raise KeyError(f"DDPPO checkpoint has no 'state_dict' (keys: {list(ckpt)})")
    out = {}
    for k, v in ckpt["state_dict"].items():
        parts = k.split(".")[2:]                                     # :41
        if not parts or parts[0] != "visual_encoder":               # :42-43 (a key with fewer than three components raises IndexError in
            continue                                                 #  the reference; nothing a policy state_dict contains -- skipped)
        out[".".join(parts[1:])] = v                                 # :45-46
    if not out:
        raise KeyError("DDPPO checkpoint holds no '<a>.<b>.visual_encoder.*' tensors")
    return out


def apply_ddppo_depth_weights(state_dict, ddppo_weights, prefix="depth_encoder.visual_encoder."):
    """`self.visual_encoder.load_state_dict(weights_dict, strict=True)` (resnet_encoders.py:49) on a model state_dict held as a plain dict:
    the DDPPO names must be EXACTLY the names under `prefix`, shapes equal -- missing / unexpected / mismatched entries raise RuntimeError
    listing them, as torch's strict load does.  Returns a new dict with those tensors replaced (how the reference gets its depth trunk when
    it trains from scratch with MODEL.DEPTH_ENCODER.ddppo_checkpoint set; a trainer checkpoint already contains the result)."""
    own = {k[len(prefix):]: k for k in state_dict if k.startswith(prefix)}
    if not own:
        raise KeyError(f"state_dict has no '{prefix}*' entries")
    missing = sorted(set(own) - set(ddppo_weights))
    unexpected = sorted(set(ddppo_weights) - set(own))
    bad_shape = [f"{n}: checkpoint {tuple(ddppo_weights[n].shape)} vs model {tuple(torch.as_tensor(state_dict[own[n]]).shape)}"
                 for n in sorted(set(own) & set(ddppo_weights)) if tuple(ddppo_weights[n].shape) != tuple(torch.as_tensor(state_dict[own[n]]).shape)]
    if missing or unexpected or bad_shape:
        raise RuntimeError("DDPPO depth weights do not fit the depth trunk (strict): " +
                           "; ".join(x for x in (f"missing {missing[:6]}{'...' if len(missing) > 6 else ''}" if missing else "",
                                                 f"unexpected {unexpected[:6]}{'...' if len(unexpected) > 6 else ''}" if unexpected else "",
                                                 f"size mismatch {bad_shape[:6]}" if bad_shape else "") if x))
    out = dict(state_dict)
    for n, full in own.items():
        out[full] = ddppo_weights[n]
    return out


def save_checkpoint(path_or_file, high_level_state_dict, low_level_state_dict, config=None):
    """Write the same three-key dict the reference trainer writes (tensors, not numpy)."""
    def as_t(sd):
        return {k: torch.as_tensor(v) for k, v in sd.items()}
    torch.save({"high_level_state_dict": as_t(high_level_state_dict), "low_level_state_dict": as_t(low_level_state_dict),
                "config": config}, path_or_file)


def engine_from_checkpoint(path, cfg, calibrate_on=None, **engine_kwargs):
    """Build an HCMEngine from a reference checkpoint (strict key/shape check inside libhcm).
    The fp16 range check of hcm_finalize runs on synthetic noise; a trained checkpoint on real frames deserves the real thing: the engine
    keeps the f32 host copies of the weights (keep_host_weights) until `calibrate_on` -- a batch of real observations -- has been through
    `engine.calibrate()`, which repeats the check and the repair (range fold / bf16 fall-back) on them and then releases the copies.
    Without `calibrate_on` the copies stay: call `engine.calibrate(observations)` once real observations exist."""
    from .policy import HCMEngine
    hi_sd, lo_sd, _ = load_checkpoint(path)
    engine_kwargs.setdefault("keep_host_weights", True)
    eng = HCMEngine(cfg, hi_sd, lo_sd, **engine_kwargs)
    if calibrate_on is not None:
        eng.calibrate(calibrate_on, release_host_weights=True)
    return eng
