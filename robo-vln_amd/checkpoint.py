"""Reader for the reference's checkpoint format.

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
