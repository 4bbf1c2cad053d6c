"""Write tests/golden/ref_ddppo_structure.pth: a DDPPO point-nav checkpoint as `VlnResnetDepthEncoder.__init__` expects one
(robo_vln_baselines/models/encoders/resnet_encoders.py:38-52: `torch.load(checkpoint)["state_dict"]`, keys
`actor_critic.net.visual_encoder.<name>`, everything else skipped, strict load into `self.visual_encoder`), hollowed like the trainer-checkpoint
fixture (names / order / shapes / dtypes real, values a stride-0 zero) -- and PROVE it with the real reference class: the script constructs
`VlnResnetDepthEncoder(checkpoint=<the fixture>)` from /root/reference, whose strict `load_state_dict` accepts exactly this key set.
The visual-encoder names come from the reference class's own `visual_encoder.state_dict()`; the extra keys (`actor_critic.net.state_encoder…`,
`actor_critic.critic.fc…`, `actor_critic.net.prev_action_embedding…`) are the kinds a habitat PointNavResNetPolicy state_dict also holds
and that the remap must skip (the third has 'visual_encoder' nowhere; `critic.fc` has only two components behind the cut).

Run in the build container only:   python oracle/gen_ddppo_fixture.py
Test infrastructure.  tests/test_checkpoint_cpu.py reads the fixture through robo-vln_amd/checkpoint.py.
"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hcm_pkg  # noqa: E402

hcm_pkg.load()
from robo_vln_amd.config import HCMConfig  # noqa: E402
from oracle import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_ddppo_structure.pth")


def hollow(v):
    return torch.zeros((), dtype=v.dtype).expand(v.shape) if v.dim() else torch.zeros((), dtype=v.dtype)


def main():
    cfg = HCMConfig().validate()
    ref_shims.install()
    from robo_vln_baselines.models.encoders.resnet_encoders import VlnResnetDepthEncoder
    space = ref_shims.obs_space(cfg)
    enc = VlnResnetDepthEncoder(space, output_size=cfg.depth_out, checkpoint="NONE", backbone="resnet50", spatial_output=True)
    sd = OrderedDict()
    sd["actor_critic.net.prev_action_embedding.weight"] = hollow(torch.zeros(5, 32))
    for k, v in enc.visual_encoder.state_dict().items():
        sd["actor_critic.net.visual_encoder." + k] = hollow(v)
    sd["actor_critic.net.state_encoder.rnn.weight_ih_l0"] = hollow(torch.zeros(2048, 576))
    sd["actor_critic.critic.fc.weight"] = hollow(torch.zeros(1, 512))
    sd["actor_critic.critic.fc.bias"] = hollow(torch.zeros(1))
    torch.save({"state_dict": sd, "config": ref_shims.model_config(cfg), "extra_state": {"step": 0}}, OUT)
    # the real reference constructor loads it (strict); a key the remap must not let through would raise here
    # (the reference's bare `torch.load(checkpoint)` dates from torch 1.3, where full un-pickling was the default; torch 2.10 defaults to
    # weights_only=True, which refuses the pickled config object a real DDPPO file also carries: restore the old default for this one call)
    real_load = torch.load
    torch.load = lambda f, *a, **k: real_load(f, *a, **{**k, "weights_only": False})
    try:
        enc2 = VlnResnetDepthEncoder(space, output_size=cfg.depth_out, checkpoint=OUT, backbone="resnet50", spatial_output=True)
    finally:
        torch.load = real_load
    n = len(enc2.visual_encoder.state_dict())
    print(f"{OUT}: {os.path.getsize(OUT)} bytes, {len(sd)} tensors ({n} of the visual encoder); the reference constructor accepted it")


if __name__ == "__main__":
    main()
