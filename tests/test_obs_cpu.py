import numpy as np
import torch

from robo_vln_amd.obs import ObsStager, batch_obs


def _obs(n, hw=8, L=5, as_float=False):
    rng = np.random.default_rng(0)
    out = []
    for i in range(n):
        rgb = rng.integers(0, 256, (hw, hw, 3), dtype=np.uint8)
        out.append({"rgb": rgb.astype(np.float32) if as_float else rgb, "depth": rng.random((hw, hw, 1), dtype=np.float32),
                    "instruction": rng.integers(1, 30000, (L,))})
    return out


def test_batch_obs_stacks_like_the_reference_helper():
    obs = _obs(3)
    b = batch_obs(obs)
    assert b["rgb"].shape == (3, 8, 8, 3) and b["rgb"].dtype == torch.uint8
    assert b["depth"].shape == (3, 8, 8, 1) and b["depth"].dtype == torch.float32
    assert b["instruction"].shape == (3, 5) and b["instruction"].dtype == torch.int32
    for i in range(3):
        assert np.array_equal(b["rgb"][i].numpy(), obs[i]["rgb"])
        assert np.array_equal(b["depth"][i].numpy(), obs[i]["depth"])
        assert np.array_equal(b["instruction"][i].numpy(), obs[i]["instruction"])
    # float 0..255 frames (the batch_obs float32 contract) round to the same uint8 values
    b2 = batch_obs(_obs(3, as_float=True))
    assert torch.equal(b["rgb"], b2["rgb"])


def test_stager_reuses_buffers_and_skips_unchanged_instruction():
    st = ObsStager(2, 8, 8, 5)
    a = st.stage(_obs(2))
    ids = a["instruction"].clone()
    ob = _obs(2)
    ob[0]["instruction"] = np.zeros(5, dtype=np.int64)
    b = st.stage(ob, instruction_changed=False)
    assert b["rgb"].data_ptr() == a["rgb"].data_ptr()
    assert torch.equal(b["instruction"], ids)
