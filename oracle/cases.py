"""Golden-vector case table shared by `oracle/gen_golden.py` (which runs the imported reference)
and the tests (which replay the same seeded inputs through the oracle / the HIP path).
Test infrastructure only."""
import numpy as np

import hcm_pkg

_pkg = hcm_pkg.load()
from robo_vln_amd.config import HCMConfig, CMAConfig  # noqa: E402
from robo_vln_amd import synth             # noqa: E402

# name -> (config kwargs, batch, steps, which models)
CASES = {
    # BASELINE.json configs[0]: B=4 in the bench; goldens use B=2 to stay small
    "cfg0_128_L20_N2": (dict(rgb_hw=128, depth_hw=128, instr_len=20, vla_layers=2), 2, 3, "both"),
    # configs[1]/[2] shape: 256x256, L=80, N=1, full BERT
    "cfg1_256_L80_N1": (dict(), 2, 3, "both"),
    # GRU state encoder (MODEL.STATE_ENCODER.rnn_type), short BERT to keep it quick
    "gru_128_L20": (dict(rgb_hw=128, depth_hw=128, instr_len=20, rnn_type="GRU", bert_layers=2), 2, 3, "both"),
    # low-level model with SimpleCNN encoders (configs[3] encoder; high-level cannot be built with them)
    "lo_simplecnn_256": (dict(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN"), 2, 2, "lo"),
    # configs[4] shape: L=160, N=6 (hi model), B=1, 128x128 frames to keep CPU time down
    "cfg4_L160_N6": (dict(rgb_hw=128, depth_hw=128, instr_len=160, vla_layers=6), 1, 1, "hi"),
    # reference-native frame sizes (robo_vln_task.yaml:10-17): RGB 224 (7x7 -> overlapping adaptive pool), depth 256
    "native_224_256": (dict(rgb_hw=224, depth_hw=256, instr_len=40, bert_layers=2), 1, 1, "both"),
    # working ablation flags of both models (seq2seq_highlevel_cma.py:185-188, seq2seq_lowlevel.py:132-135)
    "ablate_depth_128": (dict(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2, ablate_depth=True), 2, 2, "both"),
    "ablate_rgb_128": (dict(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2, ablate_rgb=True), 2, 2, "both"),
    # depth frames of 64*k pixels with k not a power of two: habitat's ResNetEncoder then has compression channel counts that are not a
    # power of two -- 192 px -> 3x3 map x 228 channels, 320 px -> 5x5 x 82 (resnet_encoders.py:37-62)
    "depth192_128": (dict(rgb_hw=128, depth_hw=192, instr_len=20, bert_layers=2), 2, 2, "both"),
    "depth320_128": (dict(rgb_hw=128, depth_hw=320, instr_len=20, bert_layers=1), 1, 1, "both"),
    "depth384_128": (dict(rgb_hw=128, depth_hw=384, instr_len=20, bert_layers=1), 1, 1, "both"),        # 6x6 x 57 (an odd count)
    # non-square RGB frames (TorchVisionResNet50 ends in adaptive pools, resnet_encoders.py:211-236): landscape 4:3 and a portrait frame
    # whose stem / pool maps have odd sizes (H 200 -> 100 -> 50, W 152 -> 76 -> 38 ... 7 x 5 final map)
    "rgb_160x224": (dict(rgb_hw=160, rgb_w=224, depth_hw=128, instr_len=20, bert_layers=2), 2, 2, "both"),
    "rgb_200x152": (dict(rgb_hw=200, rgb_w=152, depth_hw=128, instr_len=20, bert_layers=1), 1, 1, "both"),
    # ... and through the low-level model's SimpleRGBCNN (FC sized from the two dimensions on their own, simple_cnns.py:63-73): 10 x 17 final map
    "lo_simplecnn_rgb_120x176": (dict(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN", rgb_hw=120, rgb_w=176, depth_hw=128), 2, 2, "lo"),
    # ... and SimpleDepthCNN on a non-square depth frame (the width is not a multiple of 4: the generic first-conv path)
    "lo_simplecnn_depth_152x218": (dict(depth_encoder="SimpleDepthCNN", rgb_encoder="SimpleRGBCNN", rgb_hw=128, depth_hw=152, depth_w=218), 2, 2, "lo"),
}

# The reference's eval loop feeds the model the UNPADDED token ids of the episode's instruction as a (1, L) tensor
# (common/utils.py:18-20 returns `output.ids`; hierarchical_trainer.py:1193-1196), L differing from episode to episode.
# name -> (config kwargs, batch, [L of step 0, L of step 1, ...]): one engine / one reference model stepped through all lengths
# with the recurrent state carried; the instruction is (1, L), row-expanded to the batch by the model (:189-190).
VARLEN_CASES = {
    "varlen_128": (dict(rgb_hw=128, depth_hw=128, bert_layers=2, vla_layers=2), 2, [7, 37, 80, 123, 200, 320]),
}


def varlen_case_config(name):
    kw, batch, lens = VARLEN_CASES[name]
    return HCMConfig(**kw).validate(), batch, list(lens)


def varlen_ids(cfg, L, step):
    """(1, L) unpadded instruction: [CLS]=101, L-2 word pieces, [SEP]=102 -- the shape of `tokenizer.encode(text).ids`."""
    ids = synth.randint(f"obs/varlen/{step}", L, 1000, cfg.bert_vocab, SEED).reshape(1, L)
    ids[0, 0] = 101
    ids[0, L - 1] = 102
    return ids

SEED = 0
TAP_MAX = 16384


def case_config(name):
    kw, batch, steps, which = CASES[name]
    return HCMConfig(**kw).validate(), batch, steps, which


def step_masks(batch, step):
    """Scripted done pattern: every env starts an episode at step 0; env (step-1)%B is 'done' after step 1,
    so its mask is 0 at step 2 (hierarchical_trainer.py:1068,:1103,:1143-1159)."""
    m = np.ones((batch,), dtype=np.float32)
    if step == 0:
        m[:] = 0
    elif step == 2:
        m[(step - 1) % batch] = 0
    return m


def subsample(a):
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    stride = max(1, -(-a.size // TAP_MAX))
    return a[::stride][:TAP_MAX].copy()


def fixed_subtask(batch, step):
    """For low-level-only cases the sub-task ids are scripted instead of coming from a high-level argmax."""
    return ((np.arange(batch) + step) % 4).astype(np.int64)


# Training-path (seq_forward) cases: name -> (config kwargs, T, N)
# The reference's in-tree RNNStateEncoder.seq_forward calls `.detach()` on the unpacked hidden state
# (models/decoder/state_encoder.py:131), which is a tuple for LSTM -> AttributeError: with the in-tree copy the sequence
# path only runs for GRU, so only the GRU case has a golden; the LSTM case is checked HIP-vs-oracle (the oracle's LSTM
# cell is pinned by the single-step goldens, its scan logic by the GRU sequence golden).
SEQ_CASES = {
    "seq_T4_N2_gru": (dict(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2, rnn_type="GRU"), 4, 2),
}
SEQ_CASES_ORACLE_ONLY = {
    "seq_T4_N2_lstm": (dict(rgb_hw=128, depth_hw=128, instr_len=20, bert_layers=2), 4, 2),
}


def seq_masks(T, N):
    """(T*N,) time-major masks: every env starts at t=0; env 1 is reset at t=2 (a segment boundary for seq_forward)."""
    m = np.ones((T, N), dtype=np.float32)
    m[0, :] = 0
    if T > 2:
        m[2, 1 % N] = 0
    return m.reshape(-1)


def seq_observations(cfg, T, N):
    """T*N frames, time-major (row t*N + n); the instruction of env n repeated at every step (as the trainer's collate does)."""
    obs = synth.make_observations(cfg, T * N, step=7, seed=SEED)
    ids = synth.make_observations(cfg, N, step=0, seed=SEED)["instruction"]
    obs["instruction"] = np.tile(ids, (T, 1))
    return obs


# CMANet flat baseline: name -> (CMAConfig kwargs, batch, steps)
CMA_CASES = {
    # paper_configs/cma_robo.yaml: bidirectional LSTM instruction encoder, LSTM state encoders
    "cma_128_L20": (dict(rgb_hw=128, depth_hw=128, instr_len=20), 2, 3),
    # GRU state encoders + unidirectional instruction encoder (config/default.py:113 default)
    "cma_gru_uni_128_L12": (dict(rgb_hw=128, depth_hw=128, instr_len=12, rnn_type="GRU", bidirectional=False), 3, 3),
    # full frame size, L=80
    "cma_256_L80": (dict(), 1, 2),
    # 192-pixel depth frames: 3x3 map x 228 compression channels
    "cma_depth192_L12": (dict(rgb_hw=128, depth_hw=192, instr_len=12), 2, 2),
    # non-square RGB frames (portrait, odd pooled map sizes)
    "cma_rgb_200x152_L12": (dict(rgb_hw=200, rgb_w=152, depth_hw=128, instr_len=12), 2, 2),
    # round 6: the reference's option branches -- INSTRUCTION_ENCODER.rnn_type = "GRU" (instruction_encoder.py:42), bidirectional and not
    "cma_instr_gru_128_L12": (dict(rgb_hw=128, depth_hw=128, instr_len=12, instr_rnn="GRU"), 3, 2),
    "cma_instr_gru_uni_128_L9": (dict(rgb_hw=128, depth_hw=128, instr_len=9, instr_rnn="GRU", bidirectional=False, rnn_type="GRU"), 2, 2),
    # ... and the three ablations (cma.py:236-241): `embedding * 0` behind each encoder (an all-zero instruction masks EVERY text position)
    "cma_ablate_instr_128_L12": (dict(rgb_hw=128, depth_hw=128, instr_len=12, ablate_instruction=True), 2, 2),
    "cma_ablate_depth_128_L12": (dict(rgb_hw=128, depth_hw=128, instr_len=12, ablate_depth=True), 2, 2),
    "cma_ablate_rgb_128_L12": (dict(rgb_hw=128, depth_hw=128, instr_len=12, ablate_rgb=True), 2, 2),
}


def cma_case_config(name):
    kw, batch, steps = CMA_CASES[name]
    return CMAConfig(**kw).validate(), batch, steps
