"""Hyper-parameters of the HCM hot path.

Values restate the reference's `MODEL.*` defaults
(/root/reference/robo_vln_baselines/config/default.py:131,:156-164,:180-199 and
paper_configs/hierarchical_cma.yaml) as a plain dataclass; yacs is not used.
"""
from dataclasses import dataclass, asdict


@dataclass
class HCMConfig:
    # observation sizes (frames are NHWC)
    rgb_hw: int = 256              # RGB frame height (and width when rgb_w == 0)
    rgb_w: int = 0                 # RGB frame width; 0 = square (depth frames are square: habitat sizes its ResNet encoder from the height)
    depth_hw: int = 256            # depth frame height (and width when depth_w == 0)
    depth_w: int = 0               # depth frame width; 0 = square.  Non-square: SimpleDepthCNN only (habitat's ResNet encoder assumes a square map)
    instr_len: int = 80            # L: tokens per instruction of the synthetic workloads; an engine accepts any L <= its max_instr_len per call
    # encoders: cnn_type strings are the reference's
    rgb_encoder: str = "TorchVisionResNet50"      # or "SimpleRGBCNN" (low-level only)
    depth_encoder: str = "VlnResnetDepthEncoder"  # or "SimpleDepthCNN" (low-level only)
    rgb_out: int = 256             # RGB_ENCODER.output_size
    depth_out: int = 128           # DEPTH_ENCODER.output_size
    depth_baseplanes: int = 32     # resnet_encoders.py:19
    # Visual_Ling_Attn (default.py:156-164)
    vla_layers: int = 1
    d_model: int = 256
    vla_heads: int = 4
    d_ff: int = 1024
    vis_in: int = 256
    ins_in: int = 768
    cm_d_model: int = 256          # IMAGE_CROSS_MODAL_ENCODER.d_model (rnn input size sum only)
    # state encoder
    hidden: int = 512
    rnn_type: str = "LSTM"         # default.py:199; "GRU" selectable
    num_actions: int = 4           # high-level sub-task logits
    num_sub_tasks: int = 4
    lo_actions: int = 2            # (lin_vel, ang_vel)
    # BERT (bert-base-uncased architecture); bert_layers may be reduced in CPU tests
    bert_layers: int = 12
    bert_hidden: int = 768
    bert_heads: int = 12
    bert_inter: int = 3072
    bert_vocab: int = 30522
    bert_max_pos: int = 512
    # working ablation flags of both models: encoder output * 0 (seq2seq_highlevel_cma.py:185-188, seq2seq_lowlevel.py:132-135)
    ablate_depth: bool = False
    ablate_rgb: bool = False
    # flags the reference has but whose branches are broken
    use_prev_action: bool = False
    ablate_instruction: bool = False
    progress_monitor: bool = False

    def validate(self):
        if self.use_prev_action:
            raise ValueError("SEQ2SEQ.use_prev_action=True is a broken branch in the reference "
                             "(seq2seq_highlevel_cma.py:203-207, undefined `x`); not supported")
        if self.ablate_instruction:
            raise ValueError("ablate_instruction=True is a broken branch in the reference "
                             "(seq2seq_highlevel_cma.py:183-184); not supported")
        if self.rnn_type not in ("LSTM", "GRU"):
            raise ValueError("STATE_ENCODER.rnn_type must be LSTM or GRU")
        if self.rgb_encoder not in ("TorchVisionResNet50", "SimpleRGBCNN"):
            raise ValueError("RGB_ENCODER.cnn_type must be either 'SimpleRGBCNN' or 'TorchVisionResNet50'.")
        if self.depth_encoder not in ("VlnResnetDepthEncoder", "SimpleDepthCNN"):
            raise ValueError("DEPTH_ENCODER.cnn_type must be SimpleDepthCNN or VlnResnetDepthEncoder")
        if self.d_model % self.vla_heads:
            raise ValueError("d_model must be divisible by h")
        if self.depth_encoder == "VlnResnetDepthEncoder":
            # habitat's ResNetEncoder sizes its compression conv from (H // 2) / 32 (resnet_encoders.py:37-62): only multiples of 64
            # give the map that formula predicts (192 -> 3x3 x 228 channels, 256 -> 4x4 x 128, 320 -> 5x5 x 82)
            if self.depth_hw % 64 or not 64 <= self.depth_hw <= 1024:
                raise ValueError("depth frame size must be a multiple of 64 for the ResNet depth encoder")
            if self.depth_w and self.depth_w != self.depth_hw:
                raise ValueError("non-square depth frames: SimpleDepthCNN only (habitat's ResNetEncoder sizes itself from the frame height "
                                 "and assumes a square final map, resnet_encoders.py:37-62)")
        return self

    @property
    def rgb_shape(self):
        """(H, W) of the RGB frames."""
        return self.rgb_hw, (self.rgb_w or self.rgb_hw)

    @property
    def depth_shape(self):
        """(H, W) of the depth frames."""
        return self.depth_hw, (self.depth_w or self.depth_hw)

    @property
    def num_recurrent_layers(self):
        # state_encoder.py:41-45
        return 2 if self.rnn_type == "LSTM" else 1

    def depth_final_spatial(self):
        # habitat ResNetEncoder: spatial_size = H // 2; final = int(spatial * 1/32)
        return int((self.depth_hw // 2) / 32)

    def depth_compress_channels(self):
        fs = self.depth_final_spatial()
        return int(round(2048 / (fs * fs)))

    def to_dict(self):
        return asdict(self)


# BASELINE.json configs[0..4]
def baseline_config(idx: int) -> "HCMConfig":
    if idx == 0:   # plumbing: 128x128, L=20, N=2
        return HCMConfig(rgb_hw=128, depth_hw=128, instr_len=20, vla_layers=2)
    if idx in (1, 2):  # full HCM 256x256, L=80, N=1
        return HCMConfig()
    if idx == 3:   # SimpleDepthCNN + 1-layer VLA microbench
        return HCMConfig(depth_encoder="SimpleDepthCNN", vla_layers=1)
    if idx == 4:   # ResNet50 + N=6, L=160
        return HCMConfig(instr_len=160, vla_layers=6)
    raise IndexError(idx)


@dataclass
class CMAConfig:
    """`CMANet` flat baseline (models/cma.py:28-186) with the values of paper_configs/cma_robo.yaml over
    config/default.py:97-115,:180-216: bidirectional LSTM instruction encoder over a 2504-word vocabulary, both
    ResNet-50 encoders in spatial mode, two recurrent state encoders."""
    rgb_hw: int = 256              # RGB frame height (and width when rgb_w == 0)
    rgb_w: int = 0                 # RGB frame width; 0 = square (depth frames are square: habitat sizes its ResNet encoder from the height)
    depth_hw: int = 256
    instr_len: int = 80            # padded instruction length handed to the model (INSTRUCTION_ENCODER.max_length = 200)
    vocab_size: int = 2504         # INSTRUCTION_ENCODER.vocab_size
    embedding_size: int = 50
    instr_hidden: int = 256        # INSTRUCTION_ENCODER.hidden_size
    bidirectional: bool = True     # cma_robo.yaml
    instr_rnn: str = "LSTM"
    rgb_out: int = 256
    depth_out: int = 128
    depth_baseplanes: int = 32
    hidden: int = 512
    rnn_type: str = "LSTM"
    num_actions: int = 2           # robo_vln_trainer.py:327-331
    use_prev_action: bool = False
    rcm_state_encoder: bool = False
    progress_monitor: bool = False
    ablate_instruction: bool = False
    ablate_depth: bool = False
    ablate_rgb: bool = False
    final_state_only: bool = False   # INSTRUCTION_ENCODER.final_state_only: CMANet forces False whatever the config says (cma.py:32)

    def validate(self):
        if self.use_prev_action or self.rcm_state_encoder:
            # config/default.py:211-212: both default False, and no paper config of the reference sets either; cma.py:231-234 / :243-254 are the branches
            raise ValueError("CMA.use_prev_action / CMA.rcm_state_encoder are not built (config/default.py:211-212 default False; cma.py:231,:243)")
        if self.instr_rnn not in ("LSTM", "GRU"):
            raise ValueError("INSTRUCTION_ENCODER.rnn_type must be LSTM or GRU (instruction_encoder.py:42)")
        if self.rnn_type not in ("LSTM", "GRU"):
            raise ValueError("STATE_ENCODER.rnn_type must be LSTM or GRU")
        if self.progress_monitor:
            raise ValueError("the progress monitor is a training-only auxiliary loss (cma.py:320-329)")
        # INSTRUCTION_ENCODER.final_state_only is accepted and ignored, as in the reference: CMANet.__init__ overwrites it with False (cma.py:32)
        hh = self.hidden // 2
        if self.rgb_out > hh or self.depth_out > hh:
            # cma.py:281-286: `k, v = torch.split(kv(x), hidden_size // 2, dim=1)` over hidden/2 + output_size channels unpacks into exactly
            # two pieces only while output_size <= hidden/2 (the reference raises "too many values to unpack" otherwise)
            raise ValueError("CMANet needs RGB_ENCODER.output_size and DEPTH_ENCODER.output_size <= STATE_ENCODER.hidden_size / 2")
        return self

    @property
    def rgb_shape(self):
        """(H, W) of the RGB frames."""
        return self.rgb_hw, (self.rgb_w or self.rgb_hw)

    @property
    def instr_out(self):           # InstructionEncoder.output_size (instruction_encoder.py:49-51)
        return self.instr_hidden * (2 if self.bidirectional else 1)

    @property
    def num_recurrent_layers(self):   # cma.py:190-194: both state encoders
        return 2 * (2 if self.rnn_type == "LSTM" else 1)

    def depth_final_spatial(self):
        return int((self.depth_hw // 2) / 32)

    def depth_compress_channels(self):
        fs = self.depth_final_spatial()
        return int(round(2048 / (fs * fs)))

    def to_dict(self):
        return asdict(self)
