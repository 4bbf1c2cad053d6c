"""The WordPiece tokenizer of the input pipeline (robo-vln_amd/tokenizer.py) against `tokenizers.BertWordPieceTokenizer` -- the
library the reference calls in transform_obs (common/utils.py:87-107) -- on a synthetic vocabulary (the real
bert-base-uncased vocabulary cannot be downloaded here).  The library is the test oracle only."""
import os
import tempfile

import numpy as np
import pytest

from robo_vln_amd.tokenizer import InstructionCache, WordPieceTokenizer

WORDS = ["walk", "forward", "and", "turn", "left", "right", "at", "the", "table", "stop", "near", "door", "go", "past", "kitchen",
         "room", "bed", "a", "i", "in", "to", "s", "ing", "ed", "un", "re", "able", "cafe", "naive", "resume", "chair", "2", "10", "3rd",
         "wait", "then", "you", "see", "stairs", "up", "down", "hall", "way", "hallway", "o", "clock", "中", "国"]
SUFFIX = ["##s", "##ing", "##ed", "##er", "##way", "##room", "##able", "##ly", "##e", "##a", "##1", "##0", "##rd", "##k", "##w", "##al"]
PUNCT = list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~") + ["—", "…", "“", "”"]


def _vocab_file(d):
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + SUFFIX + PUNCT + list("bcdefghjklmnpqrtuvwxyz")
    path = os.path.join(d, "vocab.txt")
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(toks) + "\n")
    return path


SENTENCES = [
    "Walk forward and turn left at the table.",
    "Go past the kitchen, then turn RIGHT; stop near the door!",
    "walking  forwards\tand   re-turning to the bedroom",
    "Café résumé naïve — stop…",
    "unknownword zzzzqqq walked walker walks",
    "turn left at 10 o'clock, 3rd door (the hallway)",
    "中国 room walk中forward",
    "a" * 120 + " stop",
    "",
    "   ",
    "stop\x00now�\x07 wait",
    "“go” up-stairs & down_stairs #2",
]


def test_matches_the_reference_library_on_a_synthetic_vocab():
    tk = pytest.importorskip("tokenizers")
    with tempfile.TemporaryDirectory() as d:
        path = _vocab_file(d)
        ref = tk.BertWordPieceTokenizer(path, lowercase=True)
        mine = WordPieceTokenizer(path, lowercase=True)
        for s in SENTENCES:
            assert mine.encode(s) == ref.encode(s).ids, s
        rng = np.random.default_rng(0)
        alphabet = list("abcdefghijklmnopqrstuvwxyz") + WORDS + [" ", " ", ",", ".", "'", "-", "é", "Ü", "中"]
        for _ in range(300):
            s = "".join(rng.choice(alphabet) for _ in range(int(rng.integers(1, 40))))
            assert mine.encode(s) == ref.encode(s).ids, s


def test_padding_truncation_and_cache():
    with tempfile.TemporaryDirectory() as d:
        tok = WordPieceTokenizer(_vocab_file(d))
    ids = tok.encode_padded("walk forward and stop", 12)
    assert ids.dtype == np.int32 and ids.shape == (12,)
    assert ids[0] == tok.cls and ids[5] == tok.sep and (ids[6:] == 0).all()
    short = tok.encode_padded("walk forward and turn left at the table", 5)
    assert short[0] == tok.cls and short[-1] == tok.sep and (short != 0).all()
    cache = InstructionCache(tok, 12)
    a = cache.get("ep1", "walk forward and stop")
    b = cache.get("ep1")
    assert a is b and cache.hits == 1 and cache.misses == 1
    with pytest.raises(KeyError):
        cache.get("ep2")
    with pytest.raises(ValueError):
        WordPieceTokenizer({"a": 0})
