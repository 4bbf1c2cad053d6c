"""The WordPiece tokenizer of the input pipeline (robo-vln_amd/tokenizer.py) against `tokenizers.BertWordPieceTokenizer` -- the
library the reference calls in transform_obs (common/utils.py:87-107) -- on a synthetic vocabulary everywhere, and in the
build container also on the reference's own `vocab_files/bert-base-uncased-vocab.txt`.  The library is the test oracle only."""
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


REAL_VOCAB = "/root/reference/vocab_files/bert-base-uncased-vocab.txt"


@pytest.mark.skipif(not os.path.exists(REAL_VOCAB), reason="needs the reference checkout (build container only)")
def test_matches_the_reference_library_on_the_real_vocabulary():
    """The vocabulary file the reference loads (common/utils.py:105), read in place: token-for-token equality with the
    `tokenizers` library on navigation instructions, unicode / punctuation edge cases and random strings."""
    tk = pytest.importorskip("tokenizers")
    ref = tk.BertWordPieceTokenizer(REAL_VOCAB, lowercase=True)
    mine = WordPieceTokenizer(REAL_VOCAB, lowercase=True)
    assert (mine.cls, mine.sep, mine.pad, mine.unk) == (101, 102, 0, 100)
    texts = SENTENCES + [
        "Exit the bedroom and turn left. Walk straight passing the gray couch and stop near the rug.",
        "Go up the stairs, turn right at the top and wait by the bathroom door on your left-hand side.",
        "Walk past the dining table & chairs; enter the 2nd doorway (kitchen) — stop in front of the refrigerator!",
        "Turn around 180° and go downstairs... then take a slight right toward the potted plants.",
        "ﬁnd the café's naïve façade; ÜBER-straße №5 ½ way", "tab\there\nnewline\rreturn", "emoji 🙂 stop", "ⅷ Ⅻ ① ②", "日本語のテキスト 한국어 텍스트",
    ]
    for s in texts:
        assert mine.encode(s) == ref.encode(s).ids, s
    rng = np.random.default_rng(1)
    vocab_words = [w for w in list(mine.vocab)[2000:30000:37] if not w.startswith("##")]
    pieces = vocab_words + list("abcdefghijklmnopqrstuvwxyz0123456789") + [" ", " ", " ", ",", ".", "'", "-", "é", "ß", "中", "!", "?"]
    for _ in range(500):
        s = "".join(rng.choice(pieces) for _ in range(int(rng.integers(1, 30))))
        assert mine.encode(s) == ref.encode(s).ids, s


def test_unpadded_cache_and_pad_batch():
    """The default cache hands out what the reference's loop feeds the model -- the unpadded ids -- and pad_batch builds the
    (ids, lengths) pair of a ragged batched call."""
    from robo_vln_amd.tokenizer import pad_batch
    with tempfile.TemporaryDirectory() as d:
        tok = WordPieceTokenizer(_vocab_file(d))
    cache = InstructionCache(tok)
    a = cache.get("ep1", "walk forward and stop")
    assert a.dtype == np.int32 and a.tolist() == tok.encode("walk forward and stop") and a[0] == tok.cls and a[-1] == tok.sep
    b = cache.get("ep2", "turn left")
    ids, lens = pad_batch([a, b])
    assert ids.shape == (2, len(a)) and lens.tolist() == [len(a), len(b)] and lens.dtype == np.int32
    assert ids[1, :len(b)].tolist() == b.tolist() and (ids[1, len(b):] == 0).all()
