"""Instruction tokenisation in front of the hot path: the build's counterpart of the WordPiece step of
`transform_obs` (robo_vln_baselines/common/utils.py:18-20,:87-107), which constructs a `tokenizers.BertWordPieceTokenizer` from
`vocab_files/bert-base-uncased-vocab.txt` and encodes the instruction text on EVERY environment step
(hierarchical_trainer.py:1193-1196).  Here: a self-contained BERT tokenizer (basic tokenisation + greedy longest-match
WordPiece, same defaults: lowercase, accents stripped, CJK characters isolated, [UNK] for words over 100 characters), built once,
and an episode cache so that an instruction is tokenised once per episode: unpadded by default (what the reference feeds the
model), or padded to a common length together with per-environment token counts for batched rollouts (`pad_batch`).
"""
import unicodedata
from typing import Dict, Iterable, List, Optional, Union

import numpy as np


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or
            0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    def __init__(self, vocab: Union[str, Dict[str, int], Iterable[str]], lowercase: bool = True, unk_token="[UNK]", cls_token="[CLS]",
                 sep_token="[SEP]", pad_token="[PAD]", prefix="##", max_input_chars_per_word=100):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        elif not isinstance(vocab, dict):
            vocab = {tok: i for i, tok in enumerate(vocab)}
        self.vocab = vocab
        self.lowercase = lowercase
        self.prefix = prefix
        self.max_chars = max_input_chars_per_word
        for t in (unk_token, cls_token, sep_token):
            if t not in vocab:
                raise ValueError(f"vocabulary lacks the special token {t}")
        self.unk, self.cls, self.sep = vocab[unk_token], vocab[cls_token], vocab[sep_token]
        self.pad = vocab.get(pad_token, 0)

    # ---- BertNormalizer + BertPreTokenizer
    def _words(self, text: str) -> List[str]:
        out = []
        for ch in text:                                        # clean_text
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                out.append(" ")
            elif _is_cjk(cp):                                  # handle_chinese_chars
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        text = "".join(out)
        if self.lowercase:                                     # strip_accents follows lowercase when left unset
            text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn").lower()
        words = []
        for tok in text.split():
            cur = []
            for ch in tok:
                if _is_punctuation(ch):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(ch)
                else:
                    cur.append(ch)
            if cur:
                words.append("".join(cur))
        return words

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                piece = word[start:end]
                if start > 0:
                    piece = self.prefix + piece
                if piece in self.vocab:
                    cur = self.vocab[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str) -> List[int]:
        """[CLS] wordpieces [SEP] -- what `tokenizer.encode(sentence).ids` returns in the reference (utils.py:18-20)."""
        ids = [self.cls]
        for w in self._words(text):
            ids.extend(self._wordpiece(w))
        ids.append(self.sep)
        return ids

    def encode_padded(self, text: str, length: int) -> np.ndarray:
        """Zero-padded (pad id) / truncated to `length`; a truncated sequence keeps [SEP] as its last token."""
        ids = self.encode(text)
        if len(ids) > length:
            ids = ids[:length - 1] + [self.sep]
        out = np.full((length,), self.pad, dtype=np.int32)
        out[:len(ids)] = ids
        return out


class InstructionCache:
    """Tokenise an episode's instruction once (the reference re-tokenises every step).

    instr_len=None (default): `get` returns the UNPADDED ids `tokenizer.encode(text).ids`, exactly what the reference's eval loop
    hands to the model (common/utils.py:18-20; `max_seq_length` is accepted and ignored there).  BERT runs without an attention
    mask and the cross-modal poolers average over all positions, so padding changes the model's output: pass these ids as a
    (1, L) instruction.  instr_len=N: zero-padded / truncated to N, for batched rollouts that need one common L -- hand the
    engine `instruction_lengths` (see `pad_batch`) so that every environment still gets its unpadded result."""

    def __init__(self, tokenizer: WordPieceTokenizer, instr_len: Optional[int] = None, capacity: int = 4096):
        self.tok, self.L, self.cap = tokenizer, instr_len, capacity
        self._d: Dict[object, np.ndarray] = {}
        self.hits = self.misses = 0

    def get(self, episode_id, text: Optional[str] = None) -> np.ndarray:
        ids = self._d.get(episode_id)
        if ids is None:
            if text is None:
                raise KeyError(episode_id)
            ids = (np.asarray(self.tok.encode(text), dtype=np.int32) if self.L is None
                   else self.tok.encode_padded(text, self.L))
            if len(self._d) >= self.cap:
                self._d.pop(next(iter(self._d)))
            self._d[episode_id] = ids
            self.misses += 1
        else:
            self.hits += 1
        return ids


def pad_batch(id_lists, pad_id: int = 0):
    """Ragged token-id lists of B environments -> (ids (B, Lmax) int32 zero-padded, lengths (B,) int32) for
    observations["instruction"] / observations["instruction_lengths"] of a batched engine call."""
    lens = np.asarray([len(x) for x in id_lists], dtype=np.int32)
    out = np.full((len(id_lists), int(lens.max())), pad_id, dtype=np.int32)
    for i, x in enumerate(id_lists):
        out[i, :len(x)] = np.asarray(x, dtype=np.int32)
    return out, lens
