"""Host-side tokenizers either side of the hot path.

* :class:`GPT2BPE` -- GPT-2 byte-level BPE (``GPT2Tokenizer.from_pretrained('gpt2')`` in the reference,
  predictions_runner.py:416; the decode functions need ``encode('.')[0]`` and ``decode(ids)``,
  gpt2_prefix_eval.py:54,112,192).  Needs the model's ``vocab.json`` + ``merges.txt``.
* :class:`ClipBPE` -- the lower-cased ``</w>`` BPE of openai/CLIP (``clip.tokenize``, reference
  embeddings_generator.py:80-85, predictions_runner.py:217): SOT + ids + EOT, zero padded to 77, ``RuntimeError``
  when a text is too long (the reference retries on ``caption[:100]``).  Needs ``bpe_simple_vocab_16e6.txt(.gz)``
  (or any merges list).  ``ftfy`` is not installed here, so its mojibake repair step is skipped.

Neither vocabulary file exists offline; the algorithms are pinned in tests against the ``transformers``
implementations on synthetic vocabularies.
"""
from __future__ import annotations

import gzip
import html
import json
from functools import lru_cache
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import regex as re
import torch


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """the reversible byte <-> printable-unicode table shared by GPT-2 and CLIP BPE"""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word: Tuple[str, ...]):
    return set(zip(word[:-1], word[1:]))


class _BPE:
    def __init__(self, ranks: Dict[Tuple[str, str], int]):
        self.ranks = ranks
        self.cache: Dict[str, Tuple[str, ...]] = {}

    def merge(self, word: Tuple[str, ...], key: str) -> Tuple[str, ...]:
        if key in self.cache:
            return self.cache[key]
        pairs = _pairs(word)
        while pairs:
            bigram = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if bigram not in self.ranks:
                break
            first, second = bigram
            new: List[str] = []
            i = 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        self.cache[key] = word
        return word


class GPT2BPE:
    PAT = re.compile(r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+""")

    def __init__(self, vocab_file: str, merges_file: str):
        with open(vocab_file, encoding="utf-8") as f:
            self.encoder: Dict[str, int] = json.load(f)
        self.decoder = {v: k for k, v in self.encoder.items()}
        with open(merges_file, encoding="utf-8") as f:
            lines = [l for l in f.read().split("\n") if l and not l.startswith("#version")]
        self.bpe = _BPE({tuple(l.split()): i for i, l in enumerate(lines)})
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in self.PAT.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe.merge(tuple(tok), tok))
        return ids

    @staticmethod
    def clean_up_tokenization(out_string: str) -> str:
        """transformers' ``PreTrainedTokenizerBase.clean_up_tokenization`` (the default of ``tokenizer.decode`` in the
        pinned 4.24, which the reference calls at gpt2_prefix_eval.py:112,192): spaces before punctuation and
        abbreviated forms are removed."""
        return (out_string.replace(" .", ".").replace(" ?", "?").replace(" !", "!").replace(" ,", ",")
                .replace(" ' ", "'").replace(" n't", "n't").replace(" 'm", "'m").replace(" 's", "'s")
                .replace(" 've", "'ve").replace(" 're", "'re"))

    def decode(self, ids: Iterable[int], clean_up_tokenization_spaces: bool = True) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        text = bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace")
        return self.clean_up_tokenization(text) if clean_up_tokenization_spaces else text


class ClipBPE:
    PAT = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                     re.IGNORECASE)

    def __init__(self, bpe_path_or_merges: Union[str, Sequence[Tuple[str, str]]], n_merges: int = 49152 - 256 - 2):
        if isinstance(bpe_path_or_merges, str):
            op = gzip.open if bpe_path_or_merges.endswith(".gz") else open
            with op(bpe_path_or_merges, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [tuple(m.split()) for m in lines[1:1 + n_merges] if m]
        else:
            merges = [tuple(m) for m in bpe_path_or_merges]
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.bpe = _BPE({m: i for i, m in enumerate(merges)})
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    @staticmethod
    def clean(text: str) -> str:
        text = html.unescape(html.unescape(text))        # (ftfy.fix_text would run first in openai/CLIP)
        return re.sub(r"\s+", " ", text.strip()).strip().lower()

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in self.PAT.findall(self.clean(text)):
            if tok in ("<|startoftext|>", "<|endoftext|>"):
                ids.append(self.encoder[tok])
                continue
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            word = tuple(tok[:-1]) + (tok[-1] + "</w>",)
            ids.extend(self.encoder[t] for t in self.bpe.merge(word, tok))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """``clip.tokenize``: int [N, context_length], SOT ... EOT, zero padded"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.int32)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out
