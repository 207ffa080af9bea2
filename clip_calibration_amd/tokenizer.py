"""Byte-level BPE tokenizer and prompt packing (reference clip/simple_tokenizer.py:62-132, clip/clip.py:188-224,
trainers/classification/zsclip.py:23-39) -- SURVEY §8(f) row f-3.  Host-side, runs once per class list.

Own implementation of the published CLIP/GPT-2 byte-level BPE; the 49 152-entry merge table is OpenAI's data file
(``bpe_simple_vocab_16e6.txt.gz``) and is NOT vendored here: pass its path, or set ``CLIP_BPE_VOCAB``.  Tests run on
a sparse ``{(a, b): rank}`` table recorded from the reference tokenizer (tests/golden/tokenizer_cases.json), which
yields the same ids as the full table for the recorded prompts because ids are derived from ranks.

``ftfy`` is not installed in this image; text is HTML-unescaped and whitespace-normalised only, which is exact for the
ASCII class names and templates the reference uses.
"""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex
import torch

N_MERGES = 49152 - 256 - 2          # merges kept by the reference (simple_tokenizer.py:66)
SOT_TEXT, EOT_TEXT = "<|startoftext|>", "<|endoftext|>"

# dataset -> prompt template (zsclip.py:23-39; plain data)
CUSTOM_TEMPLATES = {
    "OxfordPets": "a photo of a {}, a type of pet.",
    "OxfordFlowers": "a photo of a {}, a type of flower.",
    "FGVCAircraft": "a photo of a {}, a type of aircraft.",
    "DescribableTextures": "{} texture.",
    "EuroSAT": "a centered satellite photo of {}.",
    "StanfordCars": "a photo of a {}.",
    "Food101": "a photo of {}, a type of food.",
    "SUN397": "a photo of a {}.",
    "Caltech101": "a photo of a {}.",
    "UCF101": "a photo of a person doing {}.",
    "ImageNet": "a photo of a {}.",
    "ImageNetSketch": "a photo of a {}.",
    "ImageNetV2": "a photo of a {}.",
    "ImageNetA": "a photo of a {}.",
    "ImageNetR": "a photo of a {}.",
}

_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                       regex.IGNORECASE)


def _byte_alphabet() -> List[str]:
    """Printable stand-in character for each of the 256 byte values: bytes that are already printable Latin-1 map to
    themselves, the other 68 map to code points 256, 257, ... in byte order (the GPT-2 byte table)."""
    keep = set(range(ord("!"), ord("~") + 1)) | set(range(0xA1, 0xAC + 1)) | set(range(0xAE, 0xFF + 1))
    table, extra = [], 0
    for b in range(256):
        if b in keep:
            table.append(chr(b))
        else:
            table.append(chr(256 + extra))
            extra += 1
    return table


def _base_symbols() -> List[str]:
    """Vocabulary order of the 256 single-byte symbols: printable ranges first, then the remapped ones."""
    alpha = _byte_alphabet()
    order = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    order += [b for b in range(256) if b not in set(order)]
    return [alpha[b] for b in order]


class ClipTokenizer:
    """ids: 0..255 byte symbols, 256..511 the same with the end-of-word marker, 512 + rank for merge `rank`,
    49406 / 49407 start / end of text."""

    def __init__(self, bpe_path: Optional[str] = None, merges: Optional[Dict[Tuple[str, str], int]] = None):
        if merges is None:
            bpe_path = bpe_path or os.environ.get("CLIP_BPE_VOCAB")
            if not bpe_path or not os.path.exists(bpe_path):
                raise FileNotFoundError(
                    "CLIP BPE merge table not found: pass bpe_path or set CLIP_BPE_VOCAB to OpenAI's "
                    "bpe_simple_vocab_16e6.txt.gz (the data file is not vendored in this repository)")
            opener = gzip.open if bpe_path.endswith(".gz") else open
            with opener(bpe_path, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")[1:N_MERGES + 1]
            merges = {tuple(line.split()): rank for rank, line in enumerate(lines)}
        self.ranks: Dict[Tuple[str, str], int] = dict(merges)
        self._alpha = _byte_alphabet()
        base = _base_symbols()
        self._id: Dict[str, int] = {s: i for i, s in enumerate(base)}
        self._id.update({s + "</w>": 256 + i for i, s in enumerate(base)})
        for (a, b), rank in self.ranks.items():
            self._id[a + b] = 512 + rank
        self.sot, self.eot = 512 + N_MERGES, 512 + N_MERGES + 1
        self._id[SOT_TEXT], self._id[EOT_TEXT] = self.sot, self.eot
        self._sym = {i: s for s, i in self._id.items()}
        self._unalpha = {c: b for b, c in enumerate(self._alpha)}
        self._cache: Dict[str, List[str]] = {}

    # ---- BPE of one pre-token ---------------------------------------------------------------------------------
    def _merge_word(self, word: str) -> List[str]:
        if word in (SOT_TEXT, EOT_TEXT):
            return [word]
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best_rank, best_pair = None, None
            for pair in zip(parts, parts[1:]):
                rank = self.ranks.get(pair)
                if rank is not None and (best_rank is None or rank < best_rank):
                    best_rank, best_pair = rank, pair
            if best_pair is None:
                break
            a, b = best_pair
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[word] = parts
        return parts

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for piece in _SPLIT.findall(text):
            word = "".join(self._alpha[b] for b in piece.encode("utf-8")) if piece not in (SOT_TEXT, EOT_TEXT) else piece
            ids.extend(self._id[s] for s in self._merge_word(word))
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        text = "".join(self._sym[int(i)] for i in ids)
        text = text.replace(SOT_TEXT, "").replace(EOT_TEXT, "")
        return bytearray(self._unalpha[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    # ---- clip.tokenize ----------------------------------------------------------------------------------------
    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.LongTensor:
        """[SOT] + bpe(text) + [EOT], zero padded to context_length (clip/clip.py:188-224); argmax of a row is its EOT."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            toks = [self.sot] + self.encode(t) + [self.eot]
            if len(toks) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                toks = toks[:context_length]
                toks[-1] = self.eot
            out[i, : len(toks)] = torch.tensor(toks)
        return out


def zeroshot_prompts(classnames: Sequence[str], dataset: str = "ImageNet") -> List[str]:
    """``temp.format(c.replace("_", " "))`` (zsclip.py:84-85)."""
    temp = CUSTOM_TEMPLATES[dataset]
    return [temp.format(c.replace("_", " ")) for c in classnames]


def coop_prompts(classnames: Sequence[str], n_ctx: int) -> List[str]:
    """``"X X ... X classname."`` placeholders for the learnable context (coop.py:101-108)."""
    prefix = " ".join(["X"] * n_ctx)
    return [prefix + " " + c.replace("_", " ") + "." for c in classnames]
