"""CLIP byte-level BPE tokenizer (host side, pure Python) for the enhancer's prompts.

The reference tokenizes with transformers' ``CLIPTokenizer`` loaded from the ``tokenizer/`` folder of the I2VGen-XL checkpoint
(pipeline_i2vgen_xl.py:213-231, 291-300: ``tokenizer(prompt, padding="max_length", max_length=model_max_length, truncation=True)``).
This module reads the same files (``vocab.json``, ``merges.txt``, optionally ``special_tokens_map.json`` / ``tokenizer_config.json`` for the
pad token and the maximum length) and restates the published algorithm: NFC normalisation, whitespace collapsing, lower-casing, the CLIP
pre-tokenisation pattern, byte-to-unicode mapping, greedy lowest-rank BPE merges with the ``</w>`` end-of-word suffix, ``<|startoftext|>`` /
``<|endoftext|>`` framing, truncation to the model length and padding.  Pinned on CPU against transformers' own CLIPTokenizer on a
synthetic vocabulary (tests/test_host_logic.py) -- no vocabulary file is available offline.  (transformers 4.40's slow tokenizer
additionally runs ``ftfy`` / BasicTokenizer text cleaning, which only matters for mojibake, HTML entities and CJK spacing.)
"""
import json
import os
import unicodedata

import regex

PATTERN = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""")


def bytes_to_unicode():
    """The GPT-2 / CLIP reversible byte -> printable-unicode table."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


class CLIPBPETokenizer:
    def __init__(self, vocab, merges, bos_token="<|startoftext|>", eos_token="<|endoftext|>", pad_token="<|endoftext|>", unk_token="<|endoftext|>",
                 model_max_length=77):
        """vocab: dict token -> id (or path to vocab.json); merges: list of "a b" strings / pairs (or path to merges.txt)."""
        if isinstance(vocab, (str, os.PathLike)):
            with open(vocab, encoding="utf-8") as f:
                vocab = json.load(f)
        if isinstance(merges, (str, os.PathLike)):
            with open(merges, encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [ln for ln in lines[1:] if ln.strip()] if lines and lines[0].startswith("#version") else [ln for ln in lines if ln.strip()]
        self.encoder = dict(vocab)
        self.ranks = {tuple(m.split()) if isinstance(m, str) else tuple(m): i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.bos, self.eos, self.unk = self.encoder[bos_token], self.encoder[eos_token], self.encoder[unk_token]
        self.pad = self.encoder[pad_token]
        # special tokens are cut out of the RAW text before normalisation and map straight to their ids, like transformers' added-token trie;
        # with the LAION text towers' pad token "!" that makes every literal "!" token id 0 instead of a BPE symbol
        self.special = {t: self.encoder[t] for t in (bos_token, eos_token, pad_token, unk_token)}
        self._split = regex.compile("(" + "|".join(regex.escape(t) for t in sorted(self.special, key=len, reverse=True)) + ")")
        self.model_max_length = model_max_length
        self._cache = {}

    @classmethod
    def from_pretrained(cls, folder):
        """folder: the checkpoint's ``tokenizer`` directory (vocab.json, merges.txt [, special_tokens_map.json, tokenizer_config.json])."""
        kw = {}
        for name in ("tokenizer_config.json", "special_tokens_map.json"):
            path = os.path.join(folder, name)
            if os.path.exists(path):
                with open(path, encoding="utf-8") as f:
                    cfg = json.load(f)
                for k in ("bos_token", "eos_token", "pad_token", "unk_token"):
                    if k in cfg:
                        kw[k] = cfg[k]["content"] if isinstance(cfg[k], dict) else cfg[k]
                if isinstance(cfg.get("model_max_length"), int) and cfg["model_max_length"] < 10 ** 6:
                    kw["model_max_length"] = cfg["model_max_length"]
        return cls(os.path.join(folder, "vocab.json"), os.path.join(folder, "merges.txt"), **kw)

    def bpe(self, token):
        """token: string of byte-level characters (one pre-token) -> list of vocabulary symbols."""
        if token in self._cache:
            return self._cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best = min(((self.ranks.get((a, b), float("inf")), i) for i, (a, b) in enumerate(zip(word, word[1:]))))
            if best[0] == float("inf"):
                break
            a, b = word[best[1]], word[best[1] + 1]
            out, i = [], 0
            while i < len(word):                       # merge EVERY occurrence of the best pair, left to right
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self._cache[token] = word
        return word

    def tokenize(self, text):
        ids = []
        for piece in self._split.split(text):
            if piece in self.special:
                ids.append(self.special[piece])
                continue
            piece = " ".join(unicodedata.normalize("NFC", piece).split()).lower()
            for tok in PATTERN.findall(piece):
                sym = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
                ids.extend(self.encoder.get(s, self.unk) for s in self.bpe(sym))
        return ids

    def __call__(self, text, max_length=None, padding="max_length", truncation=True):
        """-> dict(input_ids=[...], attention_mask=[...]) for one string, or lists of those for a list of strings."""
        if isinstance(text, (list, tuple)):
            rows = [self(t, max_length, padding, truncation) for t in text]
            return dict(input_ids=[r["input_ids"] for r in rows], attention_mask=[r["attention_mask"] for r in rows])
        n = max_length or self.model_max_length
        ids = self.tokenize(text)
        if truncation and len(ids) > n - 2:
            ids = ids[: n - 2]
        ids = [self.bos] + ids + [self.eos]
        mask = [1] * len(ids)
        if padding == "max_length" and len(ids) < n:
            mask += [0] * (n - len(ids))
            ids += [self.pad] * (n - len(ids))
        return dict(input_ids=ids, attention_mask=mask)

    def input_ids(self, text, device="cpu"):
        """[1, model_max_length] int64 tensor for one prompt: what `EnhanceCodec.set_prompts_from_ids` / `CLIPTextTower` consume."""
        import torch
        return torch.tensor([self(text)["input_ids"]], dtype=torch.long, device=device)
