"""TEST INFRASTRUCTURE — CPU restatement of the tokenizer side of the reference's pair assembly.

Only tests/ may import this module; the product path (`rmr_amd`) never does.

What it restates
----------------
`prepare_full_context_inputs` (/root/reference/src/models/rerank/utils.py:129-167) calls three methods of the FLMR
query tokenizer, which is a plain `BertTokenizer` subclass that does not override them
(/root/reference/src/models/flmr/models/flmr/tokenization_flmr.py:148-250 only overrides `__call__`):
`encode(text, add_special_tokens=False, max_length=n, truncation=True)`, `decode(ids)` and
`batch_encode_plus(pairs, add_special_tokens=True, padding="max_length", truncation=True, max_length=L)`.
Their arithmetic lives in the un-vendored dependency `transformers==4.38.2` (reference README.md:90-91), slow
(Python) tokenizer classes; the published algorithm restated here, function by function:
  * `PreTrainedTokenizer.tokenize` (tokenization_utils.py): per-character lower-casing outside special tokens, split on
    the special tokens, `_tokenize` on the rest;
  * `BasicTokenizer` (models/bert/tokenization_bert.py): `_clean_text`, `_tokenize_chinese_chars`, whitespace split,
    `lower` + `_run_strip_accents` (NFD, drop Mn), `_run_split_on_punc`;
  * `WordpieceTokenizer.tokenize`: greedy longest-match-first, `max_input_chars_per_word = 100`;
  * `_decode` / `convert_tokens_to_string` / `clean_up_tokenization`;
  * `prepare_for_model` + `truncate_sequences` (LONGEST_FIRST) + right padding.
One documented omission: `BasicTokenizer.tokenize` NFC-normalises before the whitespace split; NFD follows for every
token, so the result only differs where lower-casing does not commute with normalisation (no such case in the tests).

Pin: `tests/test_pair_tokenizer_cpu.py` checks this restatement against the `BertTokenizer` of the installed
transformers (5.x, Rust-backed) on a seeded corpus that avoids the two places where that implementation differs from
4.38.2's slow one (context-sensitive final sigma; the extra " do not" -> " don't" decode rule).
"""
from __future__ import annotations

import unicodedata
from typing import Dict, List, Sequence, Tuple

SPECIAL = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")


def _is_whitespace(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_chinese_char(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or (0x2A700 <= cp <= 0x2B73F)
            or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF) or (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


class BertTokenizerOracle:
    def __init__(self, vocab: Sequence[str], do_lower_case: bool = True):
        self.vocab = {t: i for i, t in enumerate(vocab)}
        self.inv = list(vocab)
        self.do_lower_case = do_lower_case
        self.unk, self.cls, self.sep, self.pad = (self.vocab[t] for t in ("[UNK]", "[CLS]", "[SEP]", "[PAD]"))
        self.special = [t for t in SPECIAL if t in self.vocab]

    # ---- PreTrainedTokenizer.tokenize
    def _split_special(self, text: str) -> List[Tuple[str, bool]]:
        out, i, cur = [], 0, []
        while i < len(text):
            hit = None
            for s in self.special:                 # the trie returns the longest match starting here; none is a prefix of another
                if text.startswith(s, i):
                    hit = s
                    break
            if hit:
                if cur:
                    out.append(("".join(cur), False))
                    cur = []
                out.append((hit, True))
                i += len(hit)
            else:
                cur.append(text[i])
                i += 1
        if cur:
            out.append(("".join(cur), False))
        return out

    def tokenize(self, text: str) -> List[str]:
        toks: List[str] = []
        for seg, is_special in self._split_special(text):
            if is_special:
                toks.append(seg)
                continue
            if self.do_lower_case:
                seg = "".join(c.lower() for c in seg)          # re.sub(... (.+?) -> .lower()): one character at a time
            for bt in self._basic(seg):
                toks.extend(self._wordpiece(bt))
        return toks

    # ---- BasicTokenizer
    def _basic(self, text: str) -> List[str]:
        cleaned = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            cleaned.append(" " if _is_whitespace(ch) else ch)
        spaced = []
        for ch in cleaned:
            if _is_chinese_char(ord(ch)):
                spaced += [" ", ch, " "]
            else:
                spaced.append(ch)
        out: List[str] = []
        for token in "".join(spaced).split():
            if self.do_lower_case:
                token = token.lower()
                token = "".join(c for c in unicodedata.normalize("NFD", token) if unicodedata.category(c) != "Mn")
            cur: List[str] = []
            for ch in token:
                if _is_punctuation(ch):
                    if cur:
                        out.append("".join(cur))
                        cur = []
                    out.append(ch)
                else:
                    cur.append(ch)
            if cur:
                out.append("".join(cur))
        return " ".join(out).split()

    # ---- WordpieceTokenizer
    def _wordpiece(self, token: str) -> List[str]:
        chars = list(token)
        if len(chars) > 100:
            return ["[UNK]"]
        out, start = [], 0
        while start < len(chars):
            end, cur = len(chars), None
            while start < end:
                sub = "".join(chars[start:end])
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return ["[UNK]"]
            out.append(cur)
            start = end
        return out

    # ---- encode / decode
    def encode(self, text: str, max_length: int | None = None) -> List[int]:
        ids = [self.vocab.get(t, self.unk) for t in self.tokenize(text)]
        return ids if max_length is None else ids[:max_length]

    def decode(self, ids: Sequence[int]) -> str:
        s = " ".join(self.inv[i] for i in ids).replace(" ##", "").strip()
        for a, b in ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"),
                     (" 's", "'s"), (" 've", "'ve"), (" 're", "'re")):
            s = s.replace(a, b)
        return s

    def encode_pair(self, a: str, b: str, max_length: int) -> Tuple[List[int], List[int], List[int]]:
        ia, ib = self.encode(a), self.encode(b)
        over = len(ia) + len(ib) + 3 - max_length
        for _ in range(max(0, over)):                        # truncate_sequences, LONGEST_FIRST
            if len(ia) > len(ib):
                ia = ia[:-1]
            else:
                ib = ib[:-1]
        ids = [self.cls] + ia + [self.sep] + ib + [self.sep]
        tt = [0] * (len(ia) + 2) + [1] * (len(ib) + 1)
        am = [1] * len(ids)
        padn = max_length - len(ids)
        return ids + [self.pad] * padn, am + [0] * padn, tt + [0] * padn


def prepare_full_context_inputs(query_text_sequences: Sequence[str], context_text_sequences: Sequence[str],
                                tok: BertTokenizerOracle, max_query_length: int, max_context_length: int,
                                max_decoder_source_length: int, docs_per_query: int) -> Dict[str, List[List[int]]]:
    """utils.py:129-167 on top of the restated tokenizer."""
    tq = [tok.decode(tok.encode(t, max_query_length)) for t in query_text_sequences]
    tc = [tok.decode(tok.encode(t, max_context_length)) for t in context_text_sequences]
    ids, am, tt = [], [], []
    for i, q in enumerate(tq):
        for j in range(docs_per_query):
            a, b, c = tok.encode_pair(q, tc[i * docs_per_query + j], max_decoder_source_length)
            ids.append(a)
            am.append(b)
            tt.append(c)
    return {"input_ids": ids, "attention_mask": am, "token_type_ids": tt}
