"""Pair-input assembly for the full-context reranker (host side, CPU string work).

Restates `prepare_full_context_inputs` (/root/reference/src/models/rerank/utils.py:129-167): truncate the
query to `max_query_length` tokens and every context to `max_context_length` tokens by an
encode->decode round trip (no special tokens), pair them query-major, then `batch_encode_plus` the
(query, context) pairs with special tokens, padded/truncated to `max_decoder_source_length` — which yields
input_ids, attention_mask and pair-encoding token_type_ids.  Any HF-style tokenizer object works; the build
environment has no vocab file, so tests use a small locally generated WordPiece vocab.
"""
from __future__ import annotations

from typing import Dict, List

import torch


def truncate_and_pair(query_text_sequences: List[str], context_text_sequences: List[str], tokenizer, max_query_length: int,
                      max_context_length: int, docs_per_query: int) -> List[tuple]:
    """utils.py:131-153: queries / contexts cut to their token budgets by an encode -> decode round trip (no special tokens),
    then paired query-major.  Pinned to the reference's own loop executed from source (tests/golden/reference_fn.npz `pfc.*`)."""
    def clip(text: str, n: int) -> str:
        return tokenizer.decode(tokenizer.encode(text, add_special_tokens=False, max_length=n, truncation=True))

    queries = [clip(t, max_query_length) for t in query_text_sequences]
    contexts = [clip(t, max_context_length) for t in context_text_sequences]
    return [(q, contexts[i * docs_per_query + j]) for i, q in enumerate(queries) for j in range(docs_per_query)]


def prepare_full_context_inputs(query_text_sequences: List[str], context_text_sequences: List[str], tokenizer,
                                max_query_length: int, max_context_length: int, max_decoder_source_length: int,
                                docs_per_query: int) -> Dict[str, torch.Tensor]:
    pairs = truncate_and_pair(query_text_sequences, context_text_sequences, tokenizer, max_query_length, max_context_length,
                              docs_per_query)
    kw = dict(add_special_tokens=True, return_tensors="pt", padding="max_length", truncation=True,
              max_length=max_decoder_source_length, return_attention_mask=True, return_token_type_ids=True)
    if hasattr(tokenizer, "batch_encode_plus"):          # transformers 4.x (the reference pins 4.38.2)
        enc = tokenizer.batch_encode_plus(pairs, **kw)
    else:                                                # transformers 5.x dropped it; __call__ is equivalent
        enc = tokenizer([q for q, _ in pairs], [c for _, c in pairs], **kw)
    return {"input_ids": enc["input_ids"].to(torch.int64), "attention_mask": enc["attention_mask"].to(torch.int64),
            "token_type_ids": enc["token_type_ids"].to(torch.int64)}


class NativePairTokenizer:
    """The same assembly in native code (csrc/pair_tokenizer.cpp behind `rr_tok_*`): multi-threaded WordPiece +
    truncate-by-round-trip + pair encoding straight into pinned int64 host tensors, no Python per text.

    `vocab` is the id-ordered token list (the lines of vocab.txt) or an HF tokenizer exposing `.vocab` / `get_vocab()`.
    Semantics: transformers 4.38.2 slow `BertTokenizer` (what the reference pins), see the C++ file's header."""

    def __init__(self, vocab, do_lower_case: bool = True, n_threads: int = 0):
        import ctypes as C
        from . import _lib as L
        self._C, self._L = C, L
        self.lib = L.load()
        if not isinstance(vocab, (list, tuple)):
            v = vocab.get_vocab() if hasattr(vocab, "get_vocab") else vocab.vocab
            vocab = [t for t, _ in sorted(v.items(), key=lambda kv: kv[1])]
        self.vocab = list(vocab)
        arr = (C.c_char_p * len(self.vocab))(*[t.encode("utf-8") for t in self.vocab])
        h = C.c_void_p()
        L.check(self.lib.rr_tok_create(arr, len(self.vocab), int(do_lower_case), C.byref(h)), None, "rr_tok_create")
        self.h, self.n_threads = h, int(n_threads)

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                self.lib.rr_tok_destroy(h)
            except Exception:
                pass
            self.h = None

    def encode(self, text: str, max_length: int = -1) -> List[int]:
        C = self._C
        raw = text.encode("utf-8", "replace")
        cap = len(raw) + 8                                    # every id consumes at least one input byte
        buf = (C.c_int32 * cap)()
        n = self.lib.rr_tok_encode(self.h, raw, int(max_length), buf, cap)
        if n < 0:
            raise ValueError(f"rr_tok_encode failed ({n})")
        return list(buf[:n])

    def decode(self, ids) -> str:
        C = self._C
        ids = list(ids)
        arr = (C.c_int32 * max(1, len(ids)))(*ids)
        cap = 16 + sum(len(self.vocab[i].encode("utf-8")) + 1 for i in ids)
        out = C.create_string_buffer(cap)
        n = self.lib.rr_tok_decode(self.h, arr, len(ids), out, cap)
        if n < 0:
            raise ValueError(f"rr_tok_decode failed ({n})")
        return out.raw[:n].decode("utf-8")

    def prepare_full_context_inputs(self, query_text_sequences: List[str], context_text_sequences: List[str],
                                    max_query_length: int, max_context_length: int, max_decoder_source_length: int,
                                    docs_per_query: int, pin_memory: bool = False) -> Dict[str, torch.Tensor]:
        C = self._C
        nq = len(query_text_sequences)
        N = nq * docs_per_query
        assert N == len(context_text_sequences), "expanded batch size must be batch_size * docs_per_query"   # :527
        q = (C.c_char_p * nq)(*[t.encode("utf-8", "replace") for t in query_text_sequences])
        c = (C.c_char_p * N)(*[t.encode("utf-8", "replace") for t in context_text_sequences])
        out = [torch.empty((N, max_decoder_source_length), dtype=torch.int64, pin_memory=pin_memory) for _ in range(3)]
        rc = self.lib.rr_tok_prepare_pairs(self.h, q, nq, c, docs_per_query, max_query_length, max_context_length,
                                           max_decoder_source_length, self.n_threads, out[0].data_ptr(), out[1].data_ptr(),
                                           out[2].data_ptr())
        if rc != 0:
            raise ValueError(f"rr_tok_prepare_pairs failed ({rc})")
        return {"input_ids": out[0], "attention_mask": out[1], "token_type_ids": out[2]}


def group_pairs_by_length(lengths, padded_len: int, granule: int, min_len: int = 1, segment_cost_rows: int = 0):
    """Host side of the packed forward (rr_forward_packed): pairs -> segments of equal row length.
    `lengths[i]` = token count of pair i (1 + index of its last non-pad position, as the tokenizer knows it), clipped to
    [1, padded_len]; a pair goes to the smallest multiple of `granule` (at least `min_len`: the mapping network's
    cross-attention window when image features are present; at most `padded_len`) that holds it.
    Returns (order, seg_pairs, seg_len): `order` lists the pair indices segment after segment (ascending length, input order
    kept inside a segment: a stable sort), seg_pairs / seg_len one entry per NON-EMPTY segment.  The reference pads every
    pair to padded_len (utils.py:157-165): one segment of that length.
    `segment_cost_rows` > 0: every segment costs the forward a fixed number of small launches (embeddings, masks, gathers, the
    CLS attention: ~140 us on an MI355X), so neighbouring lengths are MERGED where that is cheaper than the rows the merge pads:
    the segmentation of the sorted lengths that minimises (rows computed + segment_cost_rows x segments), by dynamic
    programming over the <= padded_len / granule distinct lengths.  Any grouping computes the same logits (a pair only has to
    fit its segment's length)."""
    import numpy as np
    if granule <= 0 or padded_len <= 0:
        raise ValueError("granule and padded_len must be positive")
    ln = np.clip(np.asarray(lengths, dtype=np.int64).reshape(-1), 1, padded_len)
    sizes = np.asarray(sorted({min(padded_len, max(int(min_len), g)) for g in range(granule, padded_len + granule, granule)}), dtype=np.int64)
    which = np.searchsorted(sizes, ln, side="left")               # smallest segment length >= len
    order = np.argsort(which, kind="stable")
    counts = np.bincount(which, minlength=len(sizes))
    keep = counts > 0
    cnt, szs = counts[keep].tolist(), sizes[keep].tolist()
    if segment_cost_rows > 0 and len(cnt) > 1:
        m = len(cnt)
        pre = [0]
        for c in cnt:
            pre.append(pre[-1] + c)
        best = [0.0] + [float("inf")] * m          # best[j] = cheapest segmentation of the first j lengths
        cut = [0] * (m + 1)
        for j in range(1, m + 1):
            for i in range(j):                        # last segment = lengths i .. j-1, run at length szs[j-1]
                c = best[i] + (pre[j] - pre[i]) * szs[j - 1] + segment_cost_rows
                if c < best[j]:
                    best[j], cut[j] = c, i
        segs, j = [], m
        while j > 0:
            segs.append((cut[j], j))
            j = cut[j]
        segs.reverse()
        cnt, szs = [pre[j] - pre[i] for i, j in segs], [szs[j - 1] for i, j in segs]
    return order, cnt, szs
