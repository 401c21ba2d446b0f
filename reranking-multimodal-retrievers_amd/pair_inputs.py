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


def prepare_full_context_inputs(query_text_sequences: List[str], context_text_sequences: List[str], tokenizer,
                                max_query_length: int, max_context_length: int, max_decoder_source_length: int,
                                docs_per_query: int) -> Dict[str, torch.Tensor]:
    def clip(text: str, n: int) -> str:
        return tokenizer.decode(tokenizer.encode(text, add_special_tokens=False, max_length=n, truncation=True))

    queries = [clip(t, max_query_length) for t in query_text_sequences]
    contexts = [clip(t, max_context_length) for t in context_text_sequences]
    pairs = [(q, contexts[i * docs_per_query + j]) for i, q in enumerate(queries) for j in range(docs_per_query)]
    kw = dict(add_special_tokens=True, return_tensors="pt", padding="max_length", truncation=True,
              max_length=max_decoder_source_length, return_attention_mask=True, return_token_type_ids=True)
    if hasattr(tokenizer, "batch_encode_plus"):          # transformers 4.x (the reference pins 4.38.2)
        enc = tokenizer.batch_encode_plus(pairs, **kw)
    else:                                                # transformers 5.x dropped it; __call__ is equivalent
        enc = tokenizer([q for q, _ in pairs], [c for _, c in pairs], **kw)
    return {"input_ids": enc["input_ids"].to(torch.int64), "attention_mask": enc["attention_mask"].to(torch.int64),
            "token_type_ids": enc["token_type_ids"].to(torch.int64)}
