"""CPU: pair-input assembly restated from utils.py:129-167, exercised with a locally generated WordPiece vocab
(no vocab file exists offline)."""
import os

import pytest
import torch


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    from transformers import BertTokenizer
    words = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    words += ["what", "is", "the", "color", "of", "this", "bus", "red", "a", "big", "city", "street", "in", "london",
              "double", "decker", ".", ",", "?", "##s", "##es", "buses", "are", "usually", "image", "query"]
    d = tmp_path_factory.mktemp("vocab")
    f = os.path.join(d, "vocab.txt")
    open(f, "w").write("\n".join(words) + "\n")
    return BertTokenizer(f, do_lower_case=True)


def test_pair_encoding_layout(tok):
    from rmr_amd.pair_inputs import prepare_full_context_inputs
    q = ["what is the color of this bus ?"]
    ctx = ["london buses are usually red .", "a big city street in london " * 10, ""]
    enc = prepare_full_context_inputs(q, ctx, tok, max_query_length=4, max_context_length=12,
                                      max_decoder_source_length=24, docs_per_query=3)
    ids, am, tt = enc["input_ids"], enc["attention_mask"], enc["token_type_ids"]
    assert ids.shape == am.shape == tt.shape == (3, 24) and ids.dtype == torch.int64
    cls, sep, pad = tok.cls_token_id, tok.sep_token_id, tok.pad_token_id
    assert pad == 0 and (ids[:, 0] == cls).all()
    for n in range(3):
        row = ids[n].tolist()
        first_sep = row.index(sep)
        assert first_sep == 1 + 4                                   # query truncated to 4 tokens
        assert tt[n, : first_sep + 1].sum() == 0                    # [CLS] q [SEP] -> type 0
        L = int(am[n].sum())
        assert row[L - 1] == sep and all(t == pad for t in row[L:])
        assert (tt[n, first_sep + 1: L] == 1).all() and tt[n, L:].sum() == 0
        assert (am[n] == (ids[n] != 0).long()).all()
    assert int(am[1].sum()) == 1 + 4 + 1 + 12 + 1                   # context truncated to 12 tokens
    assert int(am[2].sum()) == 1 + 4 + 1 + 0 + 1                    # empty context


def test_query_major_order(tok):
    from rmr_amd.pair_inputs import prepare_full_context_inputs
    enc = prepare_full_context_inputs(["red bus", "big city"], ["a", "the", "london", "street"], tok, 8, 8, 16, 2)
    ids = enc["input_ids"]
    red, big = tok.convert_tokens_to_ids("red"), tok.convert_tokens_to_ids("big")
    assert (ids[:2, 1] == red).all() and (ids[2:, 1] == big).all()


def test_group_pairs_by_length_segments():
    """Host side of rr_forward_packed (pair_inputs.group_pairs_by_length): smallest multiple of the granule that holds each
    pair, stable inside a segment, clipped to [min_len, padded length]; empty segments dropped."""
    import numpy as np
    from rmr_amd.pair_inputs import group_pairs_by_length
    lens = [1, 64, 65, 512, 600, 0, 130, 64, 33]
    order, n, L = group_pairs_by_length(lens, 512, 64)
    assert L == [64, 128, 192, 512] and n == [5, 1, 1, 2]
    assert order.tolist() == [0, 1, 5, 7, 8, 2, 6, 3, 4]                  # ascending segment, input order inside
    assert sum(n) == len(lens) and all(min(max(lens[i], 1), 512) <= L[s] for s, lo in enumerate(np.cumsum([0] + n[:-1])) for i in order[lo: lo + n[s]])
    # image features present: no segment below the mapping network's cross-attention window
    order, n, L = group_pairs_by_length([3, 20, 40], 64, 16, min_len=32)
    assert L == [32, 48] and n == [2, 1] and order.tolist() == [0, 1, 2]
    # one segment = the reference's padding; a granule beyond the padded length is that too
    assert group_pairs_by_length([5, 9], 128, 128)[1:] == ([2], [128])
    assert group_pairs_by_length([5, 9], 128, 1000)[1:] == ([2], [128])
    # a padded length that is not a multiple of the granule: the last segment is the padded length itself
    assert group_pairs_by_length([100], 100, 64)[1:] == ([1], [100])
    import pytest
    with pytest.raises(ValueError):
        group_pairs_by_length([1], 64, 0)


def test_segment_merging_is_valid_and_optimal():
    """group_pairs_by_length(segment_cost_rows > 0): every pair still fits its segment, the order is the unmerged one, and the
    segmentation minimises rows + cost x segments (checked against brute force over all cuts of a small case)."""
    import itertools
    import numpy as np
    from rmr_amd.pair_inputs import group_pairs_by_length
    rng = np.random.default_rng(3)
    ln = rng.integers(5, 130, 200)
    o0, n0, l0 = group_pairs_by_length(ln, 128, 16, 1, 0)
    for lam in (1, 50, 400, 5000, 10 ** 7):
        o, n, l = group_pairs_by_length(ln, 128, 16, 1, lam)
        assert (o == o0).all() and sum(n) == len(ln) and l == sorted(l) and l[-1] == l0[-1]
        pos = 0
        for cnt, L in zip(n, l):
            assert (ln[o[pos:pos + cnt]] <= L).all()
            pos += cnt
        cost = sum(a * b for a, b in zip(n, l)) + lam * len(n)
        best = min(sum(sum(n0[i:j]) * l0[j - 1] for i, j in zip((0,) + cuts, cuts + (len(n0),))) + lam * (len(cuts) + 1)
                   for r in range(len(n0)) for cuts in itertools.combinations(range(1, len(n0)), r))
        assert cost == best
    assert len(group_pairs_by_length(ln, 128, 16, 1, 10 ** 7)[1]) == 1           # everything in one padded segment
