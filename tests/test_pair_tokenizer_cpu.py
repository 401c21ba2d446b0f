"""CPU: the native pair tokenizer (rr_tok_*, csrc/pair_tokenizer.cpp) against the Python restatement of the reference's
tokenizer calls (oracle/bert_tokenizer_oracle.py), and that restatement against the installed HF BertTokenizer."""
import os
import random

import pytest
import torch

from helpers import ROOT  # noqa: F401  (puts the repo root on sys.path)
from oracle import bert_tokenizer_oracle as TO

WORDS = ["what", "is", "the", "color", "of", "this", "bus", "red", "a", "big", "city", "street", "in", "london", "double",
         "decker", "buses", "are", "usually", "image", "query", "cafe", "naive", "uber", "tokyo", "it", "do", "not", "don",
         "he", "she", "they", "we", "have", "re", "ve", "m", "s", "t", "n", "can", "won", "i", "you", "σ", "ς",
         "οδοσ", "привет", "мир", "ᄒ", "ᅡ", "ᆫ",
         "日", "本", "語", "1", "2", "19", "2024", "hello", "fi", "istanbul"]
PIECES = ["##s", "##es", "##ing", "##ed", "##er", "##ly", "##a", "##b", "##c", "##e", "##o", "##n", "##t", "##1", "##9", "##us",
          "##ет", "##ᅡ", "##ᆫ", "##σ"]
PUNCT = list(".,?!'\"-:;()[]#&/") + ["—", "…", "«", "»", "。", "、", "？"]


def make_vocab():
    v = ["[PAD]"] + [f"[unused{i}]" for i in range(5)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    return v + WORDS + PIECES + PUNCT


# mixed scripts, accents, ligature (no canonical decomposition), Hangul (algorithmic NFD), dotted capital I (lower() yields
# two code points), soft hyphen / zero-width space (Cf: removed), tab / NBSP / ideographic space (whitespace), U+FFFD and NUL
# (removed), fullwidth and CJK punctuation, special tokens in the text, a word over the 100-character limit
RAW_WORDS = ["What", "is", "THE", "color", "of", "this", "bus", "buses", "Café", "naïve", "Über", "Tōkyō",
             "London's", "don't", "isn't", "do", "not", "it's", "they've", "we're", "I'm", "he", "said", ":", "«hello»",
             "日本語", "한", "привет", "Мир", "İstanbul",
             "ΟΔΟΣ", "Σ", "x" * 120, "red-bus", "(big)", "city.", "street,", "2024", "1999", "[MASK]",
             "[UNK]", "[SEP]", "a­b", "zero​width", "tab\there", "nb sp", "é", "ﬁ", "　", "�",
             "？", "…", "。", "#hash", "##s", "a##b", "  ", "\n", "nul\x00l", "\U0001f600", "\U00020000"]


def corpus(n, seed, avoid=()):
    rng = random.Random(seed)
    words = [w for w in RAW_WORDS if not any(a in w for a in avoid)]
    return [" ".join(rng.choice(words) for _ in range(rng.randint(0, 40))) for _ in range(n)]


@pytest.fixture(scope="module")
def native():
    from rmr_amd.pair_inputs import NativePairTokenizer
    return NativePairTokenizer(make_vocab(), do_lower_case=True, n_threads=4)


@pytest.fixture(scope="module")
def oracle():
    return TO.BertTokenizerOracle(make_vocab(), do_lower_case=True)


def _hf(tmp_path):
    from transformers import BertTokenizer
    f = os.path.join(tmp_path, "vocab.txt")
    with open(f, "w", encoding="utf-8") as fh:
        fh.write("\n".join(make_vocab()) + "\n")
    return BertTokenizer(f, do_lower_case=True)


def test_oracle_is_pinned_to_the_installed_hf_tokenizer(oracle, tmp_path):
    """The restatement of the 4.38.2 slow tokenizer equals the installed (Rust-backed) BertTokenizer wherever the two
    implementations are meant to agree: everything except the context-sensitive final sigma, the ' do not' decode rule
    of the Rust decoder, its per-token (instead of whole-string) apostrophe clean-up, and its balanced (instead of
    one-token-at-a-time) LONGEST_FIRST split, which can differ by one token when a pair has to be cut."""
    hf = _hf(tmp_path)
    texts = corpus(300, seed=1, avoid=("Σ", "do", "\x00"))
    for t in texts:
        a = hf.encode(t, add_special_tokens=False)
        b = oracle.encode(t)
        assert a == b, (t, hf.convert_ids_to_tokens(a), [oracle.inv[i] for i in b])
        if "'" not in t:
            assert hf.decode(a) == oracle.decode(b), t
    enc = hf(texts[:40], texts[40:80], add_special_tokens=True, padding="max_length", truncation=True, max_length=48,
             return_token_type_ids=True)
    fits = 0
    for i in range(40):
        ids, am, tt = oracle.encode_pair(texts[i], texts[40 + i], 48)
        if len(oracle.encode(texts[i])) + len(oracle.encode(texts[40 + i])) + 3 <= 48:
            fits += 1
            assert (ids, am, tt) == (enc["input_ids"][i], enc["attention_mask"][i], enc["token_type_ids"][i])
        else:                                                # cut pairs: same layout, same length budget
            assert sum(am) == 48 == sum(enc["attention_mask"][i]) and ids[0] == enc["input_ids"][i][0]
    assert fits >= 5


def test_native_encode_decode_equal_the_oracle(native, oracle):
    extra = ["", " ", "[MASK][UNK]", "[mask]", "a[SEP]b", "Do not say it's héllo, worlds!", "[", "[CLS", "]]"]
    for t in corpus(600, seed=2) + RAW_WORDS + extra:
        t = t.replace("\x00", "")                            # the C ABI takes NUL-terminated strings
        want = oracle.encode(t)
        got = native.encode(t)
        assert got == want, (t, [oracle.inv[i] for i in got], [oracle.inv[i] for i in want])
        assert native.encode(t, 7) == want[:7]
        assert native.decode(got) == oracle.decode(want), t


def test_native_pair_assembly_equals_the_oracle(native, oracle):
    nq, K = 5, 7
    q = [t.replace("\x00", "") for t in corpus(nq, seed=3)]
    c = [t.replace("\x00", "") for t in corpus(nq * K, seed=4)]
    for (mq, mc, L) in [(8, 20, 32), (32, 476, 512), (0, 3, 8), (4, 0, 16)]:
        want = TO.prepare_full_context_inputs(q, c, oracle, mq, mc, L, K)
        got = native.prepare_full_context_inputs(q, c, mq, mc, L, K)
        for k in ("input_ids", "attention_mask", "token_type_ids"):
            assert got[k].dtype == torch.int64 and tuple(got[k].shape) == (nq * K, L)
            assert got[k].tolist() == want[k], (k, mq, mc, L)


def test_threads_do_not_change_the_result():
    from rmr_amd.pair_inputs import NativePairTokenizer
    q = [t.replace("\x00", "") for t in corpus(3, seed=5)]
    c = [t.replace("\x00", "") for t in corpus(3 * 40, seed=6)]
    outs = [NativePairTokenizer(make_vocab(), n_threads=n).prepare_full_context_inputs(q, c, 32, 100, 128, 40) for n in (1, 2, 8)]
    for o in outs[1:]:
        for k in o:
            assert torch.equal(o[k], outs[0][k])


def test_native_equals_python_mirror_on_the_hf_tokenizer(native, tmp_path):
    """Drop-in check at the module boundary: the native assembly returns what pair_inputs.prepare_full_context_inputs
    returns when driven by an HF tokenizer object (texts inside the common ground of both implementations)."""
    from rmr_amd.pair_inputs import prepare_full_context_inputs
    hf = _hf(tmp_path)
    q = ["What is the color of this bus?", "Café in Tōkyō"]
    c = ["London buses are usually red.", "a big city street", "日本語 bus", "", "they have (big) buses", "x" * 150]
    a = prepare_full_context_inputs(q, c, hf, 6, 12, 24, 3)
    b = native.prepare_full_context_inputs(q, c, 6, 12, 24, 3)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_errors(native):
    from rmr_amd.pair_inputs import NativePairTokenizer
    with pytest.raises(Exception):
        NativePairTokenizer(["a", "b"])                      # no [UNK]/[CLS]/[SEP]/[PAD]
    with pytest.raises(AssertionError):
        native.prepare_full_context_inputs(["a"], ["b", "c"], 4, 4, 16, 3)
    with pytest.raises(ValueError):
        native.prepare_full_context_inputs(["a"], ["b"], 4, 4, 2, 1)      # max_length < 3
    with pytest.raises(NotImplementedError):
        NativePairTokenizer(make_vocab(), do_lower_case=False)            # cased: would need the NFC pass (not built)


def test_decomposed_and_composed_input_tokenise_alike(native, oracle):
    """HF 4.38 `BasicTokenizer.tokenize` runs an NFC pass before its whitespace split; the native tokenizer does not.  In
    the uncased mode the reference uses (bert-base-uncased, do_lower_case=True) every token is NFD-decomposed and stripped
    of its marks right after, and NFD(NFC(x)) = NFD(x) for every string (canonical equivalence), so the pass cannot change
    an id: checked here on the mixed-script corpus fed composed (NFC), decomposed (NFD) and as written.  The cased mode,
    where the pass would matter, is refused by rr_tok_create (test_errors)."""
    import unicodedata
    for t in corpus(300, seed=77, avoid=("\x00",)):           # (a C string ends at NUL)
        want = oracle.encode(unicodedata.normalize("NFC", t))
        for form in (t, unicodedata.normalize("NFD", t), unicodedata.normalize("NFC", t)):
            assert native.encode(form) == want, repr(form)
