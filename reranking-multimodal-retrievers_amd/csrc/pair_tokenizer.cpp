// Host-side pair-input assembly: WordPiece tokenisation, truncate-by-round-trip, pair encoding and
// padding of (query, candidate) texts into the int64 [N, S] tensors rr_forward takes, multi-threaded, written straight
// into caller-supplied (pinned) host buffers.  No GPU, no torch, no Python.
//
// Reference anchors (/root/reference/): `prepare_full_context_inputs` src/models/rerank/utils.py:129-167 calls
// `tokenizer.encode(text, add_special_tokens=False, max_length=n, truncation=True)`, `tokenizer.decode(ids)` and
// `tokenizer.batch_encode_plus(pairs, add_special_tokens=True, padding="max_length", truncation=True, max_length=L)`
// on the FLMR query tokenizer = BertTokenizer (src/models/flmr/models/flmr/tokenization_flmr.py:148-250 overrides
// only __call__).  The algorithm is transformers 4.38.2's slow BertTokenizer (un-vendored dependency, reference
// README.md:90-91): PreTrainedTokenizer.tokenize (per-character lower-casing outside special tokens, split on special
// tokens), BasicTokenizer (_clean_text, CJK spacing, whitespace split, lower + NFD + drop Mn, split on punctuation),
// WordpieceTokenizer (greedy longest match, 100-character limit), _decode + clean_up_tokenization, prepare_for_model
// with LONGEST_FIRST truncation and right padding.  tests/test_pair_tokenizer_cpu.py requires ids identical to a Python
// restatement of the same classes.  Unicode properties come from csrc/unicode_tables.h (generated from Python's
// unicodedata).  The NFC pass BasicTokenizer runs before its whitespace split is not needed in the uncased mode (NFD and
// the mark strip follow per token, NFD(NFC(x)) = NFD(x)); cased vocabularies, where it would matter, are refused.
#include "../../include/rerank_mi355.h"
#include "unicode_tables.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

using rr_unicode::kLower;
using rr_unicode::kNfd;
using rr_unicode::kRanges;

uint8_t cp_flags(uint32_t cp) {
  int lo = 0, hi = rr_unicode::kNumRanges - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < kRanges[mid].lo) hi = mid - 1;
    else if (cp > kRanges[mid].hi) lo = mid + 1;
    else return kRanges[mid].flags;
  }
  return 0;
}
bool is_cjk(uint32_t cp) {
  return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
         (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
         (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}
template <class T>
const T* find_map(const T* tab, int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < tab[mid].cp) hi = mid - 1;
    else if (cp > tab[mid].cp) lo = mid + 1;
    else return &tab[mid];
  }
  return nullptr;
}

// UTF-8 -> code points.  Malformed bytes become U+FFFD (which _clean_text then drops), as Python's decoder with
// errors="replace" would hand them to the tokenizer.
void utf8_decode(const char* s, size_t n, std::vector<uint32_t>& out) {
  out.clear();
  size_t i = 0;
  while (i < n) {
    const unsigned char c = (unsigned char)s[i];
    uint32_t cp = 0xFFFD;
    int len = 1;
    if (c < 0x80) cp = c;
    else if ((c >> 5) == 6 && i + 1 < n && ((unsigned char)s[i + 1] >> 6) == 2) {
      cp = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu);
      len = 2;
      if (cp < 0x80) cp = 0xFFFD;
    } else if ((c >> 4) == 14 && i + 2 < n && ((unsigned char)s[i + 1] >> 6) == 2 && ((unsigned char)s[i + 2] >> 6) == 2) {
      cp = ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
      len = 3;
      if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) cp = 0xFFFD;
    } else if ((c >> 3) == 30 && i + 3 < n && ((unsigned char)s[i + 1] >> 6) == 2 && ((unsigned char)s[i + 2] >> 6) == 2 &&
               ((unsigned char)s[i + 3] >> 6) == 2) {
      cp = ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) |
           ((unsigned char)s[i + 3] & 0x3Fu);
      len = 4;
      if (cp < 0x10000 || cp > 0x10FFFF) cp = 0xFFFD;
    }
    out.push_back(cp);
    i += len;
  }
}
void utf8_append(std::string& s, uint32_t cp) {
  if (cp < 0x80) s.push_back((char)cp);
  else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

void append_lower(std::vector<uint32_t>& out, uint32_t cp) {          // str.lower() of ONE character
  if (cp < 0x80) { out.push_back((cp >= 'A' && cp <= 'Z') ? cp + 32 : cp); return; }
  if (cp == 0x03A3) { out.push_back(0x03C3); return; }                // no context for a lone capital sigma
  if (const auto* m = find_map(kLower, rr_unicode::kNumLower, cp)) {
    for (int i = 0; i < rr_unicode::kLowerLen && m->to[i]; ++i) out.push_back(m->to[i]);
  } else out.push_back(cp);
}
void append_nfd_no_marks(std::vector<uint32_t>& out, uint32_t cp) {  // NFD(cp) without its Mn characters
  if (cp < 0xC0) { out.push_back(cp); return; }
  if (cp >= 0xAC00 && cp <= 0xD7A3) {                                 // Hangul syllable -> jamo (none is Mn)
    const uint32_t s = cp - 0xAC00, l = 0x1100 + s / 588, v = 0x1161 + (s % 588) / 28, t = 0x11A7 + s % 28;
    out.push_back(l); out.push_back(v);
    if (t != 0x11A7) out.push_back(t);
    return;
  }
  if (const auto* m = find_map(kNfd, rr_unicode::kNumNfd, cp)) {
    for (int i = 0; i < rr_unicode::kNfdLen && m->to[i]; ++i)
      if (!(cp_flags(m->to[i]) & rr_unicode::MN)) out.push_back(m->to[i]);
  } else if (!(cp_flags(cp) & rr_unicode::MN)) out.push_back(cp);
}

// Persistent worker pool: thread creation is expensive next to a 100-microsecond work item (4 ms per std::thread in the
// build VM), so workers are created once per handle and parked on a condition variable between calls.
class WorkerPool {
 public:
  ~WorkerPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++gen_; }
    cv_job_.notify_all();
    for (auto& t : th_) t.join();
  }
  // runs fn on `n` threads (the caller is one of them); calls are serialised per pool
  void run(int n, const std::function<void()>& fn) {
    std::lock_guard<std::mutex> call(call_);
    if (n <= 1) { fn(); return; }
    {
      std::unique_lock<std::mutex> g(m_);
      while ((int)th_.size() < n - 1) { const int id = (int)th_.size(); th_.emplace_back([this, id] { loop(id); }); }
      job_ = &fn; want_ = n - 1; done_ = 0; ++gen_;
    }
    cv_job_.notify_all();
    // An exception in any participant (the caller's share included) is kept, every worker is still waited for — they
    // reference the caller's stack through `fn` — and the first exception is rethrown to the caller afterwards, where the
    // extern "C" entry point turns it into a status code.  Nothing ever unwinds out of a worker thread (std::terminate).
    try { fn(); } catch (...) { std::lock_guard<std::mutex> g(m_); if (!err_) err_ = std::current_exception(); }
    std::exception_ptr err;
    {
      std::unique_lock<std::mutex> g(m_);
      cv_done_.wait(g, [&] { return done_ == want_; });
      job_ = nullptr;
      err = err_;
      err_ = nullptr;
    }
    if (err) std::rethrow_exception(err);
  }

 private:
  void loop(int id) {
    unsigned long long seen = 0;
    {
      std::lock_guard<std::mutex> g(m_);
      seen = gen_ - ((job_ && id < want_) ? 1 : 0);   // a worker created for the current job takes part in it
    }
    for (;;) {
      const std::function<void()>* job = nullptr;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_job_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (id < want_) job = job_;
      }
      if (job) {
        std::exception_ptr e;
        try { (*job)(); } catch (...) { e = std::current_exception(); }
        std::lock_guard<std::mutex> g(m_);
        if (e && !err_) err_ = e;
        ++done_;
        cv_done_.notify_all();
      }
    }
  }
  std::mutex call_, m_;
  std::condition_variable cv_job_, cv_done_;
  std::vector<std::thread> th_;
  const std::function<void()>* job_ = nullptr;
  unsigned long long gen_ = 0;
  int want_ = 0, done_ = 0;
  bool stop_ = false;
  std::exception_ptr err_;
};

}  // namespace

struct rr_tokenizer {
  mutable WorkerPool pool;
  std::vector<std::string> inv;                       // id -> token
  std::unordered_map<std::string, int32_t> vocab;     // whole-word pieces
  std::unordered_map<std::string, int32_t> cont;      // "##" pieces, keyed WITHOUT the prefix
  std::vector<std::pair<std::string, int32_t>> special;
  bool lower = true;
  int32_t unk = -1, cls = -1, sep = -1, pad = -1;
  size_t max_piece_bytes = 0;
  std::string err;

  // one basic token (code points, already lower-cased / accent-stripped / punctuation-split) -> ids
  void wordpiece(const std::vector<uint32_t>& cps, size_t b, size_t e, std::vector<int32_t>& ids, std::string& scratch,
                 std::vector<uint32_t>& off) const {
    const size_t n = e - b;
    if (n > 100) { ids.push_back(unk); return; }
    scratch.clear();
    off.clear();
    for (size_t i = b; i < e; ++i) { off.push_back((uint32_t)scratch.size()); utf8_append(scratch, cps[i]); }
    off.push_back((uint32_t)scratch.size());
    const size_t first = ids.size();
    size_t start = 0;
    while (start < n) {
      size_t end = n;
      int32_t hit = -1;
      while (start < end) {
        const size_t len = off[end] - off[start];
        if (len <= max_piece_bytes) {
          const auto& tab = start ? cont : vocab;
          auto it = tab.find(scratch.substr(off[start], len));
          if (it != tab.end()) { hit = it->second; break; }
        }
        --end;
      }
      if (hit < 0) { ids.resize(first); ids.push_back(unk); return; }
      ids.push_back(hit);
      start = end;
    }
  }

  // BasicTokenizer + WordPiece over one special-token-free segment
  void basic(const std::vector<uint32_t>& seg, std::vector<int32_t>& ids, std::vector<uint32_t>& a, std::vector<uint32_t>& bb,
             std::string& scratch, std::vector<uint32_t>& off) const {
    // per-character lower (PreTrainedTokenizer.tokenize), _clean_text, CJK spacing
    a.clear();
    for (uint32_t cp0 : seg) {
      bb.clear();
      if (lower) append_lower(bb, cp0); else bb.push_back(cp0);
      for (uint32_t cp : bb) {
        const uint8_t f = cp_flags(cp);
        if (cp == 0 || cp == 0xFFFD || (f & rr_unicode::CTRL)) continue;
        if (f & rr_unicode::WS) { a.push_back(' '); continue; }
        if (is_cjk(cp)) { a.push_back(' '); a.push_back(cp); a.push_back(' '); }
        else a.push_back(cp);
      }
    }
    // whitespace tokens -> lower (idempotent here) + NFD without Mn -> split on punctuation -> wordpiece
    size_t i = 0;
    const size_t n = a.size();
    while (i < n) {
      while (i < n && a[i] == ' ') ++i;
      size_t j = i;
      while (j < n && a[j] != ' ') ++j;
      if (j > i) {
        bb.clear();
        if (lower) for (size_t k = i; k < j; ++k) append_nfd_no_marks(bb, a[k]);
        else bb.assign(a.begin() + i, a.begin() + j);
        size_t s = 0;
        for (size_t k = 0; k <= bb.size(); ++k) {
          const bool p = k < bb.size() && (cp_flags(bb[k]) & rr_unicode::PUNCT);
          if (k == bb.size() || p) {
            if (k > s) wordpiece(bb, s, k, ids, scratch, off);
            if (p) wordpiece(bb, k, k + 1, ids, scratch, off);
            s = k + 1;
          }
        }
      }
      i = j;
    }
  }

  void encode(const char* text, size_t len, std::vector<int32_t>& ids) const {
    std::vector<uint32_t> cps, seg, a, bb, off;
    std::string scratch;
    utf8_decode(text, len, cps);
    ids.clear();
    // split on the special tokens (ASCII, case-sensitive) in code-point space
    size_t i = 0;
    seg.clear();
    while (i < cps.size()) {
      int32_t hit = -1;
      size_t hl = 0;
      if (cps[i] == '[') {
        for (const auto& sp : special) {
          const std::string& t = sp.first;
          if (i + t.size() <= cps.size()) {
            bool ok = true;
            for (size_t k = 0; k < t.size() && ok; ++k) ok = cps[i + k] == (unsigned char)t[k];
            if (ok) { hit = sp.second; hl = t.size(); break; }
          }
        }
      }
      if (hit >= 0) {
        if (!seg.empty()) { basic(seg, ids, a, bb, scratch, off); seg.clear(); }
        ids.push_back(hit);
        i += hl;
      } else seg.push_back(cps[i++]);
    }
    if (!seg.empty()) basic(seg, ids, a, bb, scratch, off);
  }

  void decode(const int32_t* ids, size_t n, std::string& out) const {
    out.clear();
    for (size_t i = 0; i < n; ++i) {
      const std::string& t = inv[(size_t)ids[i]];
      if (i && t.size() >= 2 && t[0] == '#' && t[1] == '#') out.append(t, 2, std::string::npos);   // " ##" -> ""
      else { if (i) out.push_back(' '); out += t; }
    }
    // .strip(): Python strips Unicode whitespace; tokens hold none except through the vocab, ASCII is what can occur
    size_t b = 0, e = out.size();
    while (b < e && (out[b] == ' ' || out[b] == '\t' || out[b] == '\n' || out[b] == '\r')) ++b;
    while (e > b && (out[e - 1] == ' ' || out[e - 1] == '\t' || out[e - 1] == '\n' || out[e - 1] == '\r')) --e;
    out = out.substr(b, e - b);
    static const char* const rules[][2] = {{" .", "."}, {" ?", "?"}, {" !", "!"}, {" ,", ","}, {" ' ", "'"}, {" n't", "n't"},
                                           {" 'm", "'m"}, {" 's", "'s"}, {" 've", "'ve"}, {" 're", "'re"}};
    for (const auto& r : rules) {                                  // str.replace, one rule after the other
      const size_t fl = strlen(r[0]);
      std::string t;
      size_t pos = 0, f;
      while ((f = out.find(r[0], pos)) != std::string::npos) { t.append(out, pos, f - pos); t += r[1]; pos = f + fl; }
      if (pos) { t.append(out, pos, std::string::npos); out.swap(t); }
    }
  }
};

extern "C" {

int rr_tok_create(const char* const* vocab_tokens, int vocab_size, int do_lower_case, rr_tokenizer_handle* out) {
  try {
    if (!vocab_tokens || vocab_size <= 0 || !out) return RR_ERR_BAD_ARG;
    *out = nullptr;
    // cased vocabularies are refused: without lower-casing BasicTokenizer does not strip accents either, and its NFC pass
    // (not implemented here) would then decide the ids of decomposed input.  Uncased: NFD(NFC(x)) = NFD(x), the pass is moot
    // (tests/test_pair_tokenizer_cpu.py::test_decomposed_and_composed_input_tokenise_alike).  The reference is uncased.
    if (!do_lower_case) return RR_ERR_UNSUPPORTED;
    auto* t = new rr_tokenizer();
    t->lower = do_lower_case != 0;
    t->inv.reserve(vocab_size);
    for (int i = 0; i < vocab_size; ++i) {
      if (!vocab_tokens[i]) { delete t; return RR_ERR_BAD_ARG; }
      std::string s(vocab_tokens[i]);
      t->inv.push_back(s);
      if (s.size() > 2 && s[0] == '#' && s[1] == '#') {
        t->cont[s.substr(2)] = i;
        t->max_piece_bytes = std::max(t->max_piece_bytes, s.size() - 2);
      } else {
        t->max_piece_bytes = std::max(t->max_piece_bytes, s.size());
      }
      t->vocab[s] = i;                             // a repeated token keeps its LAST index, as load_vocab's dict does
    }
    auto id = [&](const char* s) { auto it = t->vocab.find(s); return it == t->vocab.end() ? -1 : it->second; };
    t->unk = id("[UNK]"); t->cls = id("[CLS]"); t->sep = id("[SEP]"); t->pad = id("[PAD]");
    if (t->unk < 0 || t->cls < 0 || t->sep < 0 || t->pad < 0) { delete t; return RR_ERR_MISSING_WEIGHT; }
    for (const char* s : {"[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]"})
      if (id(s) >= 0) t->special.emplace_back(s, id(s));
    *out = t;
    return RR_OK;
  
  } catch (const std::bad_alloc&) {
    return RR_ERR_OOM;
  } catch (...) {
    return RR_ERR_BAD_ARG;   // no C++ exception crosses the C ABI
  }
}

int rr_tok_destroy(rr_tokenizer_handle h) {
  if (!h) return RR_ERR_BAD_ARG;
  delete h;
  return RR_OK;
}

int rr_tok_encode(rr_tokenizer_handle h, const char* text, int max_tokens, int32_t* ids_out, int capacity) {
  try {
    if (!h || !text || !ids_out || capacity < 0) return RR_ERR_BAD_ARG;
    std::vector<int32_t> ids;
    h->encode(text, strlen(text), ids);
    if (max_tokens >= 0 && (int)ids.size() > max_tokens) ids.resize(max_tokens);
    if ((int)ids.size() > capacity) return RR_ERR_BAD_SHAPE;
    memcpy(ids_out, ids.data(), ids.size() * sizeof(int32_t));
    return (int)ids.size();
  
  } catch (const std::bad_alloc&) {
    return RR_ERR_OOM;
  } catch (...) {
    return RR_ERR_BAD_ARG;   // no C++ exception crosses the C ABI
  }
}

int rr_tok_decode(rr_tokenizer_handle h, const int32_t* ids, int n, char* out, int capacity) {
  try {
    if (!h || (!ids && n) || !out || n < 0 || capacity <= 0) return RR_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
      if (ids[i] < 0 || (size_t)ids[i] >= h->inv.size()) return RR_ERR_BAD_ARG;
    std::string s;
    h->decode(ids, (size_t)n, s);
    if ((int)s.size() + 1 > capacity) return RR_ERR_BAD_SHAPE;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
  
  } catch (const std::bad_alloc&) {
    return RR_ERR_OOM;
  } catch (...) {
    return RR_ERR_BAD_ARG;   // no C++ exception crosses the C ABI
  }
}

int rr_tok_prepare_pairs(rr_tokenizer_handle h, const char* const* queries, int n_queries, const char* const* contexts,
                         int docs_per_query, int max_query_length, int max_context_length, int max_length, int n_threads,
                         int64_t* input_ids, int64_t* attention_mask, int64_t* token_type_ids) {
  try {
    if (!h || !queries || !contexts || !input_ids || !attention_mask || !token_type_ids) return RR_ERR_BAD_ARG;
    if (n_queries <= 0 || docs_per_query <= 0 || max_query_length < 0 || max_context_length < 0 || max_length < 3)
      return RR_ERR_BAD_SHAPE;
    const size_t N = (size_t)n_queries * docs_per_query;
    for (int i = 0; i < n_queries; ++i) if (!queries[i]) return RR_ERR_BAD_ARG;
    for (size_t i = 0; i < N; ++i) if (!contexts[i]) return RR_ERR_BAD_ARG;
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = (int)std::min<size_t>((size_t)n_threads, N);
  
    // step 1: queries, truncated by the encode -> decode -> encode round trip (utils.py:131-136)
    std::vector<std::vector<int32_t>> qids((size_t)n_queries);
    {
      std::vector<int32_t> ids;
      std::string txt;
      for (int i = 0; i < n_queries; ++i) {
        h->encode(queries[i], strlen(queries[i]), ids);
        if ((int)ids.size() > max_query_length) ids.resize(max_query_length);
        h->decode(ids.data(), ids.size(), txt);
        h->encode(txt.data(), txt.size(), qids[(size_t)i]);
      }
    }
    // step 2: one work item per pair: context round trip (:139-144), pair encoding with LONGEST_FIRST truncation and
    // right padding (:157-165)
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      std::vector<int32_t> ids, cids;
      std::string txt;
      for (;;) {
        const size_t p = next.fetch_add(1);
        if (p >= N) break;
        h->encode(contexts[p], strlen(contexts[p]), ids);
        if ((int)ids.size() > max_context_length) ids.resize(max_context_length);
        h->decode(ids.data(), ids.size(), txt);
        h->encode(txt.data(), txt.size(), cids);
        const std::vector<int32_t>& q = qids[p / (size_t)docs_per_query];
        size_t la = q.size(), lb = cids.size();
        while (la + lb + 3 > (size_t)max_length) { if (la > lb) --la; else --lb; }
        int64_t* I = input_ids + p * (size_t)max_length;
        int64_t* A = attention_mask + p * (size_t)max_length;
        int64_t* T = token_type_ids + p * (size_t)max_length;
        size_t k = 0;
        I[k] = h->cls; T[k++] = 0;
        for (size_t i = 0; i < la; ++i) { I[k] = q[i]; T[k++] = 0; }
        I[k] = h->sep; T[k++] = 0;
        for (size_t i = 0; i < lb; ++i) { I[k] = cids[i]; T[k++] = 1; }
        I[k] = h->sep; T[k++] = 1;
        for (size_t i = 0; i < k; ++i) A[i] = 1;
        for (; k < (size_t)max_length; ++k) { I[k] = h->pad; A[k] = 0; T[k] = 0; }
      }
    };
    h->pool.run(n_threads, worker);
    return RR_OK;
  
  } catch (const std::bad_alloc&) {
    return RR_ERR_OOM;
  } catch (...) {
    return RR_ERR_BAD_ARG;   // no C++ exception crosses the C ABI
  }
}

}  // extern "C"
