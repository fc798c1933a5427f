"""In-repo sparse ("BM25") text provider: the data contract of
`EmbeddingHandler.encode_sparse` (app/core/embedding/embedding_handler.py:101-142),
which the reference delegates to fastembed's `Qdrant/bm25` model (:41, :123).

fastembed is absent from the build container and its files cannot be fetched, so this
restates its published pipeline from recollection -- PARITY UNPINNED, each step a
named switch:  lower-case -> replace non-word characters by spaces -> split ->
drop punctuation tokens, English stop-words and tokens longer than 40 characters ->
Snowball-English (Porter2) stem -> term id = abs(int32(murmur3_x86_32(token))) ->
value = tf*(k+1) / (tf + k*(1 - b + b*len/avg_len)), k=1.2, b=0.75, avg_len=256,
len = number of stemmed tokens.  The reference calls `.embed()` (document weighting)
for queries as well.  Known-answer tests: tests/test_host_logic.py."""
from __future__ import annotations

import re
import string
from collections import Counter
from typing import Dict, Iterable, List, Tuple

BM25_K = 1.2
BM25_B = 0.75
BM25_AVG_LEN = 256.0
TOKEN_MAX_LENGTH = 40

STOPWORDS = frozenset("""i me my myself we our ours ourselves you you're you've you'll you'd your yours yourself
yourselves he him his himself she she's her hers herself it it's its itself they them their theirs themselves what
which who whom this that that'll these those am is are was were be been being have has had having do does did doing
a an the and but if or because as until while of at by for with about against between into through during before
after above below to from up down in out on off over under again further then once here there when where why how
all any both each few more most other some such no nor not only own same so than too very s t can will just don
don't should should've now d ll m o re ve y ain aren aren't couldn couldn't didn didn't doesn doesn't hadn hadn't
hasn hasn't haven haven't isn isn't ma mightn mightn't mustn mustn't needn needn't shan shan't shouldn shouldn't
wasn wasn't weren weren't won won't wouldn wouldn't""".split())
PUNCTUATION = frozenset(string.punctuation)


# ---------------------------------------------------------------------------- murmur3
def murmur3_x86_32(data: bytes, seed: int = 0) -> int:
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed & 0xFFFFFFFF
    n = len(data)
    nb = n - (n & 3)
    for i in range(0, nb, 4):
        k = data[i] | (data[i + 1] << 8) | (data[i + 2] << 16) | (data[i + 3] << 24)
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
        h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
    k = 0
    rem = n & 3
    if rem == 3:
        k ^= data[nb + 2] << 16
    if rem >= 2:
        k ^= data[nb + 1] << 8
    if rem >= 1:
        k ^= data[nb]
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
    h ^= n
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def term_id(token: str) -> int:
    h = murmur3_x86_32(token.encode("utf-8"), 0)
    if h & 0x80000000:
        h -= 1 << 32
    return abs(h)


# ---------------------------------------------------------------------------- Porter2
_VOWELS = "aeiouy"
_DOUBLES = ("bb", "dd", "ff", "gg", "mm", "nn", "pp", "rr", "tt")
_LI_END = "cdeghkmnrt"
_EXC1 = {"skis": "ski", "skies": "sky", "dying": "die", "lying": "lie", "tying": "tie", "idly": "idl",
         "gently": "gentl", "ugly": "ugli", "early": "earli", "only": "onli", "singly": "singl", "sky": "sky",
         "news": "news", "howe": "howe", "atlas": "atlas", "cosmos": "cosmos", "bias": "bias", "andes": "andes"}
_EXC2 = {"inning", "outing", "canning", "herring", "earring", "proceed", "exceed", "succeed"}
_STEP2 = (("ization", "ize"), ("ational", "ate"), ("fulness", "ful"), ("ousness", "ous"), ("iveness", "ive"),
          ("tional", "tion"), ("biliti", "ble"), ("lessli", "less"), ("entli", "ent"), ("ation", "ate"),
          ("alism", "al"), ("aliti", "al"), ("ousli", "ous"), ("iviti", "ive"), ("fulli", "ful"),
          ("enci", "ence"), ("anci", "ance"), ("abli", "able"), ("izer", "ize"), ("ator", "ate"),
          ("alli", "al"), ("bli", "ble"), ("ogi", None), ("li", None))
_STEP3 = (("ational", "ate"), ("tional", "tion"), ("alize", "al"), ("icate", "ic"), ("iciti", "ic"),
          ("ative", None), ("ical", "ic"), ("ness", ""), ("ful", ""))
_STEP4 = ("ement", "ance", "ence", "able", "ible", "ment", "ant", "ent", "ism", "ate", "iti", "ous", "ive",
          "ize", "ion", "al", "er", "ic")


def _is_v(w, i):
    return w[i] in _VOWELS


def _regions(w):
    r1 = len(w)
    for pre in ("gener", "commun", "arsen"):
        if w.startswith(pre):
            r1 = len(pre)
            break
    else:
        for i in range(1, len(w)):
            if not _is_v(w, i) and _is_v(w, i - 1):
                r1 = i + 1
                break
    r2 = len(w)
    for i in range(r1 + 1, len(w)):
        if not _is_v(w, i) and _is_v(w, i - 1):
            r2 = i + 1
            break
    return r1, r2


def _short_syllable_end(w):
    n = len(w)
    if n == 2:
        return _is_v(w, 0) and not _is_v(w, 1)
    if n >= 3:
        return (not _is_v(w, n - 3)) and _is_v(w, n - 2) and (not _is_v(w, n - 1)) and w[n - 1] not in "wxY"
    return False


def _has_vowel(s):
    return any(c in _VOWELS for c in s)


def stem(word: str) -> str:
    """Snowball English (Porter2) stemmer, written from the published algorithm."""
    w = word
    if len(w) <= 2:
        return w
    if w in _EXC1:
        return _EXC1[w]
    if w[0] == "'":
        w = w[1:]
    if not w:
        return w
    chars = list(w)
    if chars[0] == "y":
        chars[0] = "Y"
    for i in range(1, len(chars)):
        if chars[i] == "y" and chars[i - 1] in _VOWELS:
            chars[i] = "Y"
    w = "".join(chars)
    r1, r2 = _regions(w)
    # step 0
    for suf in ("'s'", "'s", "'"):
        if w.endswith(suf):
            w = w[:-len(suf)]
            break
    # step 1a
    if w.endswith("sses"):
        w = w[:-2]
    elif w.endswith("ied") or w.endswith("ies"):
        w = w[:-2] if len(w) > 4 else w[:-1]
    elif w.endswith("us") or w.endswith("ss"):
        pass
    elif w.endswith("s"):
        if _has_vowel(w[:-2]):
            w = w[:-1]
    if w in _EXC2:
        return w.replace("Y", "y")
    # step 1b
    for suf in ("eedly", "eed"):
        if w.endswith(suf):
            if len(w) - len(suf) >= r1:
                w = w[:-len(suf)] + "ee"
            break
    else:
        for suf in ("ingly", "edly", "ing", "ed"):
            if w.endswith(suf):
                stem_ = w[:-len(suf)]
                if _has_vowel(stem_):
                    w = stem_
                    if w.endswith(("at", "bl", "iz")):
                        w += "e"
                    elif w.endswith(_DOUBLES):
                        w = w[:-1]
                    elif _short_syllable_end(w) and r1 >= len(w):
                        w += "e"
                break
    # step 1c
    if len(w) > 2 and w[-1] in "yY" and w[-2] not in _VOWELS:
        w = w[:-1] + "i"
    # step 2
    for suf, rep in _STEP2:
        if w.endswith(suf):
            if len(w) - len(suf) >= r1:
                if suf == "ogi":
                    if w.endswith("logi"):
                        w = w[:-1]
                elif suf == "li":
                    if len(w) >= 3 and w[-3] in _LI_END:
                        w = w[:-2]
                else:
                    w = w[:-len(suf)] + rep
            break
    # step 3
    for suf, rep in _STEP3:
        if w.endswith(suf):
            if len(w) - len(suf) >= r1:
                if suf == "ative":
                    if len(w) - len(suf) >= r2:
                        w = w[:-len(suf)]
                else:
                    w = w[:-len(suf)] + rep
            break
    # step 4
    for suf in _STEP4:
        if w.endswith(suf):
            if len(w) - len(suf) >= r2:
                if suf == "ion":
                    if len(w) > 3 and w[-4] in "st":
                        w = w[:-3]
                else:
                    w = w[:-len(suf)]
            break
    # step 5
    if w.endswith("e"):
        if len(w) - 1 >= r2 or (len(w) - 1 >= r1 and not _short_syllable_end(w[:-1])):
            w = w[:-1]
    elif w.endswith("l"):
        if len(w) - 1 >= r2 and len(w) > 1 and w[-2] == "l":
            w = w[:-1]
    return w.replace("Y", "y")


# ---------------------------------------------------------------------------- text -> sparse vector
_NONWORD = re.compile(r"[^\w]", flags=re.UNICODE)


def tokenize(text: str) -> List[str]:
    return _NONWORD.sub(" ", text.lower()).split()


def stemmed_tokens(text: str) -> List[str]:
    out = []
    for tok in tokenize(text):
        if tok in PUNCTUATION or tok in STOPWORDS or len(tok) > TOKEN_MAX_LENGTH:
            continue
        s = stem(tok)
        if s:
            out.append(s)
    return out


def embed(text: str, k: float = BM25_K, b: float = BM25_B, avg_len: float = BM25_AVG_LEN) -> Tuple[List[int], List[float]]:
    """One document (or query: the reference uses the same call) -> (indices, values)."""
    toks = stemmed_tokens(text)
    if not toks:
        return [], []
    tf = Counter(toks)
    doc_len = len(toks)
    acc: Dict[int, float] = {}
    for tok, n in tf.items():
        tid = term_id(tok)
        w = n * (k + 1.0) / (n + k * (1.0 - b + b * doc_len / avg_len))
        # two tokens may hash to one id: Qdrant rejects duplicate indices, keep the larger
        acc[tid] = max(acc.get(tid, 0.0), w)
    idx = sorted(acc)
    return idx, [acc[i] for i in idx]


def embed_batch_csr(texts: Iterable[str], native: bool = True, threads: int = 0):
    """Many texts -> one CSR (indptr int64[n+1], idx int32[nnz], val float64[nnz]): what
    hx_add_sparse takes.  Native (csrc/bm25.cpp, all host cores) for texts of ASCII + typographic
    punctuation -- identical to embed() -- and embed() for the rest."""
    import numpy as np
    texts = list(texts)
    n = len(texts)
    lib = None
    if native and n:
        try:
            import ctypes as C
            from . import _lib
            lib = _lib.lib()
        except Exception:
            lib = None
    if lib is None:
        rows = [embed(t) for t in texts]
        indptr = np.zeros(n + 1, np.int64)
        indptr[1:] = np.cumsum([len(r[0]) for r in rows])
        idx = np.fromiter((i for r in rows for i in r[0]), np.int32, int(indptr[-1]))
        val = np.fromiter((v for r in rows for v in r[1]), np.float64, int(indptr[-1]))
        return indptr, idx, val
    raw = [t.encode("utf-8") for t in texts]
    arr = (C.c_char_p * n)(*raw)
    lens = np.fromiter((len(r) for r in raw), np.int64, n)
    cap = int(lens.sum() // 2 + n)
    indptr = np.zeros(n + 1, np.int64)
    idx = np.empty(max(cap, 1), np.int32)
    val = np.empty(max(cap, 1), np.float64)
    flags = np.zeros(n, np.int32)
    rc = lib.hx_bm25_embed_batch(C.cast(arr, C.c_void_p), lens.ctypes.data, n, BM25_K, BM25_B, BM25_AVG_LEN, int(threads),
                                 indptr.ctypes.data, idx.ctypes.data, val.ctypes.data, cap, flags.ctypes.data)
    if rc != 0:
        return embed_batch_csr(texts, native=False)
    nnz = int(indptr[-1])
    idx, val = idx[:nnz], val[:nnz]
    slow = np.flatnonzero(flags)
    if slow.size:                      # splice the Python rows in
        rows = {int(i): embed(texts[int(i)]) for i in slow}
        counts = np.diff(indptr)
        for i, r in rows.items():
            counts[i] = len(r[0])
        new_ptr = np.zeros(n + 1, np.int64)
        new_ptr[1:] = np.cumsum(counts)
        nidx = np.empty(int(new_ptr[-1]), np.int32)
        nval = np.empty(int(new_ptr[-1]), np.float64)
        keep = np.ones(n, bool)
        keep[slow] = False
        # native rows keep their order: copy them in one pass, then fill the spliced rows
        src = np.concatenate([np.arange(indptr[i], indptr[i + 1]) for i in np.flatnonzero(keep)]) if keep.any() else np.zeros(0, np.int64)
        dst = np.concatenate([np.arange(new_ptr[i], new_ptr[i + 1]) for i in np.flatnonzero(keep)]) if keep.any() else np.zeros(0, np.int64)
        nidx[dst] = idx[src]
        nval[dst] = val[src]
        for i, r in rows.items():
            nidx[new_ptr[i]:new_ptr[i + 1]] = r[0]
            nval[new_ptr[i]:new_ptr[i + 1]] = r[1]
        indptr, idx, val = new_ptr, nidx, nval
    return indptr, idx, val


def embed_batch(texts: Iterable[str], native: bool = True, threads: int = 0):
    """Many texts at once -> [(indices, values)] exactly as [embed(t) for t in texts]."""
    texts = list(texts)
    indptr, idx, val = embed_batch_csr(texts, native=native, threads=threads)
    il, vl = idx.tolist(), val.tolist()
    return [(il[indptr[i]:indptr[i + 1]], vl[indptr[i]:indptr[i + 1]]) for i in range(len(texts))]
