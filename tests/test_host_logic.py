"""CPU: host-side logic of the drop-in classes (no GPU compute): parameter contract,
error conventions, sparse text provider known answers."""
import asyncio

import numpy as np
import pytest

from oracle import oracle as O


def run(coro):
    return asyncio.run(coro)


# ---------------------------------------------------------------------------- search_params contract
def test_make_params_indexes_like_the_reference():
    from rag_application_amd import engine
    good = dict(matryoshka_64_limit=100, matryoshka_128_limit=80, matryoshka_256_limit=60, dense_limit=40,
                quantized_limit=40, sparse_limit=50, final_limit=30, hnsw_ef=128)
    p = engine.make_params(good)
    assert (p.matryoshka_64_limit, p.dense_limit, p.final_limit, p.rrf_limit, p.rrf_rank_base) == (100, 40, 30, 10, 0)
    assert p.rrf_k == 2.0 and p.mode == engine.HX_MODE_TREE
    with pytest.raises(TypeError):           # hybrid_search(search_params=None): qdrant_handler.py:314
        engine.make_params(None)
    bad = dict(good)
    del bad["sparse_limit"]
    with pytest.raises(KeyError):
        engine.make_params(bad)


# ---------------------------------------------------------------------------- handler conventions
def test_handler_error_conventions_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu tests")
    from rag_application_amd.handler import QdrantHandler
    h = QdrantHandler()
    with pytest.raises(ValueError):                                   # :39-40
        run(h.create_collection(""))
    with pytest.raises(Exception):                                    # engine needs a device: re-raised
        run(h.create_collection("u1"))
    assert run(h.hybrid_search("nobody", "q", [0.0] * 768, {"indices": [], "values": []},
                               search_params=None)) == []             # :384-386
    assert run(h.get_collection_chunk_count("nobody")) == 0            # :461-462
    assert run(h.get_all_containers()) == []
    with pytest.raises(Exception):                                    # :437-439
        run(h.delete_collection("nobody"))
    with pytest.raises(Exception):                                    # :196-198
        run(h.store_document_vectors([{"dense_embedding": [0.0] * 768}], "u2"))


def test_rerank_hook_and_sparse_input_forms():
    from rag_application_amd.handler import QdrantHandler, ScoredPoint, SparseVector, _sparse_parts

    class Rev:
        def rerank_documents(self, query, documents, max_tokens):
            return list(range(len(documents)))[::-1]

    class Boom:
        def rerank_documents(self, *a):
            raise RuntimeError("no model")

    res = [ScoredPoint(id=str(i), version=0, score=1.0 - i, payload={"content": f"d{i}"}) for i in range(3)]
    out = run(QdrantHandler(reranker=Rev()).rerank_with_colbert("q", ["d0", "d1", "d2"], res, 100))
    assert [r.id for r in out] == ["2", "1", "0"]
    assert run(QdrantHandler(reranker=Boom()).rerank_with_colbert("q", ["d0"], res, 100)) is res   # :410-412
    assert run(QdrantHandler(reranker=None).rerank_with_colbert("q", ["d0"], res, 100)) is res
    assert _sparse_parts({"indices": [1, 2], "values": [0.5, 1.0]}) == ([1, 2], [0.5, 1.0])
    assert _sparse_parts(SparseVector([3], [2.0])) == ([3], [2.0])
    assert res[0].dict()["payload"] == {"content": "d0"} and hasattr(res[0], "payload")


# ---------------------------------------------------------------------------- sparse text provider
def test_bm25_known_answers():
    from rag_application_amd import bm25
    assert bm25.murmur3_x86_32(b"") == 0
    assert bm25.murmur3_x86_32(b"hello") == 613153351
    assert bm25.term_id("foo") == 156908512
    for tok in ("", "a", "ab", "abc", "abcd", "abcde", "hello world", "naïve café"):
        assert bm25.murmur3_x86_32(tok.encode()) == O.murmur3_x86_32(tok.encode())
        assert bm25.term_id(tok) == O.bm25_term_id(tok)
    # Porter2 vocabulary samples (published snowball english test pairs)
    pairs = {"consign": "consign", "consigned": "consign", "consigning": "consign", "consistency": "consist",
             "generously": "generous", "running": "run", "happily": "happili", "national": "nation",
             "caresses": "caress", "ponies": "poni", "ties": "tie", "cries": "cri", "agreed": "agre",
             "hopping": "hop", "hoping": "hope", "relational": "relat", "conditional": "condit",
             "electricity": "electr", "argument": "argument", "generalization": "general", "skies": "sky",
             "dying": "die", "news": "news", "gas": "gas", "gaps": "gap", "feed": "feed", "by": "by",
             "say": "say", "cry": "cri", "knightly": "knight", "controlling": "control", "rolling": "roll",
             "sensational": "sensat", "youthful": "youth", "yes": "yes", "embedding": "embed",
             "vectors": "vector", "retrieval": "retriev", "searching": "search", "quantized": "quantiz"}
    for w, s in pairs.items():
        assert bm25.stem(w) == s, (w, bm25.stem(w), s)
    # weights: tf=1,len=256 -> 1.0 ; tf=2,len=256 -> 1.375 (SURVEY.md §8c ii)
    idx, val = bm25.embed("vector")
    assert idx == [bm25.term_id("vector")]
    assert val[0] == pytest.approx(1 * 2.2 / (1 + 1.2 * (0.25 + 0.75 * 1 / 256)))
    idx, val = bm25.embed("The searching of vectors and the search of a vector!")
    toks = bm25.stemmed_tokens("The searching of vectors and the search of a vector!")
    assert toks == ["search", "vector", "search", "vector"]
    assert sorted(idx) == idx and len(idx) == 2
    w = 2 * 2.2 / (2 + 1.2 * (0.25 + 0.75 * 4 / 256))
    assert val == pytest.approx([w, w])
    assert bm25.embed("the of and") == ([], [])
    np.testing.assert_allclose(O.bm25_weight([2], [4]), [w], rtol=1e-7)


def test_embedding_handler_contract():
    from rag_application_amd.embedding import EmbeddingHandler, ModelType, Provider

    class Fake:
        async def embed_text(self, texts):
            return [[float(len(t))] * 4 for t in texts]

    class Empty:
        async def embed_text(self, texts):
            return []

    h = EmbeddingHandler(provider=Provider.HUGGINGFACE, model_name="m", model_type=ModelType.TEXT_EMBEDDING,
                         model=Fake())
    assert run(h.encode_dense("abc")) == [[3.0] * 4]
    assert run(h.encode_dense(["ab", "abcd"])) == [[2.0] * 4, [4.0] * 4]
    assert run(h.encode_dense("abc")) == [[3.0] * 4]                      # cache hit
    assert run(EmbeddingHandler(model=Empty()).encode_dense("x")) == []    # :96-98
    assert run(EmbeddingHandler(model_name="/nonexistent").encode_dense("x")) == []
    sv = run(h.encode_sparse("searching vectors"))
    assert sv.indices and len(sv.indices) == len(sv.values)
    assert run(h.encode_sparse("searching vectors")).indices == sv.indices
    assert Provider.HUGGINGFACE.value == "huggingface" and ModelType.RERANKER.value == "reranker"
    key = h._get_cache_key("abc", "dense")
    assert key.startswith("embedding:dense:Provider.HUGGINGFACE:m:") and len(key.split(":")[-1]) == 64


def test_bm25_native_batch_equals_python():
    """csrc/bm25.cpp (all host cores) against bm25.embed on every text: ASCII, typographic
    punctuation (handled natively), other Unicode (flagged, Python path), suffix stress, empties."""
    import random
    from rag_application_amd import bm25
    rnd = random.Random(11)
    vocab = ("retrieval engines running quickly generously communal arsenal skies dying relational conditional "
             "rationalization hopefulness sensitivities formalize electricity adjustable replacement adoption "
             "controlled rolling agreed proceeding succeed news bias sky only early cries cried caresses ponies ties "
             "gaps gas this kiwis yes toy boy happy happily sadly feudally luxuriated plotted hopping hoping").split()
    sufs = ["ization", "ational", "fulness", "ousness", "tional", "biliti", "lessli", "entli", "ation", "alism", "aliti",
            "ousli", "iviti", "fulli", "enci", "anci", "abli", "izer", "ator", "alli", "bli", "ogi", "li", "ative", "ical",
            "ness", "ful", "ement", "ance", "able", "ment", "ent", "ism", "ate", "iti", "ous", "ive", "ize", "ion", "al", "er",
            "ic", "ing", "ed", "edly", "ingly", "eed", "eedly", "s", "es", "ies", "ied", "sses", "y", "e", "l", "ll"]
    texts = [" ".join(rnd.choice(vocab) for _ in range(rnd.randint(0, 80))) for _ in range(300)]
    texts += [t.upper() for t in texts[:20]]
    texts += [" ".join(s + u for s in ("hop", "relat", "sens", "geolog", "commun", "gener", "fizz", "tray", "cry", "x")
                       for u in sufs)]
    texts += ["", "the of and", "_ __ a_b x" * 3, "it's the runners' run; e-mail: a@b.c (ok)", "w" * 41 + " fine",
              "x—y × z § 8 “quoted” …", "naïve café", "½ cup ²", "тест text",
              "tabs\tand\nnewlines\r\nmix"]
    ref = [bm25.embed(t) for t in texts]
    assert bm25.embed_batch(texts) == ref
    assert bm25.embed_batch(texts, native=False) == ref
    ip, ix, v = bm25.embed_batch_csr(texts)
    assert ip[-1] == sum(len(r[0]) for r in ref) and ix.dtype == np.int32 and v.dtype == np.float64
    for i, (ri, rv) in enumerate(ref):
        assert ix[ip[i]:ip[i + 1]].tolist() == ri and v[ip[i]:ip[i + 1]].tolist() == rv


def test_encode_sparse_batch_matches_single():
    import asyncio
    from rag_application_amd.embedding import EmbeddingHandler
    h = EmbeddingHandler()
    texts = ["hybrid dense sparse retrieval", "", "Kernels and wavefronts — quickly!", "hybrid dense sparse retrieval"]
    one = [asyncio.run(h.encode_sparse(t)) for t in texts]
    h2 = EmbeddingHandler()
    many = asyncio.run(h2.encode_sparse_batch(texts))
    assert [(list(a.indices), list(a.values)) for a in one] == [(list(b.indices), list(b.values)) for b in many]


def test_payload_filter_semantics():
    """rag_application_amd/filters.py: the Qdrant Filter model restated (qdrant_handler.py:297, 466)."""
    from rag_application_amd.filters import matches
    p = {"file_name": "a.txt", "page_number": 3, "tags": ["x", "y"], "content": "Hybrid dense retrieval",
         "meta": {"lang": "en", "score": 0.5}, "none": None, "empty": []}
    assert matches(p, None) and matches(p, {})
    assert matches(p, {"must": [{"key": "file_name", "match": {"value": "a.txt"}}]})
    assert not matches(p, {"must": [{"key": "file_name", "match": {"value": "b.txt"}}]})
    assert matches(p, {"must": [{"key": "tags", "match": {"value": "y"}}]})                 # list payload: any element
    assert matches(p, {"must": [{"key": "tags", "match": {"any": ["q", "x"]}}]})
    assert not matches(p, {"must": [{"key": "tags", "match": {"except": ["x"]}}]})
    assert matches(p, {"must": [{"key": "file_name", "match": {"except": ["z"]}}]})
    assert matches(p, {"must": [{"key": "content", "match": {"text": "dense hybrid"}}]})
    assert matches(p, {"must": [{"key": "page_number", "range": {"gte": 3, "lt": 4}}]})
    assert not matches(p, {"must": [{"key": "page_number", "range": {"gt": 3}}]})
    assert matches(p, {"must": [{"key": "meta.lang", "match": {"value": "en"}}, {"key": "meta.score", "range": {"lte": 0.5}}]})
    assert not matches(p, {"must": [{"key": "missing", "match": {"value": 1}}]})
    assert matches(p, {"must_not": [{"key": "missing", "match": {"value": 1}}]})
    assert matches(p, {"should": [{"key": "file_name", "match": {"value": "zzz"}}, {"key": "page_number", "match": {"value": 3}}]})
    assert not matches(p, {"should": [{"key": "file_name", "match": {"value": "zzz"}}]})
    assert matches(p, {"must": [{"is_empty": {"key": "empty"}}, {"is_empty": {"key": "missing"}}, {"is_null": {"key": "none"}}]})
    assert not matches(p, {"must": [{"is_null": {"key": "missing"}}]})
    assert matches(p, {"must": [{"has_id": ["id-1", "id-2"]}]}, point_id="id-2") and not matches(p, {"must": [{"has_id": ["q"]}]}, "id-2")
    assert matches(p, {"must": [{"should": [{"key": "page_number", "match": {"value": 9}}, {"key": "tags", "match": {"value": "x"}}]}]})
    with pytest.raises(ValueError):
        matches(p, {"min_should": {}})
    with pytest.raises(ValueError):
        matches(p, {"must": [{"key": "a", "geo_radius": {}}]})


def test_local_hf_encoder_unmasked_mean_and_rerank(tmp_path):
    """LocalHFEncoder on a tiny randomly initialised BERT saved locally (no network): embed_text is the
    UNMASKED mean of last_hidden_state over the padded batch (huggingface.py:165-170), the "reranker"
    is mean-pooled query . docs with argsort descending (:172-189)."""
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    from rag_application_amd.embedding import EmbeddingHandler, LocalHFEncoder
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + "hybrid dense sparse retrieval engine kernel wave front the a of query document".split()
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    tok = tr.BertTokenizer(str(tmp_path / "vocab.txt"))
    tok.save_pretrained(str(tmp_path))
    torch.manual_seed(0)
    cfg = tr.BertConfig(vocab_size=len(vocab), hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                        intermediate_size=64, max_position_embeddings=64)
    tr.BertModel(cfg).save_pretrained(str(tmp_path))
    enc = LocalHFEncoder(str(tmp_path), device="cpu")
    texts = ["hybrid dense sparse retrieval", "kernel", "the query of a document engine wave front"]
    got = np.asarray(asyncio.run(enc.embed_text(texts)), np.float32)
    inputs = enc.tokenizer(texts, return_tensors="pt", padding=True)
    with torch.no_grad():
        ref = enc.model(**inputs).last_hidden_state.mean(dim=1).numpy()      # padding positions included
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    assert got.shape == (3, 32)
    order = enc.rerank_documents("dense retrieval", texts, max_tokens=100)
    q = np.asarray(asyncio.run(enc.embed_text(["dense retrieval"])), np.float32)[0]
    assert order == np.argsort(got @ q)[::-1].tolist()
    # the reference's truncation branch tests len() of the tokenizer's BatchEncoding = its number of keys, so a
    # document far longer than max_tokens characters is NOT cut (huggingface.py:177-182 as written): the order is
    # that of the whole documents
    long_docs = ["dense " * 20 + "kernel", "sparse " * 20 + "wave", "query document"]
    assert len(enc.tokenizer(long_docs, return_tensors="pt", padding=True)) < 8000
    whole = np.asarray(asyncio.run(enc.embed_text(long_docs)), np.float32)
    assert enc.rerank_documents("dense retrieval", long_docs, max_tokens=12) == np.argsort(whole @ q)[::-1].tolist()
    cut = np.asarray(asyncio.run(enc.embed_text([d[:12 - 5] + "....." for d in long_docs])), np.float32)
    assert not np.allclose(cut @ q, whole @ q)          # (the cut documents would have scored differently)
    # no truncation anywhere (:167): a text beyond the model's 64 positions fails inside the model -- the rerank
    # hook then keeps the order (qdrant_handler.py:410-412), encode_dense returns [] (embedding_handler.py:96-98)
    too_long = "kernel " * 80
    with pytest.raises(Exception):
        enc.rerank_documents("dense", [too_long], 10)
    with pytest.raises(Exception):
        enc.rerank_documents("x", [], 10)                 # as upstream: the tokenizer refuses an empty batch
    from rag_application_amd.handler import QdrantHandler
    res = ["r0"]
    assert asyncio.run(QdrantHandler(reranker=enc).rerank_with_colbert("dense", [too_long], res, 10)) is res
    assert asyncio.run(EmbeddingHandler(model_name=str(tmp_path)).encode_dense(too_long)) == []
    # through the EmbeddingHandler drop-in: model_name = local path
    h = EmbeddingHandler(model_name=str(tmp_path))
    out = asyncio.run(h.encode_dense(texts[0]))
    assert len(out) == 1 and len(out[0]) == 32
