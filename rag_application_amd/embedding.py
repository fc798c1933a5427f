"""Drop-in for the reference's embedding-provider interface
(app/core/embedding/embedding_handler.py:13-142): `EmbeddingHandler(provider,
model_name, model_type)` with `await encode_dense(str | list[str]) -> list[list[float]]`
([] on failure, :96-98) and `await encode_sparse(str) -> SparseVector-like`
({"indices": [], "values": []} on failure, :140-142).

Dense: the reference runs a HuggingFace encoder with UNMASKED mean pooling and no
normalisation (app/core/models/huggingface/huggingface.py:165-170); `LocalHFEncoder`
does the same forward on PyTorch-ROCm from a LOCAL checkpoint directory (there is no
network here: `from_pretrained(<hub name>)` cannot work, and no weights ship with the
reference, so encoder outputs are parity-unpinned).  Sparse: rag_application_amd.bm25.
The Redis cache of the reference (:52-69, 1 h TTL) is service glue; an in-process dict
with the same key format stands in."""
from __future__ import annotations

import asyncio
import hashlib
import logging
from enum import Enum
from typing import Any, Dict, List, Optional, Union

from . import bm25
from .handler import SparseVector


class Provider(Enum):          # app/core/models/model_provider.py:3-6
    OPENAI = "openai"
    HUGGINGFACE = "huggingface"
    OLLAMA = "ollama"


class ModelType(Enum):         # app/core/models/model_type.py:3-9
    TEXT_GENERATION = "text_generation"
    IMAGE_GENERATION = "image_generation"
    TEXT_EMBEDDING = "text_embedding"
    IMAGE_EMBEDDING = "image_embedding"
    RERANKER = "reranker"
    NER = "ner"


class LocalHFEncoder:
    """The reference's HuggingFace client for this path, restated as it is written
    (app/core/models/huggingface/huggingface.py:165-189), on PyTorch-ROCm from a LOCAL checkpoint:

    embed_text (:165-170)   tokenizer(texts, padding=True) -- NO truncation: a text longer than the model's
        position table fails inside the model, as upstream (EmbeddingHandler.encode_dense then returns [],
        embedding_handler.py:96-98) -- then `last_hidden_state.mean(dim=1)`: the mean runs over EVERY position of the
        padded batch, padding included (no attention-mask weighting), and nothing is normalised.
    rerank_documents (:172-189)   not ColBERT late interaction despite the name: the query alone and the documents
        (padded together) go through the same unmasked mean pooling, scores = q . D^T, `argsort` descending.  Its
        truncation branch is kept as written: it tests `len(doc_tokens) > 8000`, and `len` of a tokenizer's
        BatchEncoding is its number of KEYS (input_ids, attention_mask, ...), so the branch never runs for a real
        tokenizer; if it does, every document is cut to `max_tokens - 5` CHARACTERS plus "....." and tokenized again.
        Named quirk, not fixed: reference parity."""

    def __init__(self, model_path: str, device: Optional[str] = None):
        import torch
        from transformers import AutoModel, AutoTokenizer
        self.torch = torch
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self.tokenizer = AutoTokenizer.from_pretrained(model_path, local_files_only=True)
        self.model = AutoModel.from_pretrained(model_path, local_files_only=True).to(self.device).eval()

    def _forward_mean(self, inputs):
        with self.torch.no_grad():
            return self.model(**inputs).last_hidden_state.mean(dim=1)

    def _pool(self, texts: List[str]):
        """:167-169, on the device (the embeddings stay there for `HxIndex.add_device`)"""
        return self._forward_mean(self.tokenizer(texts, padding=True, return_tensors="pt").to(self.device))

    async def embed_text(self, texts: List[str]) -> List[List[float]]:
        return self._pool(texts).float().cpu().numpy().tolist()

    def rerank_documents(self, query: str, documents: List[str], max_tokens: int) -> List[int]:
        import numpy as np
        query_tokens = self.tokenizer(query, return_tensors="pt").to(self.device)
        doc_tokens = self.tokenizer(documents, return_tensors="pt", padding=True).to(self.device)
        if len(doc_tokens) > 8000:          # number of keys of the BatchEncoding (see the class docstring)
            documents = [doc[:max_tokens - 5] + "....." for doc in documents]
            doc_tokens = self.tokenizer(documents, return_tensors="pt", padding=True).to(self.device)
        query_embedding = self._forward_mean(query_tokens)
        doc_embeddings = self._forward_mean(doc_tokens)
        scores = self.torch.matmul(query_embedding, doc_embeddings.T).squeeze().float().cpu().numpy()
        return np.argsort(scores)[::-1].tolist()


class EmbeddingHandler:
    def __init__(self, provider: Provider = Provider.HUGGINGFACE, model_name: str = None,
                 model_type: ModelType = ModelType.TEXT_EMBEDDING, model: Any = None):
        """`model` = any object with `async embed_text(list[str])`; by default a
        LocalHFEncoder is created lazily from `model_name` treated as a local path."""
        self.provider = provider
        self.model_name = model_name
        self.model_type = model_type
        self.logger = logging.getLogger(__name__)
        self.model = model
        self.cache: Dict[str, Any] = {}

    def _get_cache_key(self, input_data: Union[str, List[str]], embedding_type: str) -> str:
        input_str = "_".join(input_data) if isinstance(input_data, list) else input_data
        hash_key = hashlib.sha256(input_str.encode()).hexdigest()
        return f"embedding:{embedding_type}:{self.provider}:{self.model_name}:{hash_key}"

    async def encode_dense(self, input_data: Union[str, List[str]]) -> List:
        try:
            cache_key = self._get_cache_key(input_data, "dense")
            if cache_key in self.cache:
                self.logger.info("Dense embedding cache hit")
                return self.cache[cache_key]
            if isinstance(input_data, str):
                input_data = [input_data]
            if self.model is None:
                if self.provider != Provider.HUGGINGFACE:
                    raise ValueError("only a local HuggingFace encoder is available offline")
                self.model = LocalHFEncoder(self.model_name)
            result = await self.model.embed_text(input_data)
            if not result:
                raise ValueError("Embedding model returned empty result.")
            self.cache[cache_key] = result
            return result
        except Exception as e:
            self.logger.error(f"Dense embedding failed: {str(e)}")
            return []

    async def encode_sparse(self, text: str):
        try:
            cache_key = self._get_cache_key(text, "sparse")
            if cache_key in self.cache:
                self.logger.info("Sparse embedding cache hit")
                return SparseVector(**self.cache[cache_key])
            indices, values = bm25.embed(text)
            sparse_vector = {"indices": indices, "values": values}
            self.cache[cache_key] = sparse_vector
            return SparseVector(**sparse_vector)
        except Exception as e:
            self.logger.error(f"Sparse embedding failed: {str(e)}")
            return {"indices": [], "values": []}

    async def encode_sparse_batch(self, texts: List[str]):
        """Additive: the reference's per-chunk TODO (embedding_handler.py:100).  One native call
        (csrc/bm25.cpp, all host cores) for the texts that are not cached; same vectors as
        encode_sparse, same failure convention per text."""
        try:
            keys = [self._get_cache_key(t, "sparse") for t in texts]
            todo = [i for i, k in enumerate(keys) if k not in self.cache]
            if todo:
                loop = asyncio.get_running_loop()
                rows = await loop.run_in_executor(None, bm25.embed_batch, [texts[i] for i in todo])
                for i, (indices, values) in zip(todo, rows):
                    self.cache[keys[i]] = {"indices": indices, "values": values}
            return [SparseVector(**self.cache[k]) for k in keys]
        except Exception as e:
            self.logger.error(f"Sparse embedding failed: {str(e)}")
            return [{"indices": [], "values": []} for _ in texts]
