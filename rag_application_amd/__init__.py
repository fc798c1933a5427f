"""MI355X-native hybrid dense+sparse retrieval engine: drop-in for the Qdrant-backed
search path of VivekMalipatel/RAG_Application (QdrantHandler / EmbeddingHandler)."""
from ._lib import HX_MODE_H1, HX_MODE_TREE, HxError, HxParams  # noqa: F401

__all__ = ["HX_MODE_H1", "HX_MODE_TREE", "HxError", "HxParams"]
