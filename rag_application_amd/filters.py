"""Payload filters of the reference's search path.

`hybrid_search(..., filters=...)` turns the dict into `qdrant_client.models.Filter(**filters)` and
passes it as `query_filter` of the ROOT query only (app/core/vector_store/qdrant/qdrant_handler.py:297,
371; the prefetches carry no filter), i.e. the union of the two branches' candidates is re-scored by
the dense vector, filtered, and cut to `final_limit`.  `get_collection_chunk_count(filters=...)`
counts the points the filter keeps (:464-470).  No call site of the reference passes filters today.

Qdrant is absent from the image, so the condition semantics are restated from its documented model
(PARITY UNPINNED): a Filter has `must` (all hold), `should` (at least one holds, when any is listed;
`min_should` is not supported), `must_not` (none holds); a condition is a nested Filter or one of
  {"key": k, "match": {"value": v}}      payload[k] == v, or v in payload[k] when that is a list
  {"key": k, "match": {"any": [...]}}    payload[k] (or one of its elements) is in the list
  {"key": k, "match": {"except": [...]}} payload[k] exists and is not in the list
  {"key": k, "match": {"text": s}}       every whitespace-separated word of s occurs in str(payload[k])
  {"key": k, "range": {"gt"|"gte"|"lt"|"lte": x}}   numeric comparison
  {"is_empty": {"key": k}}               k missing, None or []
  {"is_null": {"key": k}}                k present with value None
  {"has_id": [ids]}                      the point id is listed
Keys may be dotted paths ("chunk_metadata.page")."""
from __future__ import annotations

from typing import Any, Dict, Iterable, Optional

_MISSING = object()


def _get(payload: Dict[str, Any], key: str):
    cur: Any = payload
    for part in str(key).split("."):
        if isinstance(cur, dict) and part in cur:
            cur = cur[part]
        else:
            return _MISSING
    return cur


def _values(v) -> Iterable[Any]:
    return v if isinstance(v, (list, tuple)) else (v,)


def _match(value, m: Dict[str, Any]) -> bool:
    if value is _MISSING or value is None:
        return False
    if "value" in m:
        return any(x == m["value"] and type(x) is type(m["value"]) or (x == m["value"] and not isinstance(x, bool)
                   and not isinstance(m["value"], bool)) for x in _values(value))
    if "any" in m:
        return any(x in m["any"] for x in _values(value))
    if "except" in m:
        return all(x not in m["except"] for x in _values(value))
    if "text" in m:
        hay = str(value).lower()
        return all(w in hay for w in str(m["text"]).lower().split())
    raise ValueError(f"unsupported match condition: {m}")


def _range(value, r: Dict[str, Any]) -> bool:
    def ok(x):
        if isinstance(x, bool) or not isinstance(x, (int, float)):
            return False
        return ((("gt" not in r) or r["gt"] is None or x > r["gt"]) and (("gte" not in r) or r["gte"] is None or x >= r["gte"])
                and (("lt" not in r) or r["lt"] is None or x < r["lt"]) and (("lte" not in r) or r["lte"] is None or x <= r["lte"]))
    return value is not _MISSING and any(ok(x) for x in _values(value))


def _condition(c: Dict[str, Any], payload: Dict[str, Any], point_id) -> bool:
    if any(k in c for k in ("must", "should", "must_not")):
        return matches(payload, c, point_id)
    if "has_id" in c:
        return point_id in c["has_id"]
    if "is_empty" in c:
        v = _get(payload, c["is_empty"]["key"])
        return v is _MISSING or v is None or v == []
    if "is_null" in c:
        return _get(payload, c["is_null"]["key"]) is None
    if "key" in c:
        v = _get(payload, c["key"])
        if "match" in c:
            return _match(v, c["match"])
        if "range" in c:
            return _range(v, c["range"])
    raise ValueError(f"unsupported filter condition: {c}")


def _as_list(x):
    if x is None:
        return []
    return list(x) if isinstance(x, (list, tuple)) else [x]


def matches(payload: Dict[str, Any], flt: Optional[Dict[str, Any]], point_id=None) -> bool:
    """True when the point (payload, id) passes the filter; an empty / None filter passes all."""
    if not flt:
        return True
    unknown = set(flt) - {"must", "should", "must_not"}
    if unknown:
        raise ValueError(f"unsupported filter clause(s): {sorted(unknown)}")
    must, should, must_not = _as_list(flt.get("must")), _as_list(flt.get("should")), _as_list(flt.get("must_not"))
    if not all(_condition(c, payload, point_id) for c in must):
        return False
    if any(_condition(c, payload, point_id) for c in must_not):
        return False
    if should and not any(_condition(c, payload, point_id) for c in should):
        return False
    return True
