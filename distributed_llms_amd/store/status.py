"""Document status store.

Key layout and TTL are the reference's (services/ocr_service/ocr_service.py:63-108):

    doc:{document_id}  -> JSON status dict      SETEX 86400
    hash:{sha256}      -> document_id            SETEX 86400   (written, never read by the reference)

Every write re-arms the 24 h TTL.  All errors are swallowed and reported as
``False`` / ``None`` exactly like the reference helpers, so a Redis outage
degrades the API instead of crashing it.

Fix over the reference: the ``queued`` claim can be made atomic with
``claim()`` (``SET ... NX EX``) so two replicas cannot both start the same id.
"""
from __future__ import annotations

import json
import logging
import threading
import time
from typing import Callable, Dict, Iterable, Optional, Tuple

from ..config import STATUS_TTL_SECONDS
from .resp import RespClient

logger = logging.getLogger("distributed_llms_amd")

RAW_STATUSES = ("queued", "processing", "completed", "failed")


def public_status(raw: str) -> str:
    """Map the internal lifecycle word to the public /check-status vocabulary (ocr_service.py:486-492)."""
    if raw == "completed":
        return "completed"
    if raw == "failed":
        return "error"
    return "processing"


class StatusStore:
    """Interface + shared JSON handling."""

    ttl: int = STATUS_TTL_SECONDS
    in_process: bool = False      # True: lookups never block (no socket), callers may skip the I/O thread

    # -- primitives implemented by backends ------------------------------------
    def _setex(self, key: str, ttl: int, value: str) -> None:
        raise NotImplementedError

    def _setnx(self, key: str, ttl: int, value: str) -> bool:
        raise NotImplementedError

    def _get(self, key: str) -> Optional[str]:
        raise NotImplementedError

    def _exists(self, key: str) -> bool:
        raise NotImplementedError

    def _delete(self, key: str) -> None:
        raise NotImplementedError

    def _keys(self, pattern: str) -> Iterable[str]:
        raise NotImplementedError

    def ping(self) -> bool:
        raise NotImplementedError

    def close(self) -> None:
        pass

    # -- reference-compatible API --------------------------------------------------
    def update(self, document_id: Optional[str], status_data: dict) -> bool:
        """``update_document_status`` (ocr_service.py:63-83)."""
        try:
            if document_id is None:
                logger.error("Cannot update status: document_id is None")
                return False
            if status_data.get("document_hash") is not None:
                self._setex(f"hash:{status_data['document_hash']}", self.ttl, document_id)
            self._setex(f"doc:{document_id}", self.ttl, json.dumps(status_data))
            logger.info(f"Updated status for document {document_id}: {status_data['status']}")
            return True
        except Exception as e:  # noqa: BLE001 - parity: helpers never raise
            logger.error(f"Error updating document status in Redis: {str(e)}")
            return False

    def claim(self, document_id: str, status_data: dict) -> bool:
        """Atomically create ``doc:{id}`` if absent.  True when this caller owns the job."""
        try:
            ok = self._setnx(f"doc:{document_id}", self.ttl, json.dumps(status_data))
            if ok and status_data.get("document_hash") is not None:
                self._setex(f"hash:{status_data['document_hash']}", self.ttl, document_id)
            if ok:
                logger.info(f"Updated status for document {document_id}: {status_data['status']}")
            return ok
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error claiming document in Redis: {str(e)}")
            return False

    def get(self, document_id: str) -> Optional[dict]:
        """``get_document_status`` (ocr_service.py:86-95)."""
        try:
            raw = self._get(f"doc:{document_id}")
            if raw:
                return json.loads(raw)
            return None
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error retrieving document status from Redis: {str(e)}")
            return None

    def get_raw(self, document_id: str) -> Optional[str]:
        """Raw JSON string; raises on connection errors (used by /result, which maps them to 500)."""
        return self._get(f"doc:{document_id}")

    def exists(self, document_id: str) -> bool:
        """``check_document_exists_by_id`` (ocr_service.py:97-108)."""
        try:
            return bool(self._exists(f"doc:{document_id}"))
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error checking document existence in Redis: {str(e)}")
            return False

    def delete(self, document_id: str) -> None:
        try:
            self._delete(f"doc:{document_id}")
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error deleting document status: {e}")

    def document_for_hash(self, sha256_hex: str) -> Optional[str]:
        try:
            return self._get(f"hash:{sha256_hex}")
        except Exception:  # noqa: BLE001
            return None

    def sweep_orphans(self, owner: Optional[str] = None, reason: str = "worker restarted") -> int:
        """Mark ``queued``/``processing`` documents (optionally only those owned by ``owner``) as failed.

        The reference strands such keys for 24 h after a restart."""
        n = 0
        try:
            for key in list(self._keys("doc:*")):
                raw = self._get(key)
                if not raw:
                    continue
                try:
                    st = json.loads(raw)
                except ValueError:
                    continue
                if st.get("status") in ("queued", "processing") and (owner is None or st.get("owner") == owner):
                    st.update(status="failed", error=f"orphaned job: {reason}")
                    self._setex(key, self.ttl, json.dumps(st))
                    n += 1
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error sweeping orphaned documents: {e}")
        return n


class RedisStatusStore(StatusStore):
    def __init__(self, client: RespClient, ttl: int = STATUS_TTL_SECONDS):
        self.client, self.ttl = client, ttl

    def _setex(self, key, ttl, value):
        self.client.setex(key, ttl, value)

    def _setnx(self, key, ttl, value):
        return self.client.set(key, value, ex=ttl, nx=True)

    def _get(self, key):
        return self.client.get(key)

    def _exists(self, key):
        return self.client.exists(key) > 0

    def _delete(self, key):
        self.client.delete(key)

    def _keys(self, pattern):
        return self.client.scan_iter(match=pattern)

    def ping(self) -> bool:
        return self.client.ping()

    def close(self) -> None:
        self.client.close()


class MemoryStatusStore(StatusStore):
    """In-process store with the same TTL semantics (``OCR_STORE=memory``, unit tests)."""

    in_process = True

    def __init__(self, ttl: int = STATUS_TTL_SECONDS, clock: Callable[[], float] = time.monotonic):
        self.ttl, self.clock = ttl, clock
        self._d: Dict[str, Tuple[str, float]] = {}
        self._lock = threading.Lock()
        self.healthy = True

    def _live(self, key):
        ent = self._d.get(key)
        if ent is None:
            return None
        if self.clock() >= ent[1]:
            del self._d[key]
            return None
        return ent[0]

    def _check(self):
        if not self.healthy:
            raise ConnectionError("status store is down")

    def _setex(self, key, ttl, value):
        self._check()
        with self._lock:
            self._d[key] = (value, self.clock() + ttl)

    def _setnx(self, key, ttl, value):
        self._check()
        with self._lock:
            if self._live(key) is not None:
                return False
            self._d[key] = (value, self.clock() + ttl)
            return True

    def _get(self, key):
        self._check()
        with self._lock:
            return self._live(key)

    def _exists(self, key):
        self._check()
        with self._lock:
            return self._live(key) is not None

    def _delete(self, key):
        self._check()
        with self._lock:
            self._d.pop(key, None)

    def _keys(self, pattern):
        import fnmatch

        self._check()
        with self._lock:
            return [k for k in list(self._d) if self._live(k) is not None and fnmatch.fnmatchcase(k, pattern)]

    def ttl_of(self, key: str) -> Optional[float]:
        with self._lock:
            ent = self._d.get(key)
            return None if ent is None else ent[1] - self.clock()

    def ping(self) -> bool:
        self._check()
        return True
