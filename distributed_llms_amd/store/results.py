"""Result store, error sink and image writer.

The Mongo document is this repository's "checkpoint format":

    document_id: str, combined_text: str, status: 'completed',
    processed_at: datetime, word_count: int (= len(text.split())),
    character_count: int (= len(text))

written with ``update_one({"document_id"}, {"$set": doc}, upsert=True)``
(services/ocr_service/ocr_service.py:357-371) and indexed on ``document_id``
(unique), ``document_hash``, ``status``, ``processed_at`` (ocr_service.py:147-150).
Failed documents are never written here; they go to ``output/error_{id}.json``
(ocr_service.py:387-410), which the CloudWatch/log agent tails.
"""
from __future__ import annotations

import datetime
import json
import logging
import os
import threading
from typing import Callable, Dict, Optional

logger = logging.getLogger("distributed_llms_amd")

INDEXES = (("document_id", True), ("document_hash", False), ("status", False), ("processed_at", False))


def make_result_document(document_id: str, text: str, now: Optional[datetime.datetime] = None) -> dict:
    """Exactly the six fields of the reference schema (ocr_service.py:357-364)."""
    return {
        "document_id": document_id,
        "combined_text": text,
        "status": "completed",
        "processed_at": now or datetime.datetime.now(),
        "word_count": len(text.split()),
        "character_count": len(text),
    }


class ResultStore:
    def connect(self) -> None:
        pass

    def ensure_indexes(self) -> None:
        pass

    def upsert(self, doc: dict) -> None:
        raise NotImplementedError

    def find(self, document_id: str) -> Optional[dict]:
        raise NotImplementedError

    def ping(self) -> dict:
        raise NotImplementedError

    def close(self) -> None:
        pass


class MemoryResultStore(ResultStore):
    def __init__(self):
        self._docs: Dict[str, dict] = {}
        self._lock = threading.Lock()
        self.healthy = True
        self.indexes = []

    def ensure_indexes(self) -> None:
        self.indexes = [name for name, _ in INDEXES]

    def _check(self):
        if not self.healthy:
            raise ConnectionError("result store is down")

    def upsert(self, doc: dict) -> None:
        self._check()
        with self._lock:
            cur = self._docs.setdefault(doc["document_id"], {})
            cur.update(doc)

    def find(self, document_id: str) -> Optional[dict]:
        self._check()
        with self._lock:
            d = self._docs.get(document_id)
            return None if d is None else dict(d)

    def ping(self) -> dict:
        self._check()
        return {"ok": 1}

    def count(self) -> int:
        return len(self._docs)


class FileResultStore(ResultStore):
    """Result rows as one JSON file each under ``<dir>/<sha1(document_id)>.json`` (atomic rename).

    For a single node with several front-end processes and no MongoDB (``OCR_STORE=redis+file``): every
    process sees every row through the shared directory.  Same row schema as Mongo; ``processed_at`` is
    stored as an ISO string and handed back as ``datetime``."""

    def __init__(self, directory: str):
        self.dir = directory

    def connect(self) -> None:
        os.makedirs(self.dir, exist_ok=True)

    def _path(self, document_id: str) -> str:
        import hashlib

        return os.path.join(self.dir, hashlib.sha1(document_id.encode("utf-8")).hexdigest() + ".json")

    def upsert(self, doc: dict) -> None:
        os.makedirs(self.dir, exist_ok=True)
        cur = self.find(doc["document_id"]) or {}
        cur.update(doc)
        row = {k: (v.isoformat() if isinstance(v, datetime.datetime) else v) for k, v in cur.items()}
        path = self._path(doc["document_id"])
        tmp = f"{path}.tmp{os.getpid()}.{threading.get_ident()}"
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(row, f)
        os.replace(tmp, path)

    def find(self, document_id: str) -> Optional[dict]:
        try:
            with open(self._path(document_id), "r", encoding="utf-8") as f:
                row = json.load(f)
        except FileNotFoundError:
            return None
        if isinstance(row.get("processed_at"), str):
            try:
                row["processed_at"] = datetime.datetime.fromisoformat(row["processed_at"])
            except ValueError:
                pass
        return row

    def ping(self) -> dict:
        os.makedirs(self.dir, exist_ok=True)
        if not os.access(self.dir, os.W_OK):
            raise ConnectionError(f"result directory {self.dir} is not writable")
        return {"ok": 1}


class MongoResultStore(ResultStore):
    """pymongo-backed store; connection options follow ocr_service.py:134-137."""

    def __init__(self, uri: str, db_name: str = "ocr_service", collection: str = "documents",
                 tls: Optional[bool] = None, server_selection_timeout_ms: int = 5000):
        if not uri:
            raise ValueError("MongoResultStore needs a connection string (no default credential is shipped)")
        self.uri, self.db_name, self.collection_name = uri, db_name, collection
        self.tls = uri.startswith("mongodb+srv://") if tls is None else tls
        self.timeout = server_selection_timeout_ms
        self.client = None
        self.collection = None

    def connect(self) -> None:
        from pymongo import MongoClient

        kw = dict(serverSelectionTimeoutMS=self.timeout)
        if self.tls:
            kw.update(tls=True, tlsAllowInvalidCertificates=True)
        self.client = MongoClient(self.uri, **kw)
        self.client.admin.command("ping")
        self.collection = self.client[self.db_name][self.collection_name]
        logger.info("Connected to MongoDB successfully")

    def ensure_indexes(self) -> None:
        for name, unique in INDEXES:
            self.collection.create_index(name, unique=unique)

    def upsert(self, doc: dict) -> None:
        self.collection.update_one({"document_id": doc["document_id"]}, {"$set": doc}, upsert=True)

    def find(self, document_id: str) -> Optional[dict]:
        return self.collection.find_one({"document_id": document_id})

    def ping(self) -> dict:
        return self.client.admin.command("ping")

    def close(self) -> None:
        if self.client is not None:
            self.client.close()
            logger.info("MongoDB connection closed")


class ErrorSink:
    """``save_error`` (ocr_service.py:387-410): output/error_{id}.json, picked up by the log agent."""

    def __init__(self, output_dir: str = "output"):
        self.output_dir = output_dir

    def path_for(self, document_id: str) -> str:
        safe = "".join(c if (c.isalnum() or c in "-_.") else "_" for c in str(document_id))
        return os.path.join(self.output_dir, f"error_{safe}.json")

    def save(self, document_id: str, error_message: str) -> bool:
        try:
            payload = {
                "document_id": document_id,
                "status": "failed",
                "error": error_message,
                "processed_at": datetime.datetime.now().isoformat(),
            }
            os.makedirs(self.output_dir, exist_ok=True)
            with open(self.path_for(document_id), "w", encoding="utf-8") as f:
                json.dump(payload, f, ensure_ascii=False, indent=2)
            logger.info(f"Saved error information for {document_id}")
            return True
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error saving error information for {document_id}: {str(e)}")
            return False


class LocalFileWriter:
    """``LocalFileWriter`` (ocr_service.py:111-124): figure / table crops under output/images."""

    def __init__(self, base_dir: str):
        self.base_dir = base_dir
        os.makedirs(self.base_dir, exist_ok=True)

    def write(self, path: str, data: bytes) -> str:
        full_path = os.path.normpath(os.path.join(self.base_dir, path))
        base = os.path.abspath(self.base_dir)
        if not os.path.abspath(full_path).startswith(base + os.sep):
            raise ValueError(f"refusing to write outside {self.base_dir}: {path}")
        os.makedirs(os.path.dirname(full_path), exist_ok=True)
        with open(full_path, "wb") as f:
            f.write(data)
        return full_path
