"""PDF text extraction service for MI355X: HIP kernels behind a FastAPI-compatible job API.

Sub-packages: ``api`` (HTTP front end), ``jobs`` / ``store`` (status and result stores), ``pdf`` (C++ PDF reader,
synthetic documents), ``runtime`` (engine, batching, worker processes), ``models`` / ``ops`` (networks and their
kernels), ``parallel`` / ``train`` (data-parallel training), ``tools``, ``utils``.  Nothing here touches the GPU at
import time.
"""

__version__ = "1.0.0"
