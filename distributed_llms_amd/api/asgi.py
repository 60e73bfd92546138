"""``uvicorn distributed_llms_amd.api.asgi:app`` - module-level ASGI object, like the reference's
``ocr_service:app`` (services/ocr_service/Dockerfile:28).  Settings come from the environment / ``.env``."""
import os

from ..config import Settings
from .app import create_app

app = create_app(Settings.from_env(dotenv=os.environ.get("OCR_DOTENV", ".env")))
