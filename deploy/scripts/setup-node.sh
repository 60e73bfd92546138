#!/usr/bin/env bash
# Prepare a fresh MI355X host (ROCm driver already installed) to run the service in Docker.
# Role of the reference's setup-ocr.sh (Docker install + image build).
set -euo pipefail

if ! command -v docker >/dev/null 2>&1; then
  curl -fsSL https://get.docker.com | sh
  sudo usermod -aG docker "$USER" || true
fi
sudo systemctl enable --now docker

# the container needs the kfd / dri device nodes and the video+render groups
for dev in /dev/kfd /dev/dri; do
  [ -e "$dev" ] || { echo "missing $dev: is the amdgpu driver loaded?" >&2; exit 1; }
done
rocm-smi --showproductname || true

REPO_DIR="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$REPO_DIR"
[ -f .env ] || { cp deploy/.env.template .env; echo "edit $REPO_DIR/.env (MONGO_CONNECTION_STRING, REDIS_HOST) before starting"; }
mkdir -p output weights
docker build -f deploy/Dockerfile -t dlamd/ocr-service:latest .
echo "image built: dlamd/ocr-service:latest"
