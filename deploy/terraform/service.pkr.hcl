# Machine image of a service node: ROCm-capable OS + Docker + the service image pre-pulled + log agent,
# started at boot.  (Role of terraform/ami.pkr.hcl in the reference.)
packer {
  required_plugins {
    amazon = { source = "github.com/hashicorp/amazon", version = ">= 1.2.0" }
  }
}

variable "region"        { default = "us-west-2" }
variable "base_ami"      { description = "Ubuntu 22.04 AMI with the amdgpu / ROCm 7 driver stack installed" }
variable "instance_type" { default = "mi355x.8xgpu" }
variable "service_image" { default = "dlamd/ocr-service:latest" }

source "amazon-ebs" "service" {
  region        = var.region
  source_ami    = var.base_ami
  instance_type = var.instance_type
  ssh_username  = "ubuntu"
  ami_name      = "ocr-service-mi355x-{{timestamp}}"
  launch_block_device_mappings {
    device_name = "/dev/sda1"
    volume_size = 200
    volume_type = "gp3"
    delete_on_termination = true
  }
}

build {
  sources = ["source.amazon-ebs.service"]

  provisioner "file" {
    source      = "../../"
    destination = "/tmp/ocr-service"
  }

  provisioner "shell" {
    inline = [
      "sudo mkdir -p /opt && sudo mv /tmp/ocr-service /opt/ocr-service && sudo chown -R ubuntu /opt/ocr-service",
      "curl -fsSL https://get.docker.com | sudo sh",
      "sudo usermod -aG docker,video,render ubuntu",
      "cd /opt/ocr-service && sudo docker build -f deploy/Dockerfile -t ${var.service_image} .",
      "cp /opt/ocr-service/deploy/.env.template /opt/ocr-service/.env",
      # metrics/log agent: host metrics + error_*.json + container logs (deploy/observability)
      "sudo mkdir -p /opt/aws/amazon-cloudwatch-agent/etc",
      "sudo cp /opt/ocr-service/deploy/observability/cloudwatch-agent-config.json /opt/aws/amazon-cloudwatch-agent/etc/amazon-cloudwatch-agent.json",
      # start at boot (user-data rewrites .env and restarts with the real Redis address)
      "(crontab -l 2>/dev/null; echo '@reboot /opt/ocr-service/deploy/scripts/run-ocr.sh >> /var/log/ocr-start.log 2>&1') | crontab -",
      # smoke check: the image exists and the GPUs are visible to the container runtime
      "sudo docker image inspect ${var.service_image} >/dev/null && echo image-ok",
      "ls /dev/kfd /dev/dri && rocm-smi --showproductname || echo 'WARNING: no GPU visible while baking the image'"
    ]
  }
}
