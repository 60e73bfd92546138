# Machine image of the status-store node: Redis with AOF persistence, started at boot.
# (Role of terraform/ami_redis.pkr.hcl in the reference, with persistence switched on - the reference's
# Redis AMI ran without it)
packer {
  required_plugins {
    amazon = { source = "github.com/hashicorp/amazon", version = ">= 1.2.0" }
  }
}

variable "region"   { default = "us-west-2" }
variable "base_ami" { description = "Ubuntu 22.04 / Amazon Linux 2023 AMI" }

source "amazon-ebs" "redis" {
  region        = var.region
  source_ami    = var.base_ami
  instance_type = "t3.small"
  ssh_username  = "ubuntu"
  ami_name      = "ocr-redis-{{timestamp}}"
}

build {
  sources = ["source.amazon-ebs.redis"]
  provisioner "file" {
    source      = "../scripts/setup-redis.sh"
    destination = "/tmp/setup-redis.sh"
  }
  provisioner "shell" {
    inline = [
      "chmod +x /tmp/setup-redis.sh && sudo mv /tmp/setup-redis.sh /usr/local/bin/setup-redis.sh",
      "/usr/local/bin/setup-redis.sh",
      "(crontab -l 2>/dev/null; echo '@reboot /usr/local/bin/setup-redis.sh >> /var/log/redis-start.log 2>&1') | crontab -",
      "sudo docker ps | grep ocr-redis"
    ]
  }
}
