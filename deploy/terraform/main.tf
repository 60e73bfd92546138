# Infrastructure for the MI355X OCR service on AWS-style primitives.
#
# Same topology as the reference's terraform/main.tf - VPC, two public subnets, ALB -> target group on
# :8000 with /health checks, launch template + Auto Scaling Group of service nodes, one Redis node reachable
# only from the service security group - with two changes:
#   * a service node is an 8-GPU MI355X host running ONE container that shards pages over its 8 GPUs, so the
#     fleet is sized in nodes of 8 GPUs (min 1 / desired 1 / max var.max_nodes) rather than t2.large boxes;
#   * the CPU target-tracking policy the reference only describes in its deployment guide is codified
#     (aws_autoscaling_policy.cpu70), plus a queue-depth alarm hook on the service's own /metrics.
# Instance type / AMI are variables: set them to whatever MI355X offering your cloud exposes; the rest of the
# stack is provider-generic.

terraform {
  required_version = ">= 1.5"
  required_providers {
    aws = { source = "hashicorp/aws", version = "~> 5.0" }
  }
}

provider "aws" {
  region = var.region
}

variable "region"           { default = "us-west-2" }
variable "service_ami_id"   { description = "AMI built by service.pkr.hcl (ROCm driver + Docker + image pre-pulled)" }
variable "redis_ami_id"     { description = "AMI built by redis.pkr.hcl" }
variable "gpu_instance_type" { default = "mi355x.8xgpu" description = "8 x MI355X instance type of the target cloud" }
variable "redis_instance_type" { default = "t3.small" }
variable "key_name"         { default = null }
variable "min_nodes"        { default = 1 }
variable "desired_nodes"    { default = 1 }
variable "max_nodes"        { default = 3 }
variable "mongo_connection_string" {
  description = "MongoDB connection string (no default: the service refuses to invent credentials)"
  sensitive   = true
}
variable "ssh_cidr" { default = "0.0.0.0/0" }

data "aws_availability_zones" "available" { state = "available" }

resource "aws_vpc" "ocr" {
  cidr_block           = "10.0.0.0/16"
  enable_dns_support   = true
  enable_dns_hostnames = true
  tags = { Name = "ocr-vpc" }
}

resource "aws_internet_gateway" "igw" {
  vpc_id = aws_vpc.ocr.id
  tags   = { Name = "ocr-igw" }
}

resource "aws_subnet" "public" {
  count                   = 2
  vpc_id                  = aws_vpc.ocr.id
  cidr_block              = cidrsubnet(aws_vpc.ocr.cidr_block, 8, count.index + 1)
  availability_zone       = data.aws_availability_zones.available.names[count.index]
  map_public_ip_on_launch = true
  tags = { Name = "ocr-public-${count.index}" }
}

resource "aws_route_table" "public" {
  vpc_id = aws_vpc.ocr.id
  route {
    cidr_block = "0.0.0.0/0"
    gateway_id = aws_internet_gateway.igw.id
  }
}

resource "aws_route_table_association" "public" {
  count          = 2
  subnet_id      = aws_subnet.public[count.index].id
  route_table_id = aws_route_table.public.id
}

resource "aws_security_group" "alb" {
  name   = "ocr-alb-sg"
  vpc_id = aws_vpc.ocr.id
  ingress { from_port = 80   to_port = 80   protocol = "tcp" cidr_blocks = ["0.0.0.0/0"] }
  ingress { from_port = 8000 to_port = 8000 protocol = "tcp" cidr_blocks = ["0.0.0.0/0"] }
  egress  { from_port = 0    to_port = 0    protocol = "-1"  cidr_blocks = ["0.0.0.0/0"] }
}

resource "aws_security_group" "service" {
  name   = "ocr-service-sg"
  vpc_id = aws_vpc.ocr.id
  ingress { from_port = 8000 to_port = 8000 protocol = "tcp" security_groups = [aws_security_group.alb.id] }
  ingress { from_port = 22   to_port = 22   protocol = "tcp" cidr_blocks = [var.ssh_cidr] }
  egress  { from_port = 0    to_port = 0    protocol = "-1"  cidr_blocks = ["0.0.0.0/0"] }
}

resource "aws_security_group" "redis" {
  name   = "ocr-redis-sg"
  vpc_id = aws_vpc.ocr.id
  # status store: reachable from service nodes only
  ingress { from_port = 6379 to_port = 6379 protocol = "tcp" security_groups = [aws_security_group.service.id] }
  ingress { from_port = 22   to_port = 22   protocol = "tcp" cidr_blocks = [var.ssh_cidr] }
  egress  { from_port = 0    to_port = 0    protocol = "-1"  cidr_blocks = ["0.0.0.0/0"] }
}

resource "aws_instance" "redis" {
  ami                    = var.redis_ami_id
  instance_type          = var.redis_instance_type
  subnet_id              = aws_subnet.public[0].id
  vpc_security_group_ids = [aws_security_group.redis.id]
  key_name               = var.key_name
  tags = { Name = "ocr-redis" }
}

resource "aws_lb" "ocr" {
  name               = "ocr-alb"
  load_balancer_type = "application"
  security_groups    = [aws_security_group.alb.id]
  subnets            = aws_subnet.public[*].id
  idle_timeout       = 300 # uploads of large scans
}

resource "aws_lb_target_group" "ocr" {
  name     = "ocr-tg"
  port     = 8000
  protocol = "HTTP"
  vpc_id   = aws_vpc.ocr.id
  health_check {
    path                = "/health"
    healthy_threshold   = 2
    unhealthy_threshold = 2
    timeout             = 5
    interval            = 30
    matcher             = "200-399" # 503 = a dependency is down -> the node is taken out of rotation
  }
}

resource "aws_lb_listener" "http8000" {
  load_balancer_arn = aws_lb.ocr.arn
  port              = 8000
  protocol          = "HTTP"
  default_action {
    type             = "forward"
    target_group_arn = aws_lb_target_group.ocr.arn
  }
}

resource "aws_launch_template" "service" {
  name_prefix   = "ocr-service-"
  image_id      = var.service_ami_id
  instance_type = var.gpu_instance_type
  key_name      = var.key_name
  vpc_security_group_ids = [aws_security_group.service.id]
  block_device_mappings {
    device_name = "/dev/sda1"
    ebs { volume_size = 200 volume_type = "gp3" }
  }
  # writes the node's .env (Redis private IP injected) and (re)starts the container
  user_data = base64encode(<<-EOT
    #!/bin/bash
    cat > /opt/ocr-service/.env <<ENV
    MONGO_CONNECTION_STRING=${var.mongo_connection_string}
    MONGO_DB_NAME=ocr_service
    MONGO_COLLECTION_NAME=documents
    REDIS_HOST=${aws_instance.redis.private_ip}
    REDIS_PORT=6379
    PORT=8000
    OCR_STORE=redis+mongo
    OCR_NUM_GPUS=-1
    ENV
    /opt/ocr-service/deploy/scripts/run-ocr.sh
  EOT
  )
  tag_specifications {
    resource_type = "instance"
    tags          = { Name = "ocr-service-node" }
  }
}

resource "aws_autoscaling_group" "service" {
  name                      = "ocr-service-asg"
  min_size                  = var.min_nodes
  desired_capacity          = var.desired_nodes
  max_size                  = var.max_nodes
  vpc_zone_identifier       = aws_subnet.public[*].id
  target_group_arns         = [aws_lb_target_group.ocr.arn]
  health_check_type         = "ELB"
  health_check_grace_period = 600 # model load + kernel warm-up + hipGraph capture
  launch_template {
    id      = aws_launch_template.service.id
    version = "$Latest"
  }
}

resource "aws_autoscaling_policy" "cpu70" {
  name                   = "ocr-cpu-target-70"
  autoscaling_group_name = aws_autoscaling_group.service.name
  policy_type            = "TargetTrackingScaling"
  target_tracking_configuration {
    predefined_metric_specification { predefined_metric_type = "ASGAverageCPUUtilization" }
    target_value = 70
  }
}

output "alb_dns_name"     { value = aws_lb.ocr.dns_name }
output "redis_private_ip" { value = aws_instance.redis.private_ip }
output "service_url"      { value = "http://${aws_lb.ocr.dns_name}:8000" }
