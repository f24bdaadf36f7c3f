#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD; mkdir -p gpurun_out/r4p
timeout 600 python -m pytest tests/test_hip_mlp.py -q -m gpu -k "fused_forward_loss_backward" 2>&1 | tail -30
