#!/bin/bash
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_dp_gpu.py -q --tb=short -k "self_launches" 2>&1 | tail -6
