#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "k_tail or wave_specialised" 2>&1 | tail -8
