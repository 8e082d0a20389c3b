#!/bin/bash
# scratch step of tools/r4_visit.sh
OUT=$1
timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu -x > "$OUT/pytest_bf16.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_bf16.log"
