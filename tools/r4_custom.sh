#!/bin/bash
OUT=$1
cd "${GRAFT_REPO_ROOT:-.}"
timeout 120 tools/probes/wino_nested_probe.bin | tee "$OUT/wino_nested_probe.txt"
timeout 120 tools/probes/wino_nested_probe.bin | tee -a "$OUT/wino_nested_probe.txt"
