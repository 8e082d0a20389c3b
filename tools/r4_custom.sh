#!/bin/bash
# scratch step of tools/r4_visit.sh
OUT=$1
echo "custom: nothing to run"
