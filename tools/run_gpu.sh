#!/bin/bash
# usage: tools/run_gpu.sh <timeout_s> <script.sh> <tag>   -- rebuild the libraries, check the ABI, then gpurun the script
set -e
cd /root/repo
make -C egovlp_amd/csrc -j8 all diag 2>&1 | grep -iE "error|warning: unused" && exit 1
python -m pytest tests/test_abi.py -x -q 2>&1 | tail -1
/usr/local/graft/bin/gpurun --timeout $1 -- "bash $2 $3"
