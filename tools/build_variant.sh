#!/bin/bash
# usage: tools/build_variant.sh <git-ref> <suffix>   -- build egovlp_amd/csrc of <git-ref> into egovlp_amd/libegovlp_hip_<suffix>.so
# (same-box A/B of kernel changes: EGOVLP_HIP_LIB=egovlp_amd/libegovlp_hip_<suffix>.so python bench.py ...; the C ABI of <git-ref>
# must be the one the working tree's Python binds)
set -e
cd /root/repo
T=$(mktemp -d)
git archive $1 egovlp_amd/csrc include | tar -x -C $T
make -C $T/egovlp_amd/csrc -j8 LIB=$T/lib.so >/dev/null
cp $T/lib.so egovlp_amd/libegovlp_hip_$2.so
rm -rf $T
ls -la egovlp_amd/libegovlp_hip_$2.so
