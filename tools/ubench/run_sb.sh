#!/bin/bash
# runs every tools/ubench/sb_* variant of the streaming-kernel harness on the GPU box
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
for f in sb_*; do [ -x $f ] && timeout 120 ./$f 256 $f 2>&1 | grep -v amdgpu.ids; done
