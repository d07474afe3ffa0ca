#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
( timeout 300 python tools/kbench.py stores ) 2>&1 | grep -v amdgpu.ids
